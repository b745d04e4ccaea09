// Layout probe for the fp64 MFMA instructions on gfx950 (no ISA manual in this environment): which lanes' A and B
// operands meet in which lane / register of D.  A[lane] = lane + 1, B = 1 in ONE lane (0 elsewhere), C = 0: the non-zero
// D entries then name the A lanes that were multiplied with that B lane, and where the products land.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe4(double* out) {       // v_mfma_f64_4x4x4f64: one double of A, B and D per lane
  const int lane = threadIdx.x;
  for (int bl = 0; bl < 64; ++bl) {
    const double a = (double)(lane + 1), b = (lane == bl) ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[bl * 64 + lane] = d;
  }
}
__global__ void probe16(double* out) {      // v_mfma_f64_16x16x4f64: one double of A and B, four of D per lane
  const int lane = threadIdx.x;
  for (int bl = 0; bl < 64; ++bl) {
    const double a = (double)(lane + 1), b = (lane == bl) ? 1.0 : 0.0;
    d4 c = {0.0, 0.0, 0.0, 0.0};
    const d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(bl * 64 + lane) * 4 + r] = d[r];
  }
}
// Issue rates: every wave runs `iters` rounds of 8 independent accumulators, so the pipe (not the dependency) is what is timed.
__global__ void rate_mfma4(double* out, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4, c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[u], 0, 0, 0);
  double s = 0; for (int u = 0; u < 8; ++u) s += c[u];
  if (s == 1.2345e300) out[0] = s;
}
__global__ void rate_mfma16(double* out, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  d4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[u], 0, 0, 0);
  double s = 0; for (int u = 0; u < 4; ++u) s += c[u][0] + c[u][1] + c[u][2] + c[u][3];
  if (s == 1.2345e300) out[0] = s;
}
__global__ void rate_fma(double* out, int iters) {
  double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-7, c[8] = {0, 1, 2, 3, 4, 5, 6, 7};
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = __builtin_fma(a, c[u], b);
  double s = 0; for (int u = 0; u < 8; ++u) s += c[u];
  if (s == 1.2345e300) out[0] = s;
}
template <class K> static double time_ms(K k, double* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<<<256 * 8, 256>>>(d, iters); hipDeviceSynchronize();
  hipEventRecord(e0); k<<<256 * 8, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  double *d4o, *d16o;
  hipMalloc(&d4o, 64 * 64 * 8); hipMalloc(&d16o, 64 * 64 * 4 * 8);
  probe4<<<1, 64>>>(d4o); probe16<<<1, 64>>>(d16o);
  static double h4[64 * 64], h16[64 * 64 * 4];
  hipMemcpy(h4, d4o, sizeof h4, hipMemcpyDeviceToHost); hipMemcpy(h16, d16o, sizeof h16, hipMemcpyDeviceToHost);
  printf("== v_mfma_f64_4x4x4f64: for B lane bl: D lane <- A lane\n");
  for (int bl = 0; bl < 64; ++bl) {
    printf("bl %2d:", bl);
    for (int l = 0; l < 64; ++l) if (h4[bl * 64 + l] != 0.0) printf(" D%d<-A%d", l, (int)h4[bl * 64 + l] - 1);
    printf("\n");
  }
  printf("== v_mfma_f64_16x16x4f64: for B lane bl: D (lane,reg) <- A lane (first 8 B lanes and lanes 16, 32, 48)\n");
  for (int bl = 0; bl < 64; ++bl) {
    if (!(bl < 8 || bl == 16 || bl == 32 || bl == 48)) continue;
    printf("bl %2d:", bl);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (h16[(bl * 64 + l) * 4 + r] != 0.0) printf(" D(%d,%d)<-A%d", l, r, (int)h16[(bl * 64 + l) * 4 + r] - 1);
    printf("\n");
  }
  // rates: 256 CUs x 8 workgroups x 4 waves
  const int iters = 20000; const double waves = 256.0 * 8 * 4;
  double ms = time_ms(rate_mfma4, d4o, iters);
  printf("== rate v_mfma_f64_4x4x4f64 : %.3f ms -> %.1f TFLOP/s (512 flop per wave-instruction)\n", ms, waves * iters * 8 * 512.0 / ms / 1e9);
  ms = time_ms(rate_mfma16, d4o, iters);
  printf("== rate v_mfma_f64_16x16x4f64: %.3f ms -> %.1f TFLOP/s (2048 flop per wave-instruction)\n", ms, waves * iters * 4 * 2048.0 / ms / 1e9);
  ms = time_ms(rate_fma, d4o, iters);
  printf("== rate v_fma_f64 (VALU)     : %.3f ms -> %.1f TFLOP/s (128 flop per wave-instruction)\n", ms, waves * iters * 8 * 128.0 / ms / 1e9);
  return 0;
}
