// Measurement prototype, not part of the library: y += A x with the entries of A re-ordered by (column slab, row) so that the x
// segment a launch window touches fits an XCD's L2; every product goes to y through a global fp64 atomic (rows ascend inside a slab,
// so the atomics of a wave fall into a few cache lines).  Built and run by tools/proto/colslab.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int U>
__global__ __launch_bounds__(256) void colslab_kernel(int64_t n, const int32_t* __restrict__ row, const int32_t* __restrict__ col,
                                                      const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y) {
  const int64_t base = (int64_t)blockIdx.x * (256 * U);
  int32_t r[U], c[U]; double v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + u * 256 + threadIdx.x;
    const bool ok = i < n;
    r[u] = ok ? row[i] : -1; c[u] = ok ? col[i] : 0; v[u] = ok ? val[i] : 0.0;
  }
  double p[U];
#pragma unroll
  for (int u = 0; u < U; ++u) p[u] = v[u] * x[c[u]];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    // neighbours with the same row fold into the lowest lane of the run (runs are short: one step of each width is enough for the test)
    double s = p[u]; const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double t = __shfl_down(s, o, 64); const int rr = __shfl_down(r[u], o, 64);
      if (lane + o < 64 && rr == r[u]) s += t;
    }
    const int rp = __shfl_up(r[u], 1, 64);
    if (r[u] >= 0 && (lane == 0 || rp != r[u])) unsafeAtomicAdd(&y[r[u]], s);
  }
}
// same stream, no atomics: the products are summed into a dummy (what the loads alone cost)
template <int U>
__global__ __launch_bounds__(256) void colslab_loads_kernel(int64_t n, const int32_t* __restrict__ row, const int32_t* __restrict__ col,
                                                            const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y) {
  const int64_t base = (int64_t)blockIdx.x * (256 * U);
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + u * 256 + threadIdx.x;
    if (i < n) acc += val[i] * x[col[i]] + (double)row[i];
  }
  if (acc == 1.2345e300) y[0] = acc;
}
extern "C" int proto_colslab(int64_t n, const void* row, const void* col, const void* val, const void* x, void* y, int variant, void* stream) {
  const int U = 8;
  const unsigned grid = (unsigned)((n + 256 * U - 1) / (256 * U));
  if (variant == 0) hipLaunchKernelGGL(colslab_kernel<8>, grid, 256, 0, (hipStream_t)stream, n, (const int32_t*)row, (const int32_t*)col, (const double*)val, (const double*)x, (double*)y);
  else hipLaunchKernelGGL(colslab_loads_kernel<8>, grid, 256, 0, (hipStream_t)stream, n, (const int32_t*)row, (const int32_t*)col, (const double*)val, (const double*)x, (double*)y);
  return (int)hipGetLastError();
}
