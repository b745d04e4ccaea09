// LDS / gather rate probe for gfx950 (measurement aid for the SpGEMM accumulators; tools only).  Every workgroup owns a table in LDS
// and every work-item issues `iters` rounds of U independent operations at pseudo-random addresses (an LCG per lane: no two lanes of a
// wave share a word except by chance, like hashed columns).  Reported: operations per clock per CU at 2.4 GHz and G operations / s
// over the whole chip, for
//   ds_read_b32 / ds_read_b64           plain reads
//   ds_or_b32 (no return)               bitmap set
//   ds_add_f64 (no return)              value accumulate
//   ds_add_rtn_u32 / ds_cmpst_rtn_b32   returning atomics (hash insert)
//   read b64 + popc + ds_add_f64        the rank-window lookup (packed {bits, prefix} word, then the add)
//   read b32 key + compare + ds_add_f64 a hash hit without a collision
// and for global gathers of 4-byte / 16-byte pieces from a 200 MB array (what entries(B) looks like at R-MAT scale 20).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kWords = 16384;       // 64 KB of 32-bit words / 128 KB of 64-bit words
template <int OP, int NT>
__global__ __launch_bounds__(NT) void lds_kernel(int iters, unsigned long long* sink) {
  extern __shared__ unsigned long long smem[];
  unsigned* w32 = reinterpret_cast<unsigned*>(smem);
  double* f64 = reinterpret_cast<double*>(smem);
  const int nw64 = OP == 0 || OP == 2 || OP == 4 || OP == 5 ? kWords / 2 : kWords;
  for (int i = threadIdx.x; i < nw64; i += NT) smem[i] = 0ull;
  __syncthreads();
  unsigned s = (threadIdx.x * 2654435761u + blockIdx.x * 40503u) | 1u;
  unsigned long long acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s = s * 1664525u + 1013904223u;
      const unsigned a = (s >> 10) & (kWords - 1);
      if (OP == 0) acc += w32[a];                                            // ds_read_b32
      else if (OP == 1) acc += smem[a];                                      // ds_read_b64
      else if (OP == 2) atomicOr(&w32[a], 1u << (s & 31));                   // ds_or_b32
      else if (OP == 3) unsafeAtomicAdd(&f64[a], 1.0);                       // ds_add_f64
      else if (OP == 4) acc += atomicAdd(&w32[a], 1u);                       // ds_add_rtn_u32
      else if (OP == 5) acc += atomicCAS(&w32[a], 0xffffffffu, s);           // ds_cmpst_rtn_b32
      else if (OP == 6) {                                                    // packed {bits, prefix} read + popc + add at the rank
        const unsigned long long pw = smem[a & (kWords / 2 - 1)];
        const unsigned r = (unsigned)(pw >> 32) + __popc((unsigned)pw & ((1u << (s & 31)) - 1u));
        unsafeAtomicAdd(&f64[kWords / 2 + ((r + a) & (kWords / 2 - 1))], 1.0);
      } else if (OP == 7) {                                                  // hash hit: key read, compare, add
        const unsigned k = w32[a & (kWords / 4 - 1)];
        if (k == 0u) unsafeAtomicAdd(&f64[kWords / 8 + (a & (kWords / 4 - 1))], 1.0);
      }
    }
  }
  if (acc == 0x123456789abcdefull) sink[0] = acc;
}

template <int BYTES>
__global__ __launch_bounds__(256) void gather_kernel(const int* __restrict__ src, size_t n_elems, int iters, int run, unsigned long long* sink) {
  unsigned s = ((blockIdx.x * 256 + threadIdx.x) * 2654435761u) | 1u;
  long long acc = 0;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    size_t at[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s = s * 1664525u + 1013904223u;
      // run = lanes that read consecutive pieces (1: every lane somewhere else; 16 / 64: contiguous runs like a B row's segment)
      unsigned base = __shfl(s, lane & ~(run - 1), 64);
      at[u] = ((size_t)(base >> 4) * 16 + (size_t)(lane & (run - 1)) * (BYTES / 4)) % (n_elems - 64);
      at[u] &= ~(size_t)(BYTES / 4 - 1);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (BYTES == 4) acc += src[at[u]];
      else { const int4 v = *reinterpret_cast<const int4*>(src + at[u]); acc += v.x + v.y + v.z + v.w; }
    }
  }
  if (acc == 0x123456789abcdefll) sink[0] = (unsigned long long)acc;
}

template <int OP, int NT> static void run_lds(const char* name, int cus, unsigned long long* sink) {
  const int iters = 2000, wg_per_cu = NT == 1024 ? 1 : 2;
  const size_t lds = (size_t)kWords * (OP == 1 || OP == 3 || OP == 6 ? 8 : 4) + (OP == 7 ? 0 : 0);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)lds_kernel<OP, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  lds_kernel<OP, NT><<<cus * wg_per_cu, NT, lds, 0>>>(10, sink);
  CK(hipEventRecord(e0));
  lds_kernel<OP, NT><<<cus * wg_per_cu, NT, lds, 0>>>(iters, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double ops = (double)cus * wg_per_cu * NT * iters * 8;
  printf("%-44s NT=%4d x %d wg/CU  %8.3f ms  %7.1f G ops/s  %5.2f lanes/clk/CU (2.4 GHz)\n", name, NT, wg_per_cu, ms, ops / ms / 1e6, ops / (ms * 1e-3) / cus / 2.4e9);
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
  unsigned long long* sink; CK(hipMalloc(&sink, 64));
  run_lds<0, 1024>("ds_read_b32 random", cus, sink);
  run_lds<1, 1024>("ds_read_b64 random", cus, sink);
  run_lds<2, 1024>("ds_or_b32 (no return) random", cus, sink);
  run_lds<3, 1024>("ds_add_f64 (no return) random", cus, sink);
  run_lds<4, 1024>("ds_add_rtn_u32 random", cus, sink);
  run_lds<5, 1024>("ds_cmpst_rtn_b32 random", cus, sink);
  run_lds<6, 1024>("read b64 + popc + ds_add_f64 (rank window)", cus, sink);
  run_lds<7, 1024>("read b32 key + cmp + ds_add_f64 (hash hit)", cus, sink);
  run_lds<2, 512>("ds_or_b32 (no return) random", cus, sink);
  run_lds<3, 512>("ds_add_f64 (no return) random", cus, sink);
  run_lds<6, 512>("read b64 + popc + ds_add_f64 (rank window)", cus, sink);
  // gathers from 200 MB
  const size_t n = (size_t)50 << 20;
  int* src; CK(hipMalloc(&src, n * 4)); CK(hipMemset(src, 1, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int run : {1, 16, 64}) {
    for (int bytes : {4, 16}) {
      const int iters = 200, grid = cus * 32;
      auto launch = [&](int it) { if (bytes == 4) gather_kernel<4><<<grid, 256>>>(src, n, it, run, sink); else gather_kernel<16><<<grid, 256>>>(src, n, it, run, sink); };
      launch(5);
      CK(hipEventRecord(e0)); launch(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double loads = (double)grid * 256 * iters * 4;
      printf("gather %2d B pieces, runs of %2d lanes, 200 MB: %8.3f ms  %7.1f G loads/s  %7.1f GB/s useful\n", bytes, run, ms, loads / ms / 1e6, loads * bytes / ms / 1e6);
    }
  }
  return 0;
}
