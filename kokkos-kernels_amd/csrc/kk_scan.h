// kk_scan.h -- workgroup-level and device-level exclusive prefix sums (K11 analogue:
// kk_exclusive_parallel_prefix_sum, common/src/KokkosKernels_SimpleUtils.hpp:86-135).
#pragma once
#include "kk_common.h"

namespace kk {

// Exclusive scan of one value per work-item across an NT-thread workgroup (NT/64 waves).
// s_wave: NT/64 shared slots.  Every thread of the workgroup must call it.  *total = workgroup sum.
template <class T, int NT> __device__ __forceinline__ T block_exclusive_scan_n(T v, T* total, T* s_wave) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  T inc = v;
  for (int o = 1; o < 64; o <<= 1) {
    const T nb = __shfl_up(inc, (unsigned)o, 64);
    if (lane >= o) inc += nb;
  }
  __syncthreads();              // s_wave may still be read from a previous call
  if (lane == 63) s_wave[w] = inc;
  __syncthreads();
  T base = T(0), tot = T(0);
  for (int i = 0; i < NT / 64; ++i) { const T sv = s_wave[i]; if (i < w) base += sv; tot += sv; }
  *total = tot;
  return base + inc - v;
}
template <class T> __device__ __forceinline__ T block_exclusive_scan(T v, T* total, T* s_wave) {
  return block_exclusive_scan_n<T, kBlock>(v, total, s_wave);
}

constexpr int kScanItems = 8;                      // consecutive items per thread
constexpr int kScanTile  = kBlock * kScanItems;    // 2048 items per workgroup

template <class T>
__global__ __launch_bounds__(kBlock) void scan_reduce_kernel(const T* __restrict__ d, int64_t n, T* __restrict__ sums) {
  __shared__ T s_wave[kBlock / 64];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  T local = T(0);
  for (int q = 0; q < kScanItems; ++q) if (base + q < n) local += d[base + q];
  T tot;
  (void)block_exclusive_scan<T>(local, &tot, s_wave);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

template <class T>
__global__ __launch_bounds__(kBlock) void scan_apply_kernel(T* __restrict__ d, int64_t n, const T* __restrict__ sums) {
  __shared__ T s_wave[kBlock / 64];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  T item[kScanItems];
  T local = T(0);
  for (int q = 0; q < kScanItems; ++q) { item[q] = (base + q < n) ? d[base + q] : T(0); local += item[q]; }
  T tot;
  T run = block_exclusive_scan<T>(local, &tot, s_wave) + (sums ? sums[blockIdx.x] : T(0));
  for (int q = 0; q < kScanItems; ++q) {
    if (base + q < n) d[base + q] = run;
    run += item[q];
  }
}

// in-place exclusive scan of d[0..n): d[i] := sum of the old d[0..i).
// ws: optional workspace of at least scan_workspace_items(n) items -- the scan then allocates nothing, frees nothing (a hipFree waits for the
// whole device) and does not synchronise the stream.
inline int64_t scan_workspace_items(int64_t n) { int64_t tot = 0; while (n > kScanTile) { n = ceil_div(n, kScanTile); tot += n; } return tot + 1; }
template <class T> static int exclusive_scan_inplace(T* d, int64_t n, hipStream_t st, T* ws = nullptr) {
  if (n <= 0) return KKAMD_OK;
  const int64_t nb = ceil_div(n, kScanTile);
  if (nb == 1) {
    KK_LAUNCH((scan_apply_kernel<T>), 1u, kBlock, 0, st, d, n, (const T*)nullptr);
    KK_LAUNCH_CHECK();
    return KKAMD_OK;
  }
  if (ws) {
    KK_LAUNCH((scan_reduce_kernel<T>), (unsigned)nb, kBlock, 0, st, (const T*)d, n, ws);
    const int rc = exclusive_scan_inplace<T>(ws, nb, st, ws + nb);
    if (rc) return rc;
    KK_LAUNCH((scan_apply_kernel<T>), (unsigned)nb, kBlock, 0, st, d, n, (const T*)ws);
    KK_LAUNCH_CHECK();
    return KKAMD_OK;
  }
  T* sums = nullptr;
  KK_HIP(hipMalloc((void**)&sums, sizeof(T) * (size_t)nb));
  KK_LAUNCH((scan_reduce_kernel<T>), (unsigned)nb, kBlock, 0, st, (const T*)d, n, sums);
  int rc = exclusive_scan_inplace<T>(sums, nb, st);
  if (rc == KKAMD_OK) {
    KK_LAUNCH((scan_apply_kernel<T>), (unsigned)nb, kBlock, 0, st, d, n, (const T*)sums);
  }
  hipError_t e1 = hipGetLastError();
  hipError_t e2 = hipStreamSynchronize(st);
  (void)hipFree(sums);
  if (rc) return rc;
  if (e1 != hipSuccess || e2 != hipSuccess) return fail(KKAMD_ERR_HIP, "exclusive_scan: kernel failed");
  return KKAMD_OK;
}

}  // namespace kk
