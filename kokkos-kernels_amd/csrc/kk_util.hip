// kk_util.hip -- helpers either side of the hot path, all on the device:
//   exclusive prefix sum   (kk_exclusive_parallel_prefix_sum, common/src/KokkosKernels_SimpleUtils.hpp:86-135)
//   per-row CRS sort       (sort_crs_matrix, sparse/src/KokkosSparse_SortCrs.hpp:43-120)
//   structured Laplacians  (test_common/KokkosKernels_Test_Structured_Matrix.hpp, every BC = 1),
//                          generated straight into HBM so the 300^3 / 600^3 benchmark inputs never
//                          cross PCIe.  Bit-identical to the reference's generator including its
//                          quirks Q1-Q6 (documented in oracle/kk_oracle.c, pinned by tests/golden).
#include "kk_common.h"
#include "kk_scan.h"
#include <climits>
#include <utility>

namespace kk {

// ------------------------------------------------------------------------------------------------
// structured Laplacians
struct Grid { int64_t nx, ny, nz; int dim; int stencil; };

__host__ __device__ __forceinline__ int row_len(const Grid& g, int64_t i, int64_t j, int64_t k) {
  const int ci = 1 + (i > 0) + (i < g.nx - 1);
  const int cj = 1 + (j > 0) + (j < g.ny - 1);
  const int ck = (g.dim == 3) ? 1 + (k > 0) + (k < g.nz - 1) : 1;
  if (g.stencil) return ci * cj * ck;
  return 1 + (ci - 1) + (cj - 1) + (ck - 1);
}

template <class OffT>
__global__ void laplace_len_kernel(Grid g, int64_t row_begin, int64_t nrows, OffT* __restrict__ row_map) {
  for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l <= nrows; l += (int64_t)gridDim.x * blockDim.x) {
    if (l == nrows) { row_map[l] = 0; continue; }
    const int64_t r = row_begin + l;
    const int64_t i = r % g.nx, j = (r / g.nx) % g.ny, k = r / (g.nx * g.ny);
    row_map[l] = (OffT)row_len(g, i, j, k);
  }
}

template <class OffT, class VT>
__global__ void laplace_fill_kernel(Grid g, int64_t row_begin, int64_t nrows, const OffT* __restrict__ row_map,
                                    int32_t* __restrict__ entries, VT* __restrict__ values) {
  // local row l of the slab is global grid row r = row_begin + l; row_map is local (starts at 0),
  // column indices stay global (the 1-D row partition of the multi-GPU SpMV keeps global columns)
  const int64_t pl = g.nx * g.ny;
  for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < nrows; l += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = row_begin + l;
    const int64_t i = r % g.nx, j = (r / g.nx) % g.ny, k = r / pl;
    const bool boundary = (i == 0 || i == g.nx - 1 || j == 0 || j == g.ny - 1 ||
                           (g.dim == 3 && (k == 0 || k == g.nz - 1)));
    const int64_t start = (int64_t)row_map[l];
    int64_t p = start;
    const int klo = (g.dim == 3) ? -1 : 0, khi = (g.dim == 3) ? 1 : 0;
    for (int dk = klo; dk <= khi; ++dk)
      for (int dj = -1; dj <= 1; ++dj)
        for (int di = -1; di <= 1; ++di) {
          const int noff = (di != 0) + (dj != 0) + (dk != 0);
          if (!g.stencil && noff > 1) continue;
          const int64_t ii = i + di, jj = j + dj, kk2 = k + dk;
          if (ii < 0 || ii >= g.nx || jj < 0 || jj >= g.ny || kk2 < 0 || kk2 >= g.nz) continue;
          entries[p] = (int32_t)((kk2 * g.ny + jj) * g.nx + ii);
          double v;
          if (boundary) v = (noff == 0) ? 1.0 : 0.0;
          else if (g.dim == 2) v = g.stencil ? (noff == 0 ? 16.0 : -2.0) : (noff == 0 ? 4.0 : -1.0);
          else if (!g.stencil) v = (noff == 0) ? 6.0 : -1.0;
          else v = (noff == 0) ? 32.0 : (noff == 1) ? 0.0 : (noff == 2) ? -2.0 : -1.0;
          values[p++] = (VT)v;
        }
    // the reference generator's quirks (see oracle/kk_oracle.c for the citations)
    if (g.dim == 2) {
      if (!g.stencil && j == 0 && i == g.nx - 1)                                   // Q1
        for (int64_t q = start; q < p; ++q) values[q] = (VT)0;
    } else if (g.stencil) {
      if (j == g.ny - 1 && i > 0 && i < g.nx - 1 && k > 0 && k < g.nz - 1) entries[start + 2] = entries[start + 3];  // Q2
      if (k == 0 && i == g.nx - 1 && j > 0 && j < g.ny - 1) { values[start + 3] = (VT)0; values[start + 9] = (VT)1; }  // Q3
      if (i == 0 && k == g.nz - 1 && j > 0 && j < g.ny - 1) {                                                          // Q6
        const int64_t o[12] = {-pl - 1, -pl, -pl + 1, -pl + g.nx - 1, -pl + g.nx, -pl + g.nx + 1, -1, 0, 1, g.nx - 1, g.nx, g.nx + 1};
        for (int q = 0; q < 12; ++q) { entries[start + q] = (int32_t)(r + o[q]); values[start + q] = (VT)(q == 7 ? 1 : 0); }
      }
    } else {
      if (i == 0 && k == g.nz - 1 && j > 0 && j < g.ny - 1) entries[start + 1] = (int32_t)(r - 1);                     // Q4
      if (j == g.ny - 1 && k == 0 && i > 0 && i < g.nx - 1) {                                                          // Q5
        const int64_t o[5] = {-1, 0, 1, g.nx, pl};
        for (int q = 0; q < 5; ++q) { entries[start + q] = (int32_t)(r + o[q]); values[start + q] = (VT)(q == 1 ? 1 : 0); }
      }
    }
  }
}

template <class OffT>
static int gen_laplace_typed(const Grid& g, int64_t row_begin, int64_t nrows, void* d_row_map, int32_t* d_entries,
                             void* d_values, int value_type, int64_t* nnz, hipStream_t st) {
  OffT* rm          = (OffT*)d_row_map;
  const int64_t nb  = ceil_div(nrows + 1, kBlock);
  const unsigned gr = (unsigned)(nb < 65536 ? nb : 65536);
  KK_LAUNCH((laplace_len_kernel<OffT>), gr, kBlock, 0, st, g, row_begin, nrows, rm);
  KK_LAUNCH_CHECK();
  int rc = exclusive_scan_inplace<OffT>(rm, nrows + 1, st);
  if (rc) return rc;
  OffT total = 0;
  KK_HIP(hipMemcpyAsync(&total, rm + nrows, sizeof(OffT), hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  if (nnz) *nnz = (int64_t)total;
  if (!d_entries) return KKAMD_OK;
  if (!d_values) return fail(KKAMD_ERR_INVALID_ARG, "kkamd_gen_laplace: entries given without values");
  if (value_type == KKAMD_F64) {
    KK_LAUNCH((laplace_fill_kernel<OffT, double>), gr, kBlock, 0, st, g, row_begin, nrows, (const OffT*)rm, d_entries, (double*)d_values);
  } else {
    KK_LAUNCH((laplace_fill_kernel<OffT, float>), gr, kBlock, 0, st, g, row_begin, nrows, (const OffT*)rm, d_entries, (float*)d_values);
  }
  KK_LAUNCH_CHECK();
  return KKAMD_OK;
}

// ------------------------------------------------------------------------------------------------
// per-row sort of (entries, values) -- sort_crs_matrix, sparse/src/KokkosSparse_SortCrs.hpp:43-120.
// A segment of up to 8192 entries is sorted by one workgroup in LDS: bitonic network over (key, original index) pairs,
// the index breaking ties, so the sort is STABLE (values of duplicate columns keep their order, like the oracle).
// Rows up to 8192 entries are one segment.  Longer rows (hub rows of a transposed R-MAT matrix reach 5e5) are sorted in
// 8192-entry chunks and then merged pairwise, log2(len/8192) passes of a merge-path kernel over a ping-pong buffer.
constexpr int kSortCap = 8192;
constexpr int kMergePerThread = 8;

template <class VT, bool HAS_VAL>
__device__ __forceinline__ void sort_segment_lds(int64_t s, int len, int32_t* __restrict__ entries, VT* __restrict__ values,
                                                 int* s_key, int* s_idx) {
  const int t = threadIdx.x;
  int n2 = 1;
  while (n2 < len) n2 <<= 1;
  __syncthreads();
  for (int i = t; i < n2; i += kBlock) { s_key[i] = (i < len) ? entries[s + i] : INT_MAX; s_idx[i] = i; }
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < n2; i += kBlock) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const int ka = s_key[i], kb = s_key[ixj], ia = s_idx[i], ib = s_idx[ixj];
          const bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);   // index tie-break keeps it stable
          const bool up     = ((i & k) == 0);
          if (a_gt_b == up) { s_key[i] = kb; s_key[ixj] = ka; s_idx[i] = ib; s_idx[ixj] = ia; }
        }
      }
      __syncthreads();
    }
  // permute values through registers: read all, barrier, write
  VT tmp[kSortCap / kBlock];
  if (HAS_VAL) {
    KK_UNROLL
    for (int q = 0; q < kSortCap / kBlock; ++q) { const int i = t + q * kBlock; if (i < len) tmp[q] = values[s + s_idx[i]]; }
  }
  __syncthreads();
  KK_UNROLL
  for (int q = 0; q < kSortCap / kBlock; ++q) {
    const int i = t + q * kBlock;
    if (i < len) { entries[s + i] = s_key[i]; if (HAS_VAL) values[s + i] = tmp[q]; }
  }
}

template <class OffT, class VT, bool HAS_VAL>
__global__ __launch_bounds__(kBlock) void sort_rows_kernel(int64_t nrows, const OffT* __restrict__ row_map,
                                                           int32_t* __restrict__ entries, VT* __restrict__ values) {
  __shared__ int s_key[kSortCap];
  __shared__ int s_idx[kSortCap];
  for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
    const int64_t s   = (int64_t)row_map[r];
    const int64_t len = (int64_t)row_map[r + 1] - s;
    if (len < 2 || len > kSortCap) continue;                 // long rows: sort_long_chunks_kernel + merge passes
    sort_segment_lds<VT, HAS_VAL>(s, (int)len, entries, values, s_key, s_idx);
  }
}

// rows longer than kSortCap, compacted (order irrelevant)
template <class OffT>
__global__ __launch_bounds__(kBlock) void long_rows_kernel(int64_t nrows, const OffT* __restrict__ row_map, int32_t* __restrict__ list,
                                                           unsigned long long* __restrict__ count) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int lane  = threadIdx.x & 63;
  const bool is_long = r < nrows && (int64_t)row_map[r + 1] - (int64_t)row_map[r] > kSortCap;
  const unsigned long long m = __ballot(is_long);
  unsigned long long start = 0;
  if (lane == 0 && m) start = atomicAdd(count, (unsigned long long)__popcll(m));
  start = __shfl(start, 0, 64);
  if (is_long) list[start + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)r;
}

template <class OffT, class VT, bool HAS_VAL>
__global__ __launch_bounds__(kBlock) void sort_long_chunks_kernel(const int32_t* __restrict__ list, const OffT* __restrict__ row_map,
                                                                  int32_t* __restrict__ entries, VT* __restrict__ values) {
  __shared__ int s_key[kSortCap];
  __shared__ int s_idx[kSortCap];
  const int64_t r = list[blockIdx.y];
  const int64_t s = (int64_t)row_map[r], len = (int64_t)row_map[r + 1] - s;
  for (int64_t c0 = (int64_t)blockIdx.x * kSortCap; c0 < len; c0 += (int64_t)gridDim.x * kSortCap) {
    const int n = (int)(len - c0 < kSortCap ? len - c0 : kSortCap);
    if (n > 1) sort_segment_lds<VT, HAS_VAL>(s + c0, n, entries, values, s_key, s_idx);
  }
}

// one merge pass over every long row: runs of w sorted entries -> runs of 2w.  Each work-item produces kMergePerThread
// consecutive outputs: a merge-path binary search (ties go to the left run: stable) finds where they start in the two
// input runs, then it merges sequentially.
template <class OffT, class VT, bool HAS_VAL>
__global__ __launch_bounds__(kBlock) void merge_pass_kernel(const int32_t* __restrict__ list, const OffT* __restrict__ row_map,
                                                            const int32_t* __restrict__ src_e, const VT* __restrict__ src_v,
                                                            int32_t* __restrict__ dst_e, VT* __restrict__ dst_v, int64_t w) {
  const int64_t r = list[blockIdx.y];
  const int64_t s = (int64_t)row_map[r], len = (int64_t)row_map[r + 1] - s;
  for (int64_t o0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kMergePerThread; o0 < len;
       o0 += (int64_t)gridDim.x * kBlock * kMergePerThread) {
    const int64_t pbase = (o0 / (2 * w)) * (2 * w);
    const int64_t a_n = (len - pbase < w) ? len - pbase : w;
    const int64_t b_lo = pbase + w;
    const int64_t b_n = (len - b_lo <= 0) ? 0 : ((len - b_lo < w) ? len - b_lo : w);
    const int32_t* A = src_e + s + pbase;
    const int32_t* B = src_e + s + b_lo;
    const int64_t d  = o0 - pbase;
    int64_t lo = d - b_n > 0 ? d - b_n : 0, hi = d < a_n ? d : a_n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (A[mid] <= B[d - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    int64_t ia = lo, ib = d - lo;
    const int64_t o_end = (o0 + kMergePerThread < pbase + a_n + b_n) ? o0 + kMergePerThread : pbase + a_n + b_n;
    for (int64_t o = o0; o < o_end; ++o) {
      const bool take_a = ib >= b_n || (ia < a_n && A[ia] <= B[ib]);
      const int64_t src = take_a ? pbase + ia : b_lo + ib;
      dst_e[s + o] = src_e[s + src];
      if (HAS_VAL) dst_v[s + o] = src_v[s + src];
      if (take_a) ++ia; else ++ib;
    }
  }
}

template <class OffT, class VT, bool HAS_VAL>
__global__ __launch_bounds__(kBlock) void copy_long_rows_kernel(const int32_t* __restrict__ list, const OffT* __restrict__ row_map,
                                                                const int32_t* __restrict__ src_e, const VT* __restrict__ src_v,
                                                                int32_t* __restrict__ dst_e, VT* __restrict__ dst_v) {
  const int64_t r = list[blockIdx.y];
  const int64_t s = (int64_t)row_map[r], len = (int64_t)row_map[r + 1] - s;
  for (int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x; o < len; o += (int64_t)gridDim.x * kBlock) {
    dst_e[s + o] = src_e[s + o];
    if (HAS_VAL) dst_v[s + o] = src_v[s + o];
  }
}

template <class OffT> __global__ void max_row_len_kernel(int64_t nrows, const OffT* __restrict__ row_map, int* __restrict__ out) {
  int mx = 0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t l = (int64_t)row_map[r + 1] - (int64_t)row_map[r];
    const int li    = l > INT_MAX ? INT_MAX : (int)l;
    mx = li > mx ? li : mx;
  }
  for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(mx, o, 64); mx = other > mx ? other : mx; }
  if ((threadIdx.x & 63) == 0) atomicMax(out, mx);
}

template <class OffT, class VT, bool HAS_VAL>
static int sort_vt(int64_t nrows, const OffT* rm, int32_t* d_entries, VT* d_values, hipStream_t st) {
  DevBuf max_b;                                                // temporaries free themselves on every early return
  KK_HIP(max_b.alloc(sizeof(int)));
  int* d_max = max_b.as<int>();
  KK_HIP(hipMemsetAsync(d_max, 0, sizeof(int), st));
  const int64_t nb = ceil_div(nrows, kBlock);
  KK_LAUNCH((max_row_len_kernel<OffT>), (unsigned)(nb < 4096 ? nb : 4096), kBlock, 0, st, nrows, rm, d_max);
  int h_max = 0;
  KK_HIP(hipMemcpyAsync(&h_max, d_max, sizeof(int), hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  max_b.reset();
  KK_LAUNCH((sort_rows_kernel<OffT, VT, HAS_VAL>), (unsigned)(nrows < 65536 ? nrows : 65536), kBlock, 0, st, nrows, rm, d_entries, d_values);
  KK_LAUNCH_CHECK();
  if (h_max <= kSortCap) return KKAMD_OK;
  // long rows
  DevBuf list_b, cnt_b;
  unsigned long long h_cnt = 0;
  OffT h_nnz = 0;
  KK_HIP(list_b.alloc(sizeof(int32_t) * (size_t)nrows));
  KK_HIP(cnt_b.alloc(sizeof(unsigned long long)));
  int32_t* d_list = list_b.as<int32_t>(); unsigned long long* d_cnt = cnt_b.as<unsigned long long>();
  KK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), st));
  KK_LAUNCH((long_rows_kernel<OffT>), (unsigned)nb, kBlock, 0, st, nrows, rm, d_list, d_cnt);
  KK_HIP(hipMemcpyAsync(&h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost, st));
  KK_HIP(hipMemcpyAsync(&h_nnz, rm + nrows, sizeof(OffT), hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  cnt_b.reset();
  const unsigned L = (unsigned)h_cnt;
  int rc = KKAMD_OK;
  int32_t* t_e = nullptr; VT* t_v = nullptr;
  if (L > 65535) rc = fail(KKAMD_ERR_UNSUPPORTED, "kkamd_sort_crs: more than 65535 rows longer than %d entries", kSortCap);
  if (rc == KKAMD_OK && hipMalloc((void**)&t_e, sizeof(int32_t) * (size_t)h_nnz) != hipSuccess) rc = fail(KKAMD_ERR_ALLOC, "kkamd_sort_crs: out of device memory");
  if (rc == KKAMD_OK && HAS_VAL && hipMalloc((void**)&t_v, sizeof(VT) * (size_t)h_nnz) != hipSuccess) rc = fail(KKAMD_ERR_ALLOC, "kkamd_sort_crs: out of device memory");
  if (rc == KKAMD_OK) {
    const unsigned chunks = (unsigned)ceil_div(h_max, kSortCap);
    KK_LAUNCH((sort_long_chunks_kernel<OffT, VT, HAS_VAL>), dim3(chunks < 1024 ? chunks : 1024, L), kBlock, 0, st, (const int32_t*)d_list,
              rm, d_entries, d_values);
    const int64_t tiles = ceil_div(h_max, (int64_t)kBlock * kMergePerThread);
    const unsigned gx   = (unsigned)(tiles < 4096 ? tiles : 4096);
    int32_t *se = d_entries, *de = t_e;
    VT *sv = d_values, *dv = t_v;
    for (int64_t w = kSortCap; w < h_max; w *= 2) {
      KK_LAUNCH((merge_pass_kernel<OffT, VT, HAS_VAL>), dim3(gx, L), kBlock, 0, st, (const int32_t*)d_list, rm, (const int32_t*)se,
                (const VT*)sv, de, dv, w);
      std::swap(se, de); std::swap(sv, dv);
    }
    if (se != d_entries)
      KK_LAUNCH((copy_long_rows_kernel<OffT, VT, HAS_VAL>), dim3(gx, L), kBlock, 0, st, (const int32_t*)d_list, rm, (const int32_t*)se,
                (const VT*)sv, d_entries, d_values);
    hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(st);
    if (e1 != hipSuccess || e2 != hipSuccess) rc = fail(KKAMD_ERR_HIP, "kkamd_sort_crs: long-row merge failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  }
  if (t_e) (void)hipFree(t_e);
  if (t_v) (void)hipFree(t_v);
  list_b.reset();
  return rc;
}

template <class OffT>
static int sort_typed(int64_t nrows, const void* d_row_map, int32_t* d_entries, void* d_values, int value_type, hipStream_t st) {
  const OffT* rm = (const OffT*)d_row_map;
  if (!d_values) return sort_vt<OffT, float, false>(nrows, rm, d_entries, (float*)nullptr, st);
  if (value_type == KKAMD_F64) return sort_vt<OffT, double, true>(nrows, rm, d_entries, (double*)d_values, st);
  return sort_vt<OffT, float, true>(nrows, rm, d_entries, (float*)d_values, st);
}

// ------------------------------------------------------------------------------------------------
// sort_and_merge_matrix (sparse/src/KokkosSparse_SortCrs.hpp:304-363, MergedRowmapFunctor / MatrixMergedEntriesFunctor in
// sparse/impl/KokkosSparse_sort_crs_impl.hpp:120-216): after the row sort, runs of equal columns collapse into one entry
// whose value is the sum of the run taken left to right.
template <class OffT>
__global__ __launch_bounds__(kBlock) void merged_count_kernel(int64_t nrows, const OffT* __restrict__ rm, const int32_t* __restrict__ ent,
                                                              OffT* __restrict__ out_rm) {
  const int lane = threadIdx.x & 7;
  const int64_t stride = (int64_t)gridDim.x * (kBlock / 8);
  for (int64_t r0 = (int64_t)blockIdx.x * (kBlock / 8); r0 <= nrows; r0 += stride) {      // workgroup-uniform trip count
    const int64_t r = r0 + threadIdx.x / 8;
    int cnt = 0;
    if (r < nrows) {
      const int64_t s = (int64_t)rm[r], e = (int64_t)rm[r + 1];
      for (int64_t j = s + lane; j < e; j += 8) cnt += (j == s || ent[j] != ent[j - 1]) ? 1 : 0;
    }
    cnt = group_sum(cnt, 8);
    if (r <= nrows && lane == 0) out_rm[r] = (OffT)(r < nrows ? cnt : 0);
  }
}
template <class OffT, class VT>
__global__ __launch_bounds__(kBlock) void merged_fill_kernel(int64_t nrows, const OffT* __restrict__ rm, const int32_t* __restrict__ ent,
                                                             const VT* __restrict__ val, const OffT* __restrict__ out_rm,
                                                             int32_t* __restrict__ out_ent, VT* __restrict__ out_val) {
  // one work-item per input entry that STARTS a run: it sums its run (runs are short) and knows its output slot from
  // the number of run starts before it in the row, counted by the 8 lanes of the row in chunks
  const int lane = threadIdx.x & 7;
  const int64_t stride = (int64_t)gridDim.x * (kBlock / 8);
  for (int64_t r0 = (int64_t)blockIdx.x * (kBlock / 8); r0 < nrows; r0 += stride) {
    const int64_t r = r0 + threadIdx.x / 8;
    const int64_t s = r < nrows ? (int64_t)rm[r] : 0, e = r < nrows ? (int64_t)rm[r + 1] : 0;
    int64_t pos = r < nrows ? (int64_t)out_rm[r] : 0;
    // all 8-lane groups of the wave iterate as long as the longest row of the wave needs (uniform ballots)
    int64_t wave_span = e - s;
    for (int o = 32; o >= 8; o >>= 1) { const int64_t other = __shfl_xor(wave_span, o, 64); wave_span = other > wave_span ? other : wave_span; }
    for (int64_t j0 = 0; j0 < wave_span; j0 += 8) {
      const int64_t j   = s + j0 + lane;
      const bool start  = j < e && (j == s || ent[j] != ent[j - 1]);
      const unsigned long long m = __ballot(start);
      const unsigned long long mine = (m >> ((threadIdx.x & 63) & ~7)) & 0xffull;
      if (start) {
        const int64_t slot = pos + __popcll(mine & ((1ull << lane) - 1ull));
        VT acc = val ? val[j] : VT(0);
        int64_t q = j + 1;
        while (q < e && ent[q] == ent[j]) { if (val) acc += val[q]; ++q; }
        out_ent[slot] = ent[j];
        if (val) out_val[slot] = acc;
      }
      pos += __popcll(mine);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// transpose_matrix (sparse/src/KokkosSparse_Utils.hpp:338-400: count per column with atomics, prefix sum, fill through
// atomic cursors).  The fill order inside a transposed row is whatever the atomics produce -- as in the reference --
// so the rows are sorted afterwards and the result is deterministic for matrices without duplicate entries.
__device__ __forceinline__ int32_t fetch_inc(int32_t* p) { return atomicAdd(p, 1); }
__device__ __forceinline__ int64_t fetch_inc(int64_t* p) { return (int64_t)atomicAdd(reinterpret_cast<unsigned long long*>(p), 1ull); }
template <class OffT>
__global__ __launch_bounds__(kBlock) void transpose_count_kernel(int64_t nnz, const int32_t* __restrict__ ent, OffT* __restrict__ t_rm) {
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * kBlock) (void)fetch_inc(&t_rm[ent[j]]);
}
template <class OffT, class VT>
__global__ __launch_bounds__(kBlock) void transpose_fill_kernel(int64_t nrows, const OffT* __restrict__ rm, const int32_t* __restrict__ ent,
                                                                const VT* __restrict__ val, OffT* __restrict__ cursor,
                                                                int32_t* __restrict__ t_ent, VT* __restrict__ t_val) {
  const int lane = threadIdx.x & 7;
  for (int64_t r = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / 8; r < nrows; r += (int64_t)gridDim.x * (kBlock / 8))
    for (int64_t j = (int64_t)rm[r] + lane; j < (int64_t)rm[r + 1]; j += 8) {
      const OffT p = fetch_inc(&cursor[ent[j]]);
      t_ent[p] = (int32_t)r;
      if (val) t_val[p] = val[j];
    }
}

}  // namespace kk

extern "C" {

int kkamd_exclusive_scan(void* d_data, int64_t n, int offset_type, kkamd_stream_t stream) {
  if (n < 0) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_exclusive_scan: negative length");
  if (n == 0) return KKAMD_OK;
  if (!d_data) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_exclusive_scan: null data");
  if (offset_type == KKAMD_I32) return kk::exclusive_scan_inplace<int32_t>((int32_t*)d_data, n, kk::to_hip(stream));
  if (offset_type == KKAMD_I64) return kk::exclusive_scan_inplace<int64_t>((int64_t*)d_data, n, kk::to_hip(stream));
  return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_exclusive_scan: unknown offset_type %d", offset_type);
}

int kkamd_sort_crs(int64_t num_rows, const void* d_row_map, int32_t* d_entries, void* d_values, int offset_type,
                   int value_type, kkamd_stream_t stream) {
  if (num_rows < 0) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_sort_crs: negative row count");
  if (num_rows == 0) return KKAMD_OK;
  if (!d_row_map || !d_entries) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_sort_crs: null pointer");
  if (offset_type == KKAMD_I32) return kk::sort_typed<int32_t>(num_rows, d_row_map, d_entries, d_values, value_type, kk::to_hip(stream));
  if (offset_type == KKAMD_I64) return kk::sort_typed<int64_t>(num_rows, d_row_map, d_entries, d_values, value_type, kk::to_hip(stream));
  return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_sort_crs: unknown offset_type %d", offset_type);
}

int kkamd_sort_and_merge(int64_t num_rows, const void* d_row_map, int32_t* d_entries, void* d_values, int offset_type,
                         int value_type, void* d_row_map_out, int32_t* d_entries_out, void* d_values_out, int64_t* nnz_out,
                         kkamd_stream_t stream) {
  if (num_rows < 0) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_sort_and_merge: negative row count");
  if (offset_type != KKAMD_I32 && offset_type != KKAMD_I64) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_sort_and_merge: unknown offset_type %d", offset_type);
  if (!d_row_map_out || !nnz_out) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_sort_and_merge: null output");
  hipStream_t st = kk::to_hip(stream);
  const size_t osz = offset_type == KKAMD_I64 ? 8 : 4;
  if (num_rows == 0) { KK_HIP(hipMemsetAsync(d_row_map_out, 0, osz, st)); *nnz_out = 0; return KKAMD_OK; }
  if (!d_row_map) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_sort_and_merge: null row_map");
  const int64_t nbk  = kk::ceil_div(num_rows + 1, kk::kBlock / 8);
  const unsigned grid = (unsigned)(nbk < 65536 ? nbk : 65536);
  if (!d_entries_out) {
    // phase 1: sort in place, merged row_map and nnz
    if (!d_entries) { KK_HIP(hipMemsetAsync(d_row_map_out, 0, osz * (size_t)(num_rows + 1), st)); *nnz_out = 0; return KKAMD_OK; }
    int rc = kkamd_sort_crs(num_rows, d_row_map, d_entries, d_values, offset_type, value_type, stream);
    if (rc) return rc;
    if (offset_type == KKAMD_I64) {
      KK_LAUNCH((kk::merged_count_kernel<int64_t>), grid, kk::kBlock, 0, st, num_rows, (const int64_t*)d_row_map, (const int32_t*)d_entries, (int64_t*)d_row_map_out);
      if ((rc = kk::exclusive_scan_inplace<int64_t>((int64_t*)d_row_map_out, num_rows + 1, st))) return rc;
      int64_t total = 0;
      KK_HIP(hipMemcpyAsync(&total, (int64_t*)d_row_map_out + num_rows, 8, hipMemcpyDeviceToHost, st));
      KK_HIP(hipStreamSynchronize(st));
      *nnz_out = total;
    } else {
      KK_LAUNCH((kk::merged_count_kernel<int32_t>), grid, kk::kBlock, 0, st, num_rows, (const int32_t*)d_row_map, (const int32_t*)d_entries, (int32_t*)d_row_map_out);
      if ((rc = kk::exclusive_scan_inplace<int32_t>((int32_t*)d_row_map_out, num_rows + 1, st))) return rc;
      int32_t total = 0;
      KK_HIP(hipMemcpyAsync(&total, (int32_t*)d_row_map_out + num_rows, 4, hipMemcpyDeviceToHost, st));
      KK_HIP(hipStreamSynchronize(st));
      *nnz_out = total;
    }
    return KKAMD_OK;
  }
  // phase 2: fill (inputs already sorted by phase 1)
  if (*nnz_out == 0) return KKAMD_OK;
  if (offset_type == KKAMD_I64) {
    if (!d_values) KK_LAUNCH((kk::merged_fill_kernel<int64_t, float>), grid, kk::kBlock, 0, st, num_rows, (const int64_t*)d_row_map, (const int32_t*)d_entries, (const float*)nullptr, (const int64_t*)d_row_map_out, d_entries_out, (float*)nullptr);
    else if (value_type == KKAMD_F64) KK_LAUNCH((kk::merged_fill_kernel<int64_t, double>), grid, kk::kBlock, 0, st, num_rows, (const int64_t*)d_row_map, (const int32_t*)d_entries, (const double*)d_values, (const int64_t*)d_row_map_out, d_entries_out, (double*)d_values_out);
    else KK_LAUNCH((kk::merged_fill_kernel<int64_t, float>), grid, kk::kBlock, 0, st, num_rows, (const int64_t*)d_row_map, (const int32_t*)d_entries, (const float*)d_values, (const int64_t*)d_row_map_out, d_entries_out, (float*)d_values_out);
  } else {
    if (!d_values) KK_LAUNCH((kk::merged_fill_kernel<int32_t, float>), grid, kk::kBlock, 0, st, num_rows, (const int32_t*)d_row_map, (const int32_t*)d_entries, (const float*)nullptr, (const int32_t*)d_row_map_out, d_entries_out, (float*)nullptr);
    else if (value_type == KKAMD_F64) KK_LAUNCH((kk::merged_fill_kernel<int32_t, double>), grid, kk::kBlock, 0, st, num_rows, (const int32_t*)d_row_map, (const int32_t*)d_entries, (const double*)d_values, (const int32_t*)d_row_map_out, d_entries_out, (double*)d_values_out);
    else KK_LAUNCH((kk::merged_fill_kernel<int32_t, float>), grid, kk::kBlock, 0, st, num_rows, (const int32_t*)d_row_map, (const int32_t*)d_entries, (const float*)d_values, (const int32_t*)d_row_map_out, d_entries_out, (float*)d_values_out);
  }
  KK_LAUNCH_CHECK();
  return KKAMD_OK;
}

int kkamd_transpose(int64_t num_rows, int64_t num_cols, int64_t nnz, const void* d_row_map, const int32_t* d_entries,
                    const void* d_values, int offset_type, int value_type, void* d_t_row_map, int32_t* d_t_entries,
                    void* d_t_values, kkamd_stream_t stream) {
  if (num_rows < 0 || num_cols < 0 || nnz < 0) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_transpose: negative dimension");
  if (offset_type != KKAMD_I32 && offset_type != KKAMD_I64) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_transpose: unknown offset_type %d", offset_type);
  if (!d_t_row_map) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_transpose: null output row_map");
  hipStream_t st   = kk::to_hip(stream);
  const size_t osz = offset_type == KKAMD_I64 ? 8 : 4;
  KK_HIP(hipMemsetAsync(d_t_row_map, 0, osz * (size_t)(num_cols + 1), st));
  if (nnz == 0 || num_rows == 0 || num_cols == 0) return KKAMD_OK;
  if (!d_row_map || !d_entries || !d_t_entries || (d_values && !d_t_values)) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_transpose: null pointer");
  const int64_t nb1 = kk::ceil_div(nnz, kk::kBlock), nb2 = kk::ceil_div(num_rows * 8, kk::kBlock);
  const unsigned g1 = (unsigned)(nb1 < 65536 ? nb1 : 65536), g2 = (unsigned)(nb2 < 65536 ? nb2 : 65536);
  kk::DevBuf cursor_b;                       // frees itself on every return
  KK_HIP(cursor_b.alloc(osz * (size_t)(num_cols + 1)));
  void* d_cursor = cursor_b.p;
  int rc = KKAMD_OK;
#define KK_TR(OT)                                                                                                              \
  do {                                                                                                                         \
    KK_LAUNCH((kk::transpose_count_kernel<OT>), g1, kk::kBlock, 0, st, nnz, d_entries, (OT*)d_t_row_map);                       \
    rc = kk::exclusive_scan_inplace<OT>((OT*)d_t_row_map, num_cols + 1, st);                                                   \
    if (rc == KKAMD_OK) {                                                                                                      \
      (void)hipMemcpyAsync(d_cursor, d_t_row_map, osz * (size_t)(num_cols + 1), hipMemcpyDeviceToDevice, st);                  \
      if (!d_values) KK_LAUNCH((kk::transpose_fill_kernel<OT, float>), g2, kk::kBlock, 0, st, num_rows, (const OT*)d_row_map, d_entries, (const float*)nullptr, (OT*)d_cursor, d_t_entries, (float*)nullptr); \
      else if (value_type == KKAMD_F64) KK_LAUNCH((kk::transpose_fill_kernel<OT, double>), g2, kk::kBlock, 0, st, num_rows, (const OT*)d_row_map, d_entries, (const double*)d_values, (OT*)d_cursor, d_t_entries, (double*)d_t_values); \
      else KK_LAUNCH((kk::transpose_fill_kernel<OT, float>), g2, kk::kBlock, 0, st, num_rows, (const OT*)d_row_map, d_entries, (const float*)d_values, (OT*)d_cursor, d_t_entries, (float*)d_t_values); \
    }                                                                                                                          \
  } while (0)
  if (offset_type == KKAMD_I64) KK_TR(int64_t); else KK_TR(int32_t);
#undef KK_TR
  hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(st);
  if (rc) return rc;
  if (e1 != hipSuccess || e2 != hipSuccess) return kk::fail(KKAMD_ERR_HIP, "kkamd_transpose failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  // deterministic order inside every transposed row
  return kkamd_sort_crs(num_cols, d_t_row_map, d_t_entries, d_values ? d_t_values : nullptr, offset_type, value_type, stream);
}

int kkamd_gen_laplace_rows(int dim, int stencil, int64_t nx, int64_t ny, int64_t nz, int64_t row_begin, int64_t row_count,
                           void* d_row_map, int32_t* d_entries, void* d_values, int offset_type, int value_type,
                           int64_t* nnz, kkamd_stream_t stream) {
  if (dim != 2 && dim != 3) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_gen_laplace: dim must be 2 or 3");
  if (stencil != 0 && stencil != 1) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_gen_laplace: stencil must be 0 (FD) or 1 (FE)");
  if (dim == 2) nz = 1;
  if (nx < 2 || ny < 2 || (dim == 3 && (nx < 3 || ny < 3 || nz < 3)))
    return kk::fail(KKAMD_ERR_INVALID_ARG, "You need at least two points per direction to obtain a valid discretization!");
  const int64_t nrows = nx * ny * nz;
  if (nrows > INT32_MAX) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_gen_laplace: grid exceeds the int32 ordinal range");
  if (row_begin < 0 || row_count < 0 || row_begin + row_count > nrows)
    return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_gen_laplace_rows: row range outside the grid");
  if (!d_row_map) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_gen_laplace: null row_map");
  kk::Grid g{nx, ny, nz, dim, stencil};
  if (offset_type == KKAMD_I32) return kk::gen_laplace_typed<int32_t>(g, row_begin, row_count, d_row_map, d_entries, d_values, value_type, nnz, kk::to_hip(stream));
  if (offset_type == KKAMD_I64) return kk::gen_laplace_typed<int64_t>(g, row_begin, row_count, d_row_map, d_entries, d_values, value_type, nnz, kk::to_hip(stream));
  return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_gen_laplace: unknown offset_type %d", offset_type);
}

int kkamd_gen_laplace(int dim, int stencil, int64_t nx, int64_t ny, int64_t nz, void* d_row_map, int32_t* d_entries,
                      void* d_values, int offset_type, int value_type, int64_t* nnz, kkamd_stream_t stream) {
  return kkamd_gen_laplace_rows(dim, stencil, nx, ny, nz, 0, nx * ny * (dim == 2 ? 1 : nz), d_row_map, d_entries, d_values,
                                offset_type, value_type, nnz, stream);
}

}  // extern "C"
