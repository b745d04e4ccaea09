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
// per-row sort of (entries, values): one wave per row, stable rank sort for rows up to 1024 entries
// (rank = #smaller keys + #equal keys with a smaller index), longer rows in 1024-entry LDS passes of
// a block-wide bitonic network.
template <class OffT, class VT, bool HAS_VAL>
__global__ __launch_bounds__(kBlock) void sort_rows_kernel(int64_t nrows, const OffT* __restrict__ row_map,
                                                           int32_t* __restrict__ entries, VT* __restrict__ values,
                                                           int max_len_cap) {
  // workgroup per row, bitonic sort of (key, original index) pairs in LDS, capacity SORT_CAP
  constexpr int SORT_CAP = 8192;
  __shared__ int s_key[SORT_CAP];
  __shared__ int s_idx[SORT_CAP];
  const int t = threadIdx.x;
  for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
    const int64_t s = (int64_t)row_map[r];
    const int len   = (int)((int64_t)row_map[r + 1] - s);
    if (len < 2 || len > max_len_cap) continue;
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
    VT tmp[SORT_CAP / kBlock];
    if (HAS_VAL) {
      KK_UNROLL
      for (int q = 0; q < SORT_CAP / kBlock; ++q) { const int i = t + q * kBlock; if (i < len) tmp[q] = values[s + s_idx[i]]; }
    }
    __syncthreads();
    KK_UNROLL
    for (int q = 0; q < SORT_CAP / kBlock; ++q) {
      const int i = t + q * kBlock;
      if (i < len) { entries[s + i] = s_key[i]; if (HAS_VAL) values[s + i] = tmp[q]; }
    }
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

template <class OffT>
static int sort_typed(int64_t nrows, const void* d_row_map, int32_t* d_entries, void* d_values, int value_type, hipStream_t st) {
  int* d_max = nullptr;
  KK_HIP(hipMalloc((void**)&d_max, sizeof(int)));
  KK_HIP(hipMemsetAsync(d_max, 0, sizeof(int), st));
  const int64_t nb = ceil_div(nrows, kBlock);
  KK_LAUNCH((max_row_len_kernel<OffT>), (unsigned)(nb < 4096 ? nb : 4096), kBlock, 0, st, nrows, (const OffT*)d_row_map, d_max);
  int h_max = 0;
  KK_HIP(hipMemcpyAsync(&h_max, d_max, sizeof(int), hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  KK_HIP(hipFree(d_max));
  if (h_max > 8192)
    return fail(KKAMD_ERR_UNSUPPORTED, "kkamd_sort_crs: a row has %d entries; rows longer than 8192 are not supported yet", h_max);
  const unsigned grid = (unsigned)(nrows < 65536 ? nrows : 65536);
  if (!d_values) {
    KK_LAUNCH((sort_rows_kernel<OffT, float, false>), grid, kBlock, 0, st, nrows, (const OffT*)d_row_map, d_entries, (float*)nullptr, 8192);
  } else if (value_type == KKAMD_F64) {
    KK_LAUNCH((sort_rows_kernel<OffT, double, true>), grid, kBlock, 0, st, nrows, (const OffT*)d_row_map, d_entries, (double*)d_values, 8192);
  } else {
    KK_LAUNCH((sort_rows_kernel<OffT, float, true>), grid, kBlock, 0, st, nrows, (const OffT*)d_row_map, d_entries, (float*)d_values, 8192);
  }
  KK_LAUNCH_CHECK();
  return KKAMD_OK;
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
