// kk_spmv_mv.hip -- rank-2 CSR SpMV (SpMV_MV, Y := alpha*op(A)*X + beta*Y) for gfx950.
//
// Reference: sparse/impl/KokkosSparse_spmv_impl.hpp:634-1004 (SPMV_MV_LayoutLeft_Functor: a team of rows, strip-mined 8
// right-hand sides at a time on a GPU, X gathered per entry through the texture path), :547-632 (transpose, atomics).
// Here (bound: HBM; the contraction is 2 flop per 12-20 bytes moved):
//   spmv_mv3_kernel -- analysed handles: the plan cuts the matrix into row blocks whose X rows are a few contiguous runs,
//       stages those runs in LDS once per tile with coalesced 16-byte loads and walks the tile's rows out of LDS; per
//       nonzero the plan keeps a 16-bit LDS slot instead of the 32-bit column.  Tile order follows the grid strides the
//       analysis finds, so that every XCD's L2 sees each X row once.
//   spmv_mv2_kernel -- no analysis: wave-private row blocks, X rows gathered with 16-byte loads.
//   spmv_mv_kernel  -- generic fallback (any strides, odd widths); spmv_mv_transpose_kernel -- modes T/H (atomics).
#include "kk_spmv_plan.h"
#include "kk_scan.h"
#include <new>
#include <cstring>
#include <climits>
#include <vector>

namespace kk {

// ------------------------------------------------------------------------------------------------
// rank-2, no transpose.  A workgroup takes RPB = 256/SW consecutive rows; SW lanes (one per right-hand
// side of the current strip) form a row group.  The rows' nnz range is contiguous in CSR, so the
// workgroup streams it through LDS in CH-sized chunks with coalesced loads (A is read once per strip),
// and each group walks its own row's part of the chunk: LDS broadcast of (val, col), then one
// X(col, strip) access per lane -- a contiguous 8*SW bytes when X is row-major.
template <class OffT, class AT, class YT, int SW>
__global__ __launch_bounds__(kBlock) void spmv_mv_kernel(int64_t nrows, const OffT* __restrict__ row_map,
                                                         const int32_t* __restrict__ entries,
                                                         const AT* __restrict__ values, const YT* __restrict__ X,
                                                         int64_t xs0, int64_t xs1, YT* __restrict__ Y, int64_t ys0,
                                                         int64_t ys1, int64_t nvec, YT alpha, YT beta, int remap) {
  constexpr int RPB = kBlock / SW;
  constexpr int CH  = 2048;
  __shared__ AT s_val[CH];
  __shared__ int s_col[CH];
  const int t        = threadIdx.x;
  const int64_t wg   = remap ? xcd_remap(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
  const int64_t row0 = wg * RPB;
  const int64_t rowN = (row0 + RPB < nrows) ? row0 + RPB : nrows;
  const int64_t row  = row0 + t / SW;
  const int k        = t % SW;
  const int64_t lo   = (int64_t)row_map[row0];
  const int64_t hi   = (int64_t)row_map[rowN];
  int64_t rs = 0, re = 0;
  if (row < nrows) { rs = (int64_t)row_map[row]; re = (int64_t)row_map[row + 1]; }
  for (int64_t kk = 0; kk < nvec; kk += SW) {
    const bool col_ok = (kk + k) < nvec;
    YT acc            = YT(0);
    for (int64_t c = lo; c < hi; c += CH) {
      const int64_t ce = (c + CH < hi) ? c + CH : hi;
      __syncthreads();
      for (int64_t i = c + t; i < ce; i += kBlock) { s_val[i - c] = values[i]; s_col[i - c] = entries[i]; }
      __syncthreads();
      if (col_ok) {
        const int64_t a = rs > c ? rs : c, z = re < ce ? re : ce;
        const YT* xp    = X + (kk + k) * xs1;
        for (int64_t i = a; i < z; ++i) acc += (YT)s_val[i - c] * xp[(int64_t)s_col[i - c] * xs0];
      }
    }
    if (col_ok && row < nrows) {
      acc *= alpha;
      YT* yp = Y + row * ys0 + (kk + k) * ys1;
      *yp    = (beta == YT(0)) ? acc : beta * (*yp) + acc;
    }
  }
}

// rank-2, no transpose, row-major X (the fast path).  One 64-lane WAVE owns RW = 64/LPRW consecutive rows;
// LPRW lanes form a row group and each lane carries TWO right-hand sides, so one X access is a 16-byte load and a
// wave-level load instruction moves 64 x 16 B = 1 KB (the generic kernel above moves 512 B per instruction with
// 4 rows in flight and is bound by the texture path at ~13 % of the HBM roofline).  The wave's contiguous CSR
// range is staged through its private LDS slice with 16-byte loads (4-aligned windows), no workgroup barrier.
typedef int kk_i32x4 __attribute__((vector_size(16)));
template <class OffT, class AT, class YT, int LPRW, int RPL, int CHW, bool NT = false>
__global__ __launch_bounds__(kBlock) void spmv_mv2_kernel(int64_t nrows, int64_t nnz, const OffT* __restrict__ row_map,
                                                          const int32_t* __restrict__ entries,
                                                          const AT* __restrict__ values, const YT* __restrict__ X,
                                                          int64_t xs0, YT* __restrict__ Y, int64_t ys0, int64_t ys1,
                                                          int64_t nvec, YT alpha, YT beta, int y_vec_ok, int remap,
                                                          const int32_t* __restrict__ order, int64_t long_T) {
  // XCD-contiguous workgroup order (remap): the 128-byte X rows a row block touches are shared with the blocks
  // that handle rows i+-1, j+-1 (and k+-1); keeping neighbouring blocks on ONE XCD keeps those X rows in its
  // 4 MiB L2.  With the dispatcher's round-robin order every XCD fetched every X row: rocprof showed 10.7
  // memory fetches per X row (45 GB per launch on the 300^3 x 16 case against 12 GB of compulsory reads).
  // LPRW lanes per row, RPL (2 or 4) right-hand sides per lane: strip width SW = LPRW*RPL, RW rows per wave.
  // RPL = 4 halves the per-nnz LDS-read / address arithmetic per FMA; a quad of lanes then covers one 128 B X row.
  constexpr int RW  = kWave / LPRW;
  constexpr int SW  = RPL * LPRW;
  constexpr int NV2 = RPL / 2;           // 16-byte pieces per lane
  // CHW = nnz staged per wave per pass (256 / 512 / 1024, picked from the average row length): a window smaller than the
  // RW rows of the wave makes every wave run several passes with part of its lanes idle -- with 256 on the 27-pt matrix
  // (16 rows x 27 = 432 nnz) the kernel issued twice the X load instructions it needed and the texture addresser was busy
  // 95 % of the time (rocprof TA_BUSY).
  using AV = typename vec2<AT>::type;
  using XV = typename vec2<YT>::type;
  __shared__ AT s_val_all[kBlock / kWave][CHW];
  __shared__ int s_col_all[kBlock / kWave][CHW];
  const int lane64 = threadIdx.x & 63, w = threadIdx.x >> 6;
  AT* s_val  = s_val_all[w];
  int* s_col = s_col_all[w];
  // row block of this workgroup: the plan's strip order (see mv_build_strip_order) when it has one, else 0 dispatch order,
  // 1 XCD-contiguous, 4 / 8 / 16 grouped
  const int64_t wg   = order ? (int64_t)order[blockIdx.x] : xcd_order(blockIdx.x, gridDim.x, remap);
  const int64_t row0 = (wg * (kBlock / kWave) + w) * RW;
  if (row0 >= nrows) return;                                   // whole wave leaves together
  const int64_t rowN = (row0 + RW < nrows) ? row0 + RW : nrows;
  const int grp = lane64 / LPRW, l = lane64 % LPRW;
  const int64_t row = row0 + grp;
  const int64_t lo  = (int64_t)row_map[row0] & ~(int64_t)3;    // 4-aligned staging windows
  const int64_t hi  = (int64_t)row_map[rowN];
  int64_t rs = 0, re = 0;
  if (row < rowN) { rs = (int64_t)row_map[row]; re = (int64_t)row_map[row + 1]; }
  // a row above long_T entries is not walked here (its row group would work through it alone, the other 15 rows' lanes idle,
  // the whole wave staging chunk after chunk for it: R-MAT scale 22 x 16 took 37.7 ms): it gets 0 + beta y here and
  // spmv_mv_long_kernel adds its products afterwards; chunks that hold nothing but such a row are not even staged
  if (long_T > 0 && re - rs > long_T) re = rs;
  for (int64_t kk = 0; kk < nvec; kk += SW) {
    // piece q of lane l covers right-hand sides kk + q*2*LPRW + 2l and +1: the LPRW lanes of a row then read ONE contiguous
    // 16*LPRW-byte run per load instruction (one 64 B sector of the X row for LPRW = 4) instead of 16 B out of every
    // 32 B, which made each of the two instructions pull both sectors of the 128 B line through the L1
    const int64_t cA = kk + 2 * l;
    constexpr int64_t PQ = 2 * LPRW;                 // column distance between a lane's pieces
    const bool all_ok = kk + SW <= nvec;
    YT acc[RPL];
    KK_UNROLL
    for (int q = 0; q < RPL; ++q) acc[q] = YT(0);
    for (int64_t c = lo; c < hi; c += CHW) {
      if (long_T > 0) {
        const int64_t a_ = rs > c ? rs : c, z_ = re < c + CHW ? re : c + CHW;
        if (__ballot(row < rowN && a_ < z_) == 0ull) continue;           // nothing of a walked row in this chunk (wave-uniform)
      }
      KK_WAVE_SYNC();
      if (c + CHW <= nnz) {   // whole window inside the arrays (wave-uniform): three unguarded 16-byte loads per lane and 256 nnz
        KK_UNROLL
        for (int sub = 0; sub < CHW; sub += 256) {
          // NT: the matrix streams are read once -- nontemporal loads keep them from pushing X rows out of the XCD's L2
          const kk_i32x4 cc = NT ? KK_NT_LOAD(reinterpret_cast<const kk_i32x4*>(entries + c + sub + lane64 * 4)) : *reinterpret_cast<const kk_i32x4*>(entries + c + sub + lane64 * 4);
          const AV va = NT ? KK_NT_LOAD(reinterpret_cast<const AV*>(values + c + sub + lane64 * 2)) : *reinterpret_cast<const AV*>(values + c + sub + lane64 * 2);
          const AV vb = NT ? KK_NT_LOAD(reinterpret_cast<const AV*>(values + c + sub + 128 + lane64 * 2)) : *reinterpret_cast<const AV*>(values + c + sub + 128 + lane64 * 2);
          s_col[sub + lane64 * 4] = cc[0]; s_col[sub + lane64 * 4 + 1] = cc[1]; s_col[sub + lane64 * 4 + 2] = cc[2]; s_col[sub + lane64 * 4 + 3] = cc[3];
          s_val[sub + lane64 * 2] = va[0]; s_val[sub + lane64 * 2 + 1] = va[1];
          s_val[sub + 128 + lane64 * 2] = vb[0]; s_val[sub + 128 + lane64 * 2 + 1] = vb[1];
        }
      } else {
        for (int q = 0; q < CHW / 64; ++q) {
          const int64_t i = c + lane64 * (CHW / 64) + q;
          s_col[lane64 * (CHW / 64) + q] = (i < nnz) ? entries[i] : 0;
          s_val[lane64 * (CHW / 64) + q] = (i < nnz) ? values[i] : AT(0);
        }
      }
      KK_WAVE_SYNC();
      const int64_t ce = c + CHW;
      const int a = (int)((rs > c ? rs : c) - c), z = (int)((re < ce ? re : ce) - c);
      if (all_ok) {
        // batches of U entries: all U*NV2 16-byte X loads are issued before the first FMA consumes one
        // (left to itself hipcc emitted load -> s_waitcnt vmcnt(0) -> fma per entry: one load in flight per wave)
        // batches of 8, then 4, 2, 1 entries: inside a batch all X loads are issued before the first FMA consumes one
        // (left to itself hipcc emitted load -> s_waitcnt vmcnt(0) -> fma per entry: one load in flight per wave)
#define KK_MV_BATCH(UU)                                                                                                  \
        {                                                                                                                \
          YT v[UU]; XV xv[UU][NV2];                                                                                      \
          KK_UNROLL                                                                                                      \
          for (int u = 0; u < UU; ++u) {                                                                                 \
            v[u] = (YT)s_val[i + u];                                                                                     \
            const YT* xp = X + (int64_t)s_col[i + u] * xs0 + cA;                                                         \
            KK_UNROLL                                                                                                    \
            for (int q = 0; q < NV2; ++q) xv[u][q] = *reinterpret_cast<const XV*>(xp + q * PQ);                          \
          }                                                                                                              \
          KK_UNROLL                                                                                                      \
          for (int u = 0; u < UU; ++u) {                                                                                 \
            KK_UNROLL                                                                                                    \
            for (int q = 0; q < NV2; ++q) { acc[2 * q] += v[u] * xv[u][q][0]; acc[2 * q + 1] += v[u] * xv[u][q][1]; }    \
          }                                                                                                              \
          i += UU;                                                                                                       \
        }
        int i = a;
        while (i + 8 <= z) KK_MV_BATCH(8)
        if (i + 4 <= z) KK_MV_BATCH(4)
        if (i + 2 <= z) KK_MV_BATCH(2)
        if (i < z) KK_MV_BATCH(1)
#undef KK_MV_BATCH
      } else {
        for (int i = a; i < z; ++i) {
          const YT v = (YT)s_val[i];
          const YT* xp = X + (int64_t)s_col[i] * xs0 + cA;
          for (int q = 0; q < RPL; ++q) { const int64_t cq = (q >> 1) * PQ + (q & 1); if (cA + cq < nvec) acc[q] += v * xp[cq]; }
        }
      }
    }
    if (row < rowN) {
      YT* yp = Y + row * ys0 + cA * ys1;
      if (all_ok && y_vec_ok) {
        KK_UNROLL
        for (int q = 0; q < NV2; ++q) {
          XV out;
          XV* yq = reinterpret_cast<XV*>(yp + q * PQ);
          if (beta == YT(0)) { out[0] = alpha * acc[2 * q]; out[1] = alpha * acc[2 * q + 1]; }
          else { const XV old = *yq; out[0] = beta * old[0] + alpha * acc[2 * q]; out[1] = beta * old[1] + alpha * acc[2 * q + 1]; }
          *yq = out;
        }
      } else {
        for (int q = 0; q < RPL; ++q) {
          const int64_t cq = (q >> 1) * PQ + (q & 1);
          if (cA + cq < nvec) { const YT r = alpha * acc[q]; yp[cq * ys1] = (beta == YT(0)) ? r : beta * yp[cq * ys1] + r; }
        }
      }
    }
  }
}

// rows above T entries: count, then list (any order)
template <class OffT>
__global__ __launch_bounds__(kBlock) void mv_long_list_kernel(int64_t nrows, const OffT* __restrict__ row_map, int64_t T, int32_t* __restrict__ list,
                                                              unsigned long long* __restrict__ count) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r < nrows && (int64_t)row_map[r + 1] - (int64_t)row_map[r] > T) {
    const unsigned long long at = atomicAdd(count, 1ull);
    if (list) list[at] = (int32_t)r;
    else atomicAdd(count + 1, (unsigned long long)((int64_t)row_map[r + 1] - (int64_t)row_map[r]));   // counting pass: their entries too
  }
}
// Y(row, :) += alpha * A(row, :) X for the listed rows (the gather kernel left beta * Y there): one workgroup per row and strip of
// 16 right-hand sides, 16 lanes per entry (one right-hand side each: a 128-byte X row per load when X is row-major), 16 entries in
// flight per step, the 16 partial sums of a column meet in LDS
template <class OffT, class AT, class YT>
__global__ __launch_bounds__(kBlock) void spmv_mv_long_kernel(const int32_t* __restrict__ list, const OffT* __restrict__ row_map,
                                                              const int32_t* __restrict__ entries, const AT* __restrict__ values,
                                                              const YT* __restrict__ X, int64_t xs0, int64_t xs1, YT* __restrict__ Y, int64_t ys0,
                                                              int64_t ys1, int64_t nvec, YT alpha) {
  __shared__ YT part[kBlock];
  const int64_t row = list[blockIdx.x];
  const int j = threadIdx.x & 15, e = threadIdx.x >> 4;
  const int64_t col0 = (int64_t)blockIdx.y * 16;
  const int64_t jc = col0 + j < nvec ? col0 + j : nvec - 1;        // the spare lanes of the last strip read a valid column and store nothing
  const int64_t b = (int64_t)row_map[row], end = (int64_t)row_map[row + 1];
  YT acc = YT(0);
  for (int64_t a = b + e; a < end; a += 4 * (kBlock / 16)) {          // four independent entries per lane and step
    YT v[4]; int32_t c[4];
    KK_UNROLL
    for (int u = 0; u < 4; ++u) { const int64_t i = a + u * (kBlock / 16); const bool ok = i < end; c[u] = ok ? entries[i] : entries[b]; v[u] = ok ? (YT)values[i] : YT(0); }
    KK_UNROLL
    for (int u = 0; u < 4; ++u) acc += v[u] * X[(int64_t)c[u] * xs0 + jc * xs1];
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 16) {
    YT sum = YT(0);
    for (int q = 0; q < kBlock / 16; ++q) sum += part[q * 16 + j];
    if (col0 + j < nvec) Y[row * ys0 + (col0 + j) * ys1] += alpha * sum;
  }
}

// X(ncols x nvec, column-major or any strides) -> row-major, leading dimension ldp (even): the packing step that
// lets a LayoutLeft multivector use the 16-byte-per-lane row-major kernel.  32x32 LDS tile transpose.
template <class YT>
__global__ __launch_bounds__(kBlock) void pack_rows_kernel(int64_t n, int64_t nvec, const YT* __restrict__ X, int64_t xs0,
                                                           int64_t xs1, YT* __restrict__ Xp, int64_t ldp) {
  __shared__ YT tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  const int64_t i0 = (int64_t)blockIdx.x * 32;
  for (int64_t j0 = 0; j0 < nvec; j0 += 32) {
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {                             // read: consecutive lanes walk i (stride xs0)
      const int64_t i = i0 + tx, j = j0 + q;
      tile[q][tx] = (i < n && j < nvec) ? X[i * xs0 + j * xs1] : YT(0);
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {                             // write: consecutive lanes walk j (contiguous)
      const int64_t i = i0 + q, j = j0 + tx;
      if (i < n && j < ldp) Xp[i * ldp + j] = tile[tx][q];
    }
  }
}

// rank-2 transpose: Y(col, k) += alpha * val * X(row, k) after Y := beta*Y (K6 analogue).
template <class OffT, class AT, class YT>
__global__ __launch_bounds__(kBlock) void spmv_mv_transpose_kernel(int64_t nrows, const OffT* __restrict__ row_map,
                                                                   const int32_t* __restrict__ entries,
                                                                   const AT* __restrict__ values,
                                                                   const YT* __restrict__ X, int64_t xs0, int64_t xs1,
                                                                   YT* __restrict__ Y, int64_t ys0, int64_t ys1,
                                                                   int64_t nvec, YT alpha) {
  constexpr int SW  = 16;
  constexpr int RPB = kBlock / SW;
  const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / SW;
  const int k0      = threadIdx.x % SW;
  if (row >= nrows) return;
  const OffT s = row_map[row], e = row_map[row + 1];
  for (int64_t k = k0; k < nvec; k += SW) {
    const YT xv = alpha * X[row * xs0 + k * xs1];
    for (OffT j = s; j < e; ++j) atomicAdd(&Y[(int64_t)entries[j] * ys0 + k * ys1], (YT)values[j] * xv);
  }
}


// ================================================================================================================
// LDS-staged rank-2 kernel (analysed handles).
//
// A tile is RB consecutive rows (RB = 32 for 16 right-hand sides, 64 for 8).  On matrices whose rows touch a few contiguous
// column runs (stencils, banded and block-structured matrices) the X rows a tile needs are a few contiguous RUNS of X
// rows; the analysis (mv_build_kernel, once per handle) finds them -- greedily, left to right, a run ends at the first four
// consecutive unused columns, at most 16 runs -- and keeps, per tile, the first column of every 4-row CHUNK of the staged
// window and, per nonzero, a 16-bit SLOT (the X row's position in that window) instead of the 32-bit column.  The kernel
// then fetches the window with coalesced 16-byte loads -- every X row crosses L2 -> LDS once per tile instead of once per
// nonzero through the texture path (27-pt: 34 x 9 rows instead of 32 x 27 gathers) -- and walks the rows out of LDS.
// Tiles that do not stage (too many runs, window or row block too large) are handled by the same kernel through `entries`
// and gathers from X, tile by tile.
constexpr int kMvRuns     = 16;                    // contiguous X-row runs per tile
constexpr int kMvChunk    = 4;                     // X rows per chunk of the staged window
constexpr int kMvXBytes   = 44 * 1024;             // LDS a tile's X window may take
constexpr int kMvNnzCap   = 2048;                  // nonzeros of a staged tile (A sits in LDS as 12-byte records)
constexpr int kMvHdr      = 40;                    // tile meta (ints): [0] mode, [1] chunks, [2] runs, [3] nnz, [4..20) first column of every run,
                                                   // [20..36) rows of every run (multiple of kMvChunk), [40..) first column of every chunk
constexpr int kMvGather = 0, kMvStaged = 1;

}  // namespace kk

struct kkamd_mv_plan {
  int rb = 0, nv = 0;                // rows per tile, right-hand sides per strip the tiling was made for
  int meta_stride = 0;               // ints per tile in d_meta
  int64_t ntiles = 0, staged_tiles = 0;
  int32_t* d_meta = nullptr;         // [ntiles * meta_stride]
  uint16_t* d_slot = nullptr;        // [nnz] LDS slot of every nonzero's X row (staged tiles)
  int32_t* d_order = nullptr;        // [ntiles] workgroup -> tile (strip order), or null
  int order_used = 0;                // 0 dispatch, 1 XCD-contiguous, 2 strips
  int64_t period = 0;                // rows between tiles that share X rows across the far stride (0 = none found)
  int max_slots = 0, max_nnz = 0;    // over the staged tiles: sizes the dynamic LDS
  size_t bytes = 0;
};

namespace kk {

void mv_plan_destroy(kkamd_mv_plan* mv) {
  if (!mv) return;
  if (mv->d_meta) (void)hipFree(mv->d_meta);
  if (mv->d_slot) (void)hipFree(mv->d_slot);
  if (mv->d_order) (void)hipFree(mv->d_order);
  delete mv;
}
int64_t mv_plan_query(const kkamd_mv_plan* mv, int what) {
  if (!mv) return 0;
  switch (what) {
    case 0: return mv->ntiles;
    case 1: return mv->staged_tiles;
    case 2: return mv->order_used;
    case 3: return (int64_t)mv->bytes;
    case 4: return mv->period;
    default: return 0;
  }
}

template <class OffT>
__global__ __launch_bounds__(kBlock) void mv_build_kernel(int64_t nrows, int64_t ncols, const OffT* __restrict__ row_map,
                                                          const int32_t* __restrict__ entries, int rb, int max_slots, int meta_stride,
                                                          int rowbytes, int lds_budget, int32_t* __restrict__ meta, uint16_t* __restrict__ slot,
                                                          int* __restrict__ stats) {
  // stats: [0] staged tiles, [1] max slots, [2] max nnz over the staged tiles
  constexpr int PER = kMvNnzCap / kBlock;
  __shared__ int s_base[kMvRuns], s_len[kMvRuns], s_off[kMvRuns + 1];
  __shared__ unsigned s_bits[128];                             // used columns among the 4096 after the run's base
  __shared__ int s_min, s_zero, s_flag;
  const int t = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int64_t row0 = b * rb, rowN = (row0 + rb < nrows) ? row0 + rb : nrows;
  const int64_t a0 = (int64_t)row_map[row0], a1 = (int64_t)row_map[rowN];
  const int64_t n = a1 - a0;
  int32_t* m = meta + b * meta_stride;
  if (n > kMvNnzCap || n == 0) {                               // workgroup-uniform
    if (t == 0) { m[0] = kMvGather; m[1] = 0; m[2] = 0; m[3] = (int32_t)(n > INT_MAX ? INT_MAX : n); }
    return;
  }
  int c[PER];
  KK_UNROLL
  for (int k = 0; k < PER; ++k) { const int64_t i = (int64_t)k * kBlock + t; c[k] = i < n ? entries[a0 + i] : -1; }
  long long bound = 0;                                         // columns < bound are covered
  int nruns = 0;
  for (int w = 0; w < kMvRuns; ++w) {                          // workgroup-uniform control flow throughout
    if (t == 0) { s_min = INT_MAX; s_zero = 1024; }
    if (t < 128) s_bits[t] = 0u;
    __syncthreads();
    int mn = INT_MAX;
    KK_UNROLL
    for (int k = 0; k < PER; ++k) if (c[k] >= 0 && (long long)c[k] >= bound && c[k] < mn) mn = c[k];
    if (mn != INT_MAX) atomicMin(&s_min, mn);
    __syncthreads();
    const int base = s_min;
    if (base == INT_MAX) break;
    KK_UNROLL
    for (int k = 0; k < PER; ++k) {
      const long long d = (long long)c[k] - base;
      if (c[k] >= 0 && d >= 0 && d < 4096) atomicOr(&s_bits[d >> 5], 1u << (d & 31));
    }
    __syncthreads();
    // the run ends at the first kMvChunk-aligned group of kMvChunk unused columns (nibble j = columns [4j, 4j + 4))
    for (int j = t; j < 1024; j += kBlock) if (((s_bits[j >> 3] >> ((j & 7) * 4)) & 0xfu) == 0u) atomicMin(&s_zero, j);
    __syncthreads();
    const int len = kMvChunk * s_zero;
    if (t == 0) { s_base[w] = base; s_len[w] = len; }
    bound = (long long)base + len;
    nruns = w + 1;
    __syncthreads();
  }
  bool uncovered = false;
  KK_UNROLL
  for (int k = 0; k < PER; ++k) uncovered |= (c[k] >= 0 && (long long)c[k] >= bound);
  if (t == 0) {
    s_flag = 0;
    int off = 0;
    for (int w = 0; w < nruns; ++w) { s_off[w] = off; off += s_len[w]; }
    s_off[nruns] = off;
  }
  __syncthreads();
  if (uncovered) atomicOr(&s_flag, 1);
  __syncthreads();
  const int total = s_off[nruns];
  if (s_flag || total > max_slots || total * rowbytes + 12 * (int)n > lds_budget) {      // workgroup-uniform
    if (t == 0) { m[0] = kMvGather; m[1] = 0; m[2] = 0; m[3] = (int32_t)n; }
    return;
  }
  KK_UNROLL
  for (int k = 0; k < PER; ++k) {
    if (c[k] < 0) continue;
    int w = 0;
    for (int q = 1; q < nruns; ++q) if (s_base[q] <= c[k]) w = q;       // run bases ascend
    slot[a0 + (int64_t)k * kBlock + t] = (uint16_t)(s_off[w] + (c[k] - s_base[w]));
  }
  const int nchunks = total / kMvChunk;
  if (t == 0) {
    m[0] = kMvStaged; m[1] = nchunks; m[2] = nruns; m[3] = (int32_t)n;
    atomicAdd(stats, 1); atomicMax(stats + 1, total); atomicMax(stats + 2, (int)n);
  }
  if (t < kMvRuns) { m[4 + t] = t < nruns ? s_base[t] : 0; m[20 + t] = t < nruns ? s_len[t] : 0; }
  for (int ch = t; ch < nchunks; ch += kBlock) {
    const int s0 = ch * kMvChunk;
    int w = 0;
    for (int q = 1; q < nruns; ++q) if (s_off[q] <= s0) w = q;
    m[kMvHdr + ch] = s_base[w] + (s0 - s_off[w]);
  }
}

// Rank-2 kernel over the plan's row-block tiles.  256 work-items = RB rows x 2 halves of every row's entries (even / odd
// positions) x NV/4 lanes of four right-hand sides each (two 16-byte pieces of the X row); the halves are summed with one
// lane exchange at the end.
//   1. everything the tile needs is requested up front: its A entries (values + slots, coalesced), the row bounds, and the
//      X window in 16-byte pieces (piece g -> chunk g / PPC, whose first column comes from the tile's chunk table);
//   2. A goes to LDS as (value, byte offset of the X row in the window), the window as it is; one barrier;
//   3. every lane walks its half row in batches of four entries: four LDS reads of A (broadcast within the row's lanes),
//      eight 16-byte LDS reads of X, sixteen FMAs;
//   4. Y leaves as 16 bytes per lane, 128 contiguous bytes per row.
// Tiles in gather mode (kMvGather) take the same route with the column in place of the offset and X read from memory.
template <class OffT, class AT, int NV, bool GLDS>
__global__ __launch_bounds__(kBlock, 3) void spmv_mv3_kernel(int64_t nrows, int64_t ncols, const OffT* __restrict__ row_map,
                                                          const int32_t* __restrict__ entries, const AT* __restrict__ values,
                                                          const uint16_t* __restrict__ slot, const int32_t* __restrict__ meta,
                                                          int meta_stride, const int32_t* __restrict__ order, int order_mode,
                                                          const double* __restrict__ X, int64_t xs0, double* __restrict__ Y,
                                                          int64_t ys0, int64_t ys1, int64_t nvec, double alpha, double beta,
                                                          int y_vec_ok, int xbytes, int cap_rec KK_ABL_PARAM) {
  // KK_ABL bits (measurement build): 1 = no X staging, 2 = no contraction loop, 4 = no A staging, 8 = no Y store
  constexpr int ROWB = NV * 8;                  // bytes of one staged X row
  constexpr int LQ   = NV / 4;                  // lanes (of four right-hand sides) per half row
  constexpr int LPR  = 2 * LQ;                  // lanes per row
  constexpr int RB   = kBlock / LPR;            // rows per tile
  constexpr int PPR  = ROWB / 16;               // 16-byte pieces per X row
  constexpr int PPC  = kMvChunk * PPR;          // ... per chunk
  using XV = kk_f64x2;
  KK_DYN_SMEM(char, smem);                      // [X window: xbytes][A values: 8 * cap_rec][A offsets: 4 * cap_rec]
  char* xwin    = smem;
  double* a_val = reinterpret_cast<double*>(smem + xbytes);
  int* a_off    = reinterpret_cast<int*>(smem + xbytes + 8 * (size_t)cap_rec);
  const int t = threadIdx.x;
  const int64_t b = order ? (int64_t)order[blockIdx.x] : xcd_order(blockIdx.x, gridDim.x, order_mode);
  const int32_t* m = meta + b * meta_stride;
  const int mode = m[0], nchunks = m[1];        // workgroup-uniform
  const int64_t row0 = b * RB, rowN = (row0 + RB < nrows) ? row0 + RB : nrows;
  const int64_t a0 = (int64_t)row_map[row0], a1 = (int64_t)row_map[rowN];
  const int r = t / LPR, h = (t / LQ) & 1, l = t % LQ;
  const int64_t row = row0 + r;
  int64_t rs = 0, re = 0;
  if (row < rowN) { rs = (int64_t)row_map[row] - a0; re = (int64_t)row_map[row + 1] - a0; }
  const int64_t n = a1 - a0;
  for (int64_t kk = 0; kk < nvec; kk += NV) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t c = 0; c < n; c += cap_rec) {                 // one pass for staged tiles; gather tiles may be longer
      const int64_t ce = (c + cap_rec < n) ? c + cap_rec : n;
      if (c > 0 || kk > 0) __syncthreads();                    // the previous pass is done with the LDS
      // 1. requests: the X window first, 16 bytes per request, straight into LDS (global_load_lds: no register round trip;
      //    piece g lands at byte 16 g, i.e. wave-uniform base + 16 * lane) ...
      constexpr int AMAX = kMvNnzCap / kBlock, AR = 4;           // A entries per work-item, in rounds of AR
      if (mode == kMvStaged) {
        const int npieces = nchunks * PPC;
        constexpr int XMAX = (kMvXBytes / 16 + kBlock - 1) / kBlock;
        XV xv[GLDS ? 1 : XMAX];                                // GLDS = false (knob mv_glds 0): through registers, for comparison
        KK_UNROLL
        for (int k = 0; k < XMAX; ++k) {
          if (k * kBlock >= npieces) break;                    // workgroup-uniform
          const int g = k * kBlock + t;
          if (g < npieces && !KK_ABL(1)) {
            const int ch = g / PPC, p = g % PPC;
            int64_t col = (int64_t)m[kMvHdr + ch] + p / PPR;
            col = col < ncols ? col : ncols - 1;               // the padding of the last run may reach past the matrix
            if (GLDS) KK_GLDS16(X + col * xs0 + kk + (p % PPR) * 2, xwin + (size_t)(k * kBlock + (t & ~63)) * 16, t & 63);
            else xv[GLDS ? 0 : k] = *reinterpret_cast<const XV*>(X + col * xs0 + kk + (p % PPR) * 2);
          }
        }
        // ... then the tile's A entries (values + slots, coalesced), AR per work-item in flight, into LDS as (value, byte offset)
        for (int k0 = 0; k0 < AMAX; k0 += AR) {
          if (c + (int64_t)k0 * kBlock >= ce || KK_ABL(4)) break;             // workgroup-uniform
          AT av[AR]; int ao[AR];
          KK_UNROLL
          for (int k = 0; k < AR; ++k) {                       // unguarded (clamped) loads stay in flight together
            int64_t i = c + (int64_t)(k0 + k) * kBlock + t;
            i = i < ce ? i : ce - 1;
            av[k] = values[a0 + i]; ao[k] = (int)slot[a0 + i] * ROWB;
          }
          KK_UNROLL
          for (int k = 0; k < AR; ++k) { const int i = (k0 + k) * kBlock + t; if (c + i < ce) { a_val[i] = (double)av[k]; a_off[i] = ao[k]; } }
        }
        if (!GLDS) {
          KK_UNROLL
          for (int k = 0; k < XMAX; ++k) { const int g = k * kBlock + t; if (g < npieces && !KK_ABL(1)) *reinterpret_cast<XV*>(xwin + (size_t)g * 16) = xv[GLDS ? 0 : k]; }
        }
        KK_GLDS_WAIT();                                        // this wave's pieces have landed (the barrier covers the other waves')
      } else {
        for (int64_t i = c + t; i < ce; i += kBlock) { a_val[i - c] = (double)values[a0 + i]; a_off[i - c] = entries[a0 + i]; }
      }
      __syncthreads();
      // 3. my half of my row's entries inside this pass: positions rs + h, rs + h + 2, ...
      const int64_t lo = rs > c ? rs : c, hi = re < ce ? re : ce;
      int i = (int)(lo - c) + (int)((h - (lo - rs)) & 1);
      const int e = KK_ABL(2) ? i : (int)(hi - c);
      if (mode == kMvStaged) {
        // piece P of lane (l, h) sits in half h of the X row, piece Q in the other half: the 16 lanes ds_read_b128 serves per
        // cycle (four rows x four lanes, two of each half) then cover all 64 banks once
        const char* xl = xwin + l * 16 + h * (ROWB / 2);
        const char* xq = xwin + l * 16 + (1 - h) * (ROWB / 2);
        KK_NOUNROLL
        for (; i + 6 < e; i += 8) {
          const double v0 = a_val[i], v1 = a_val[i + 2], v2 = a_val[i + 4], v3 = a_val[i + 6];
          const int o0 = a_off[i], o1 = a_off[i + 2], o2 = a_off[i + 4], o3 = a_off[i + 6];
          const XV p0 = *reinterpret_cast<const XV*>(xl + o0), q0 = *reinterpret_cast<const XV*>(xq + o0);
          const XV p1 = *reinterpret_cast<const XV*>(xl + o1), q1 = *reinterpret_cast<const XV*>(xq + o1);
          const XV p2 = *reinterpret_cast<const XV*>(xl + o2), q2 = *reinterpret_cast<const XV*>(xq + o2);
          const XV p3 = *reinterpret_cast<const XV*>(xl + o3), q3 = *reinterpret_cast<const XV*>(xq + o3);
          acc[0] += v0 * p0[0]; acc[1] += v0 * p0[1]; acc[2] += v0 * q0[0]; acc[3] += v0 * q0[1];
          acc[0] += v1 * p1[0]; acc[1] += v1 * p1[1]; acc[2] += v1 * q1[0]; acc[3] += v1 * q1[1];
          acc[0] += v2 * p2[0]; acc[1] += v2 * p2[1]; acc[2] += v2 * q2[0]; acc[3] += v2 * q2[1];
          acc[0] += v3 * p3[0]; acc[1] += v3 * p3[1]; acc[2] += v3 * q3[0]; acc[3] += v3 * q3[1];
        }
        KK_NOUNROLL
        for (; i < e; i += 2) {
          const double v0 = a_val[i];
          const int o0 = a_off[i];
          const XV p0 = *reinterpret_cast<const XV*>(xl + o0), q0 = *reinterpret_cast<const XV*>(xq + o0);
          acc[0] += v0 * p0[0]; acc[1] += v0 * p0[1]; acc[2] += v0 * q0[0]; acc[3] += v0 * q0[1];
        }
      } else {
        const double* xg = X + kk + l * 2 + h * (NV / 2);
        constexpr int QD = NV / 2;
        const int qd = h ? -QD : QD;                           // the lane's other piece, in doubles
        KK_NOUNROLL
        for (; i + 2 < e; i += 4) {
          const double v0 = a_val[i], v1 = a_val[i + 2];
          const double* x0 = xg + (int64_t)a_off[i] * xs0;
          const double* x1 = xg + (int64_t)a_off[i + 2] * xs0;
          const XV p0 = *reinterpret_cast<const XV*>(x0), q0 = *reinterpret_cast<const XV*>(x0 + qd);
          const XV p1 = *reinterpret_cast<const XV*>(x1), q1 = *reinterpret_cast<const XV*>(x1 + qd);
          acc[0] += v0 * p0[0]; acc[1] += v0 * p0[1]; acc[2] += v0 * q0[0]; acc[3] += v0 * q0[1];
          acc[0] += v1 * p1[0]; acc[1] += v1 * p1[1]; acc[2] += v1 * q1[0]; acc[3] += v1 * q1[1];
        }
        for (; i < e; i += 2) {
          const double v0 = a_val[i];
          const double* x0 = xg + (int64_t)a_off[i] * xs0;
          const XV p0 = *reinterpret_cast<const XV*>(x0), q0 = *reinterpret_cast<const XV*>(x0 + qd);
          acc[0] += v0 * p0[0]; acc[1] += v0 * p0[1]; acc[2] += v0 * q0[0]; acc[3] += v0 * q0[1];
        }
      }
    }
    // the two halves of a row meet: the partner (LQ lanes away) holds this lane's piece P as its piece Q
    const double f0 = acc[0] + __shfl_xor(acc[2], LQ, 64), f1 = acc[1] + __shfl_xor(acc[3], LQ, 64);
    // 4. lane (l, h) writes right-hand sides kk + 2l + (NV/2) h, + 1: 16 bytes per lane, the row's 8 NV bytes contiguous
    if (row < rowN && !KK_ABL(8)) {
      const int64_t cq = kk + 2 * l + (NV / 2) * h;
      const double s0 = alpha * f0, s1 = alpha * f1;
      double* yp = Y + row * ys0 + cq * ys1;
      if (y_vec_ok) {
        XV out;
        if (beta == 0.0) { out[0] = s0; out[1] = s1; }
        else { const XV old = *reinterpret_cast<const XV*>(yp); out[0] = beta * old[0] + s0; out[1] = beta * old[1] + s1; }
        *reinterpret_cast<XV*>(yp) = out;
      } else {
        yp[0]   = (beta == 0.0) ? s0 : beta * yp[0] + s0;
        yp[ys1] = (beta == 0.0) ? s1 : beta * yp[ys1] + s1;
      }
    }
  }
}

// ---- host side of the LDS-staged kernel ------------------------------------------------------------------------
constexpr int kMvLdsBytes = 60 * 1024;     // X window + A records of one tile (the analysis accepts a tile only if it fits)

// Far stride of the matrix, from the run tables of a few tiles: a run's centre minus the tile's centre is the offset d of
// a "diagonal band" (27-pt: 0, +-S1, +-S2, +-S2 +-S1); the far cluster (offsets above half the largest) has S2 as its median.
static int64_t mv_detect_period(const kkamd_mv_plan* mv, int64_t nrows, hipStream_t st) {
  if (mv->ntiles < 64) return 0;
  int64_t votes[16]; int nv = 0;
  for (int s = 1; s <= 15; ++s) {
    const int64_t tile = mv->ntiles * s / 16;
    int32_t hdr[kMvHdr];
    if (hipMemcpyAsync(hdr, mv->d_meta + tile * mv->meta_stride, sizeof hdr, hipMemcpyDeviceToHost, st) != hipSuccess) return 0;
    if (hipStreamSynchronize(st) != hipSuccess) return 0;
    if (hdr[0] != kMvStaged) continue;
    const double centre = (double)tile * mv->rb + 0.5 * mv->rb;
    double off[kMvRuns]; int no = 0; double mx = 0;
    for (int w = 0; w < hdr[2] && w < kMvRuns; ++w) {
      const double d = (double)hdr[4 + w] + 0.5 * (double)hdr[20 + w] - centre;
      if (d > 0) { off[no++] = d; if (d > mx) mx = d; }
    }
    double far[kMvRuns]; int nf = 0;
    for (int i = 0; i < no; ++i) if (off[i] > 0.5 * mx) far[nf++] = off[i];
    if (!nf) continue;
    for (int i = 1; i < nf; ++i) { const double v = far[i]; int j = i - 1; while (j >= 0 && far[j] > v) { far[j + 1] = far[j]; --j; } far[j + 1] = v; }
    votes[nv++] = (int64_t)(far[nf / 2] + 0.5);
  }
  if (nv < 3) return 0;
  // the most frequent vote (run lengths are padded, so centres wobble by a row or two: votes within 2 rows agree)
  int64_t best = 0; int best_n = 0;
  for (int i = 0; i < nv; ++i) { int c = 0; for (int j = 0; j < nv; ++j) if (votes[j] >= votes[i] - 2 && votes[j] <= votes[i] + 2) ++c; if (c > best_n) { best_n = c; best = votes[i]; } }
  return (best_n * 2 > nv && best > 0 && best < nrows) ? best : 0;
}

// Strip order.  Tiles P rows apart share X rows (the far stride); a sweep in row order brings them P rows * row bytes apart
// in time -- 11.5 MB of X on the 27-pt 300^3 matrix, against a 4 MB L2 -- so every L2 fetches every X row once per far
// neighbour and once per XCD.  Here every XCD instead owns STRIPS of W consecutive rows of every period and walks a strip
// period after period: the three periods' worth of X rows a strip needs (3 W rows) stay in its L2, and an X row crosses the
// fabric about once.  order[8 i + x] = i-th tile of XCD x (workgroup b runs on XCD b % 8).
static int build_strip_order(int64_t nt, int64_t rb, int64_t period, int rowbytes, int l2_kb, int32_t** d_order, hipStream_t st) {
  const double l2_rows = 1e3 * (double)l2_kb / (3.0 * (double)rowbytes);   // rows of X per period a strip may keep in a 4 MB L2
  int64_t nstrips = (int64_t)((double)period / l2_rows) + 1;
  nstrips = (nstrips + kNumXcd - 1) / kNumXcd * kNumXcd;
  const int64_t width = (period + nstrips - 1) / nstrips;                 // rows per strip
  if (width < 4 * rb) return KKAMD_ERR_UNSUPPORTED;                       // strips of a few tiles: nothing to gain
  std::vector<std::vector<int32_t>> lists((size_t)nstrips);
  for (int64_t t = 0; t < nt; ++t) lists[(size_t)(((t * rb) % period) / width)].push_back((int32_t)t);   // natural order = (period, position) ascending
  std::vector<std::vector<int32_t>> xcd(kNumXcd);
  for (int64_t s = 0; s < nstrips; ++s) { auto& dst = xcd[(size_t)(s % kNumXcd)]; dst.insert(dst.end(), lists[(size_t)s].begin(), lists[(size_t)s].end()); }
  std::vector<int32_t> order; order.reserve((size_t)nt);
  size_t longest = 0;
  for (auto& v : xcd) if (v.size() > longest) longest = v.size();
  for (size_t i = 0; i < longest; ++i) for (int x = 0; x < kNumXcd; ++x) if (i < xcd[(size_t)x].size()) order.push_back(xcd[(size_t)x][i]);
  KK_HIP(hipMalloc((void**)d_order, sizeof(int32_t) * (size_t)nt));
  KK_HIP(hipMemcpyAsync(*d_order, order.data(), sizeof(int32_t) * (size_t)nt, hipMemcpyHostToDevice, st));
  KK_HIP(hipStreamSynchronize(st));
  return KKAMD_OK;
}
static int mv_build_strip_order(kkamd_mv_plan* mv, int64_t period, int rowbytes, int l2_kb, hipStream_t st) {
  int rc = build_strip_order(mv->ntiles, mv->rb, period, rowbytes, l2_kb, &mv->d_order, st);
  if (rc == KKAMD_OK) mv->bytes += sizeof(int32_t) * (size_t)mv->ntiles;
  return rc;
}

// Far stride of the matrix from the columns of a few rows (no tile analysis needed): the offsets col - row of a row cluster
// around 0, +-S1, +-S2, +-S2 +-S1; the far cluster (above half the largest) has S2 as its median.  0 = none found.
template <class OffT>
static int64_t detect_period_rows(const kkamd_crs_t* A, hipStream_t st) {
  if (A->num_rows < 4096 || A->num_rows != A->num_cols) return 0;
  // rows at scattered positions (a regular sample lands on one grid face: 27e6 * s / 16 is a multiple of 300), majority vote
  int64_t votes[32]; int nv = 0;
  for (int s = 1; s <= 32; ++s) {
    const int64_t r = (int64_t)((((unsigned long long)s * 0x9E3779B97F4A7C15ull) >> 11) % (unsigned long long)A->num_rows);
    OffT rm[2];
    if (hipMemcpyAsync(rm, (const OffT*)A->d_row_map + r, sizeof rm, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 0;
    const int64_t len = (int64_t)(rm[1] - rm[0]);
    if (len < 2 || len > 256) continue;
    int32_t cols[256];
    if (hipMemcpyAsync(cols, (const int32_t*)A->d_entries + (int64_t)rm[0], sizeof(int32_t) * (size_t)len, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) return 0;
    int64_t mx = 0;
    for (int64_t i = 0; i < len; ++i) if (cols[i] - r > mx) mx = cols[i] - r;
    int64_t far[256]; int nf = 0;
    for (int64_t i = 0; i < len; ++i) if (2 * (cols[i] - r) > mx) far[nf++] = cols[i] - r;
    if (!nf) continue;
    for (int i = 1; i < nf; ++i) { const int64_t v = far[i]; int j = i - 1; while (j >= 0 && far[j] > v) { far[j + 1] = far[j]; --j; } far[j + 1] = v; }
    votes[nv++] = far[nf / 2];
  }
  if (nv < 3) return 0;
  int64_t best = 0; int best_n = 0;
  for (int i = 0; i < nv; ++i) { int c = 0; for (int j = 0; j < nv; ++j) if (votes[j] == votes[i]) ++c; if (c > best_n) { best_n = c; best = votes[i]; } }
  return (best_n * 2 > nv && best > 0 && best < A->num_rows) ? best : 0;
}

template <class OffT>
static int mv_plan_build(kkamd_spmv_plan* plan, const kkamd_crs_t* A, int nv, hipStream_t st) {
  if (plan->mv) { mv_plan_destroy(plan->mv); plan->mv = nullptr; }
  kkamd_mv_plan* mv = new (std::nothrow) kkamd_mv_plan();
  if (!mv) return fail(KKAMD_ERR_ALLOC, "kkamd_spmv_mv: out of host memory");
  mv->nv = nv; mv->rb = 2 * kBlock / nv;                       // kBlock / (2 halves * nv / 4 lanes)
  mv->ntiles = ceil_div(A->num_rows, mv->rb);
  const int rowbytes  = nv * 8;
  const int max_slots = kMvXBytes / rowbytes;
  mv->meta_stride = (kMvHdr + max_slots / kMvChunk + 31) / 32 * 32;
  DevBuf stats;
  auto give_up = [&]() { (void)hipGetLastError(); mv_plan_destroy(mv); plan->mv_failed = true; return KKAMD_OK; };
  if (hipMalloc((void**)&mv->d_meta, sizeof(int32_t) * (size_t)mv->ntiles * (size_t)mv->meta_stride) != hipSuccess ||
      hipMalloc((void**)&mv->d_slot, sizeof(uint16_t) * (size_t)(A->nnz + 8)) != hipSuccess || stats.alloc(4 * sizeof(int)) != hipSuccess)
    return give_up();                                          // an optimisation: without the memory the wave-private kernel serves
  mv->bytes = sizeof(int32_t) * (size_t)mv->ntiles * (size_t)mv->meta_stride + sizeof(uint16_t) * (size_t)(A->nnz + 8);
  int* d_stats = stats.as<int>();
  KK_HIP(hipMemsetAsync(d_stats, 0, 4 * sizeof(int), st));
  KK_LAUNCH((mv_build_kernel<OffT>), (unsigned)mv->ntiles, kBlock, 0, st, A->num_rows, A->num_cols, (const OffT*)A->d_row_map,
            (const int32_t*)A->d_entries, mv->rb, max_slots, mv->meta_stride, rowbytes, kMvLdsBytes, mv->d_meta, mv->d_slot, d_stats);
  if (hipGetLastError() != hipSuccess) return give_up();
  int h[4] = {0, 0, 0, 0};
  KK_HIP(hipMemcpyAsync(h, stats.p, sizeof h, hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  mv->staged_tiles = h[0]; mv->max_slots = h[1]; mv->max_nnz = h[2];
  if ((double)mv->staged_tiles < 0.5 * (double)mv->ntiles) return give_up();     // mostly gathers anyway: the wave-private kernel is the better gather kernel
  mv->order_used = plan->tune.mv_order ? 1 : 0;
  if (plan->tune.mv_order == 2) {
    mv->period = mv_detect_period(mv, A->num_rows, st);
    // worth it when a period's worth of X rows overflows an L2 (4 MB per XCD)
    if (mv->period > 0 && (double)mv->period * rowbytes * 3.0 > 1e3 * (double)plan->tune.mv_strip_min_kb &&
        mv_build_strip_order(mv, mv->period, rowbytes, plan->tune.mv_strip_l2_kb, st) == KKAMD_OK) mv->order_used = 2;
    (void)hipGetLastError();
  }
  plan->mv = mv;
  return KKAMD_OK;
}

template <class OffT, class AT, int NV>
static int launch_mv3(const kkamd_spmv_plan* plan, const kkamd_crs_t* A, const double* X, int64_t ldx, double* Y, int64_t ys0,
                      int64_t ys1, int64_t nvec, double alpha, double beta, hipStream_t st) {
  const kkamd_mv_plan* mv = plan->mv;
  const int xbytes  = mv->max_slots * NV * 8;
  int cap_rec       = (mv->max_nnz + 63) / 64 * 64;
  if (cap_rec < 256) cap_rec = 256;
  const size_t lds  = (size_t)xbytes + 12 * (size_t)cap_rec;
  const int yv      = (ys1 == 1 && (ys0 % 2 == 0) && ((uintptr_t)Y % 16 == 0)) ? 1 : 0;
#define KK_MV3(G)                                                                                                            \
  KK_LAUNCH((spmv_mv3_kernel<OffT, AT, NV, G>), (unsigned)mv->ntiles, kBlock, lds, st, A->num_rows, A->num_cols, (const OffT*)A->d_row_map, \
            (const int32_t*)A->d_entries, (const AT*)A->d_values, (const uint16_t*)mv->d_slot, (const int32_t*)mv->d_meta,           \
            mv->meta_stride, (const int32_t*)(mv->order_used == 2 ? mv->d_order : nullptr), mv->order_used == 1 ? 1 : 0, X, ldx, Y, ys0, \
            ys1, nvec, alpha, beta, yv, xbytes, cap_rec KK_ABL_ARG(plan))
  if (plan->tune.mv_glds) { KK_MV3(true); } else { KK_MV3(false); }
#undef KK_MV3
  KK_LAUNCH_CHECK();
  return KKAMD_OK;
}

template <class OffT, class AT, class YT, int SW>
static int launch_mv(const kkamd_crs_t* A, const YT* X, int64_t xs0, int64_t xs1, YT* Y, int64_t ys0, int64_t ys1,
                     int64_t nvec, YT alpha, YT beta, int remap, hipStream_t st) {
  KK_LAUNCH((spmv_mv_kernel<OffT, AT, YT, SW>), (unsigned)ceil_div(A->num_rows, kBlock / SW), kBlock, 0, st,
            A->num_rows, (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, X, xs0, xs1,
            Y, ys0, ys1, nvec, alpha, beta, remap);
  KK_LAUNCH_CHECK();
  return KKAMD_OK;
}

// the wave-private kernel's row blocks in strip order (mv_order = 2, analysed handles, matrices with a far stride): built on
// the first rank-2 call for the kernel's block size and kept with the plan
template <class OffT>
static const int32_t* mv2_order_t(kkamd_spmv_plan* plan, const kkamd_crs_t* A, int rows_per_wg, int rowbytes, hipStream_t st) {
  if (!plan || plan->tile == 0 || plan->tune.mv_order != 2 || plan->entries != A->d_entries) return nullptr;
  if (plan->d_mv2_order && plan->mv2_rb == rows_per_wg) return plan->d_mv2_order;
  if (plan->mv2_tried && plan->mv2_rb == rows_per_wg) return nullptr;
  if (plan->d_mv2_order) { (void)hipStreamSynchronize(st); (void)hipFree(plan->d_mv2_order); plan->d_mv2_order = nullptr; }
  plan->mv2_tried = true; plan->mv2_rb = rows_per_wg;
  if (!plan->mv_period_known) { plan->mv_period = detect_period_rows<OffT>(A, st); plan->mv_period_known = true; }
  const int64_t period = plan->mv_period;
  if (period <= 0 || (double)period * rowbytes * 3.0 <= 1e3 * (double)plan->tune.mv_strip_min_kb) return nullptr;
  if (build_strip_order(ceil_div(A->num_rows, rows_per_wg), rows_per_wg, period, rowbytes, plan->tune.mv_strip_l2_kb, &plan->d_mv2_order, st) != KKAMD_OK) {
    (void)hipGetLastError();
    plan->d_mv2_order = nullptr;
  }
  return plan->d_mv2_order;
}
static const int32_t* mv2_order(kkamd_spmv_plan* plan, const kkamd_crs_t* A, int rows_per_wg, int rowbytes, hipStream_t st) {
  return A->offset_type == KKAMD_I64 ? mv2_order_t<int64_t>(plan, A, rows_per_wg, rowbytes, st) : mv2_order_t<int32_t>(plan, A, rows_per_wg, rowbytes, st);
}

// ================================================================================================================
// (the kernel, its gather kernel for the rows outside the stencil and launch_mv4 live in kk_spmv_mv4.h; one translation unit per type pair instantiates them)
}  // namespace kk
#include "kk_spmv_mv4.h"
namespace kk {

void mv4_plan_destroy(kkamd_mv4_plan* p) {
  if (!p) return;
  if (p->d_arow) (void)hipFree(p->d_arow);
  if (p->d_amask) (void)hipFree(p->d_amask);
  if (p->d_nc) (void)hipFree(p->d_nc);
  delete p;
}
int64_t mv4_plan_query(const kkamd_mv4_plan* p, int what) {
  if (!p) return 0;
  switch (what) {
    case 0: return p->npi * p->npj * p->nchunk;   // workgroups
    case 1: return p->n_nc;                       // rows outside the stencil
    case 2: return p->grp.n;
    case 3: return (int64_t)p->bytes;
    case 4: return p->S1;
    case 5: return p->npi * p->npj * p->nchunk1;
    default: return 0;
  }
}

// (mv4_verify_kernel and mv4_list_kernel, the analysis kernels, live in kk_spmv_mv4.h and are instantiated in kk_spmv_mv4_aux.hip: launch_mv4_verify / launch_mv4_list)
// floor((d + s / 2) / s): the lattice step an offset d makes along a stride s
static inline int64_t mv4_round_div(int64_t d, int64_t s) {
  const int64_t v = d + s / 2;
  return v >= 0 ? v / s : -((-v + s - 1) / s);
}

template <class OffT>
static int mv4_plan_build(kkamd_spmv_plan* plan, const kkamd_crs_t* A, hipStream_t st) {
  plan->mv4_tried = true;
  if (A->num_rows != A->num_cols || A->num_rows < 4096) return KKAMD_OK;
  if (!plan->mv_period_known) { plan->mv_period = detect_period_rows<OffT>(A, st); plan->mv_period_known = true; }
  int64_t S2 = plan->mv_period;
  if (S2 <= 0 || A->num_rows % S2) return KKAMD_OK;
  // the near stride and the offset list: the longest of a few scattered rows whose offsets all decompose as
  // dk S2 + dj S1 + di with |dk|, |dj|, |di| <= 1 (a shorter one is a boundary row)
  int64_t S1 = 0; Mv4Tab offs{};
  int64_t S1_fixed = 0;                                // 2-D lattices (second attempt below): the near stride is given
  for (int attempt = 0; attempt < 2 && !S1; ++attempt) {
  if (attempt == 1) {
    // No near stride below the far one: a 2-D lattice whose only stride is the line length.  Its lines are taken m at a time as
    // the "planes" the kernel marches through (S1 = line length, S2 = m lines): the first and the last line of every group then
    // miss the neighbours the lattice of m lines says they should not have -- mv4_verify_kernel sends them (2 / m of the rows)
    // to the gather rows, everything else marches.  m: a divisor of the number of lines, 32..128, a multiple of the patch's 4 lines if possible.
    if (!plan->tune.mv4_2d) break;
    const int64_t line = plan->mv_period, nlines = A->num_rows / line;
    int64_t m_best = 0;
    for (int pass = 0; pass < 2 && !m_best; ++pass)
      for (int64_t mm = 128; mm >= 32; --mm)
        if (nlines % mm == 0 && nlines / mm >= 4 && (pass == 1 || mm % kMv4RJ == 0)) { m_best = mm; break; }
    if (!m_best || line < 8) break;
    S1_fixed = line; S2 = line * m_best;
  }
  for (int s = 1; s <= 32; ++s) {
    const int64_t r = (int64_t)((((unsigned long long)s * 0x9E3779B97F4A7C15ull) >> 11) % (unsigned long long)A->num_rows);
    OffT rm[2];
    if (hipMemcpyAsync(rm, (const OffT*)A->d_row_map + r, sizeof rm, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return KKAMD_OK;
    const int len = (int)(rm[1] - rm[0]);
    if (rm[1] - rm[0] < 3 || rm[1] - rm[0] > kMv4MaxL || len <= offs.n) continue;
    int32_t cols[kMv4MaxL];
    if (hipMemcpyAsync(cols, (const int32_t*)A->d_entries + (int64_t)rm[0], sizeof(int32_t) * (size_t)len, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) return KKAMD_OK;
    int64_t near[kMv4MaxL]; int nn = 0;                // offsets between the unit neighbours and the far cluster: S1 - 1, S1, S1 + 1
    for (int q = 0; q < len; ++q) { const int64_t d = cols[q] - r; if (d > 1 && 2 * d < S2) near[nn++] = d; }
    if (!nn && !S1_fixed) continue;
    for (int a = 1; a < nn; ++a) { const int64_t v = near[a]; int e = a - 1; while (e >= 0 && near[e] > v) { near[e + 1] = near[e]; --e; } near[e + 1] = v; }
    const int64_t cand = S1_fixed ? S1_fixed : near[nn / 2];
    if (cand < 3 || S2 % cand) continue;
    bool okr = true;
    for (int q = 0; q < len && okr; ++q) {
      const int64_t d = cols[q] - r, dk = mv4_round_div(d, S2), rem = d - dk * S2, dj = mv4_round_div(rem, cand), di = rem - dj * cand;
      okr = dk >= -1 && dk <= 1 && dj >= -1 && dj <= 1 && di >= -1 && di <= 1;
    }
    // a row that stores one column twice (the reference sums duplicates) is no pattern: two of its entries would land in
    // the same slot of the value buffer.  With a duplicate-free pattern the in-order match of mv4_verify_kernel sends
    // every row that does hold a duplicate to the gather rows.
    for (int q = 0; q < len && okr; ++q)
      for (int p = 0; p < q && okr; ++p) okr = cols[p] != cols[q];
    if (!okr) continue;
    S1 = cand; offs.n = len;
    for (int q = 0; q < len; ++q) offs.e[q] = (int)(cols[q] - r);
  }
  }
  if (!S1) return KKAMD_OK;
  const int64_t nx = S1, ny = S2 / S1, nz = A->num_rows / S2;
  if (nx < 8 || ny < 3 || nz < 3 || nx > (1 << 24) || ny > (1 << 24) || nz > (1 << 24)) return KKAMD_OK;
  kkamd_mv4_plan* m = new (std::nothrow) kkamd_mv4_plan();
  if (!m) return KKAMD_OK;
  auto drop = [&]() { (void)hipGetLastError(); mv4_plan_destroy(m); return KKAMD_OK; };   // an optimisation: the gather kernel serves
  m->nx = (int)nx; m->ny = (int)ny; m->nz = (int)nz; m->S1 = S1; m->S2 = S2;
  DevBuf cnt;
  if (hipMalloc(&m->d_arow, sizeof(OffT) * (size_t)A->num_rows) != hipSuccess || hipMalloc((void**)&m->d_amask, sizeof(uint32_t) * (size_t)A->num_rows) != hipSuccess ||
      cnt.alloc(2 * sizeof(unsigned long long)) != hipSuccess) return drop();
  unsigned long long* d_cnt = cnt.as<unsigned long long>();
  OffT* d_arow = (OffT*)m->d_arow;
  if (hipMemsetAsync(d_cnt, 0, 2 * sizeof(unsigned long long), st) != hipSuccess) return drop();
  Mv4Tab steps{};                                      // the lattice step of every stencil entry, for the in/out-of-lattice tests
  steps.n = offs.n;
  for (int q = 0; q < offs.n; ++q) {
    const int64_t d = offs.e[q], dk = mv4_round_div(d, S2), rem = d - dk * S2, dj = mv4_round_div(rem, S1), di = rem - dj * S1;
    steps.e[q] = (int)((dk + 1) | ((dj + 1) << 2) | ((di + 1) << 4));
  }
  uint32_t* d_amask = m->d_amask;
  launch_mv4_verify<OffT>(A->num_rows, (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, offs, steps, m->nx, m->ny, m->nz, d_arow, d_amask, d_cnt, st);
  unsigned long long h_bad = 0;
  if (hipMemcpyAsync(&h_bad, d_cnt, sizeof h_bad, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return drop();
  m->n_nc = (int64_t)h_bad;
  if ((double)m->n_nc > 0.5 * (double)A->num_rows) return drop();               // mostly irregular: not this kernel's matrix
  if (hipMalloc((void**)&m->d_nc, sizeof(int32_t) * (size_t)(m->n_nc > 0 ? m->n_nc : 1)) != hipSuccess) return drop();
  int32_t* d_nc = m->d_nc;
  launch_mv4_list<OffT>(A->num_rows, (const OffT*)d_arow, d_nc, d_cnt + 1, st);
  if (hipStreamSynchronize(st) != hipSuccess) return drop();
  // the kernel's view of the stencil: groups (dk, di), in ascending order of that key, of up to three entries dj = -1, 0, 1
  m->grp.n = offs.n; m->grp.ng = 0;
  auto step_of = [&](int q, int64_t& dk, int64_t& dj, int64_t& di) {
    const int64_t d = offs.e[q]; dk = mv4_round_div(d, S2); const int64_t rem = d - dk * S2; dj = mv4_round_div(rem, S1); di = rem - dj * S1;
  };
  for (int kd = -1; kd <= 1; ++kd)
    for (int id = -1; id <= 1; ++id) {
      int pres = 0;
      for (int q = 0; q < offs.n; ++q) { int64_t dk, dj, di; step_of(q, dk, dj, di); if (dk == kd && di == id) pres |= 1 << (int)(dj + 1); }
      if (!pres) continue;
      const int g = m->grp.ng++;
      m->grp.e[g] = (kd + 1) | (id * 4); m->grp.pres[g] = pres;
      for (int q = 0; q < offs.n; ++q) { int64_t dk, dj, di; step_of(q, dk, dj, di); if (dk == kd && di == id) m->grp.perm[q] = 3 * g + (int)(dj + 1); }
    }
  for (int q = 0; q < offs.n; ++q) { int64_t dk, dj, di; step_of(q, dk, dj, di); m->grp.ent[q] = (int)((dk + 1) | ((dj * (kMv4RI + 2) + di) * 4)); }
  m->npi = ceil_div(nx, (int64_t)kMv4RI); m->npj = ceil_div(ny, (int64_t)kMv4RJ);
  // k-chunks: enough workgroups to fill the chip several times over (one workgroup per CU at a time), few halo planes
  int64_t nchunk = ceil_div((int64_t)plan->num_cus * plan->tune.mv4_wg_per_cu, m->npi * m->npj);
  if (nchunk > nz / 4) nchunk = nz / 4;
  if (nchunk < 1) nchunk = 1;
  m->kc = (int)ceil_div(nz, nchunk); m->nchunk = ceil_div(nz, (int64_t)m->kc);
  if (m->npi * m->npj * m->nchunk > (int64_t)INT32_MAX) return drop();          // more workgroups than a launch takes
  m->bytes = (sizeof(OffT) + sizeof(uint32_t)) * (size_t)A->num_rows + sizeof(int32_t) * (size_t)(m->n_nc > 0 ? m->n_nc : 1);
  plan->mv4 = m;
  return KKAMD_OK;
}

// ================================================================================================================
// Rank 1 on the same analysis (knob march): the patch, the marching order and the X ring of the kernel above with 8-byte X
// rows -- a plane of the patch with its halo is 1.6 KB, so four workgroups share a CU and occupancy, not a register pipeline,
// hides the latency.  8 lanes per row pair; lane c multiplies entries c, c + 8, ... of both rows (their values come straight
// from HBM into that lane: 64-byte pieces, no staging), the 8 partial sums meet in three xor-shuffles.  No column index is
// read: values 8 B per nonzero, 8 B of row words per row, x once per patch plane.
template <class OffT, class AT, bool BETA0>
__global__ __launch_bounds__(kMv4Threads) void spmv_march1_kernel(const OffT* __restrict__ arow, const uint32_t* __restrict__ amask,
                                                                  const AT* __restrict__ values, Mv4Groups G, const double* __restrict__ x,
                                                                  double* __restrict__ y, double alpha, double beta, int nx, int ny, int nz,
                                                                  int64_t S1, int64_t S2, int64_t npi, int64_t npj, int kc) {
  constexpr int RI = kMv4RI, RJ = kMv4RJ, W = RI + 2, SLABR = (RJ + 2) * W;
  constexpr int AV = (kMv4MaxL + 7) / 8;
  __shared__ double ring[4][SLABR];
  __shared__ int ent_s[kMv4MaxL];
  const int t = threadIdx.x, rs = t >> 3, line = rs / RI, ii = rs % RI, c = t & 7;
  const int64_t b = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t npatch = npi * npj;
  const int64_t ch = b / npatch, p = b % npatch;
  const int i0 = (int)(p % npi) * RI, j0 = (int)(p / npi) * RJ;
  const int kbeg = (int)ch * kc, kend = (kbeg + kc < nz) ? kbeg + kc : nz;
  const int njj = (j0 + RJ <= ny) ? RJ : ny - j0;
  KK_UNROLL
  for (int q = 0; q < kMv4MaxL; ++q) if (t == q) ent_s[q] = G.ent[q];
  bool lane_ok[2];
  int64_t r0[2];
  KK_UNROLL
  for (int u = 0; u < 2; ++u) {
    const int jj = 2 * line + u;
    lane_ok[u] = jj < njj && i0 + ii < nx;
    r0[u] = lane_ok[u] ? (int64_t)(j0 + jj) * S1 + i0 + ii : 0;
  }
  // this thread's point of a slab (threads below SLABR): clamped address, zero when it lies outside the lattice
  const int xg = t < SLABR ? t : SLABR - 1;
  const int xjr = j0 - 1 + xg / W, xir = i0 - 1 + xg % W;
  const bool x_in = xjr >= 0 && xjr < ny && xir >= 0 && xir < nx;
  const double* xpt = x + (int64_t)(xjr < 0 ? 0 : (xjr > ny - 1 ? ny - 1 : xjr)) * S1 + (xir < 0 ? 0 : (xir > nx - 1 ? nx - 1 : xir));
  auto plane_clamped = [&](int kp) -> int64_t { return kp < 0 ? 0 : (kp > nz - 1 ? nz - 1 : kp); };
  auto conforms = [&](int k, int u, OffT w) { return lane_ok[u] && k < kend && w >= 0; };
  auto load_words = [&](int kp, OffT (&w)[2], uint32_t (&mk)[2]) {
    const int64_t off = plane_clamped(kp) * S2;
    KK_UNROLL
    for (int u = 0; u < 2; ++u) { w[u] = arow[r0[u] + off]; mk[u] = amask[r0[u] + off]; }
  };
  auto put_slab = [&](int kp, double v) {               // into ring slot kp & 3; planes and points outside the lattice hold 0
    if (t < SLABR && kp <= kend) ring[kp & 3][t] = (x_in && kp >= 0 && kp < nz) ? v : 0.0;
  };
  const uint32_t full = G.n >= 32 ? 0xffffffffu : ((1u << G.n) - 1u);
  __syncthreads();
  int qoff[AV], edk[AV], erel[AV];                     // this lane's entries c + 8 m: clamped index, plane selector, position relative to the row's point
  KK_UNROLL
  for (int m = 0; m < AV; ++m) {
    qoff[m] = (c + 8 * m < G.n) ? c + 8 * m : G.n - 1;
    const int e = ent_s[qoff[m]];
    edk[m] = e & 3; erel[m] = e >> 2;
  }
  auto load_values = [&](int k, const OffT (&w)[2], const uint32_t (&mk)[2], AT (&ra)[2][AV]) {
    KK_UNROLL
    for (int u = 0; u < 2; ++u) {
      const bool cf = conforms(k, u, w[u]);
      const AT* vb = values + (cf ? (int64_t)w[u] : 0);
      const bool part = cf && mk[u] != full;
      if (!__any(part)) {
        KK_UNROLL
        for (int m = 0; m < AV; ++m) ra[u][m] = vb[qoff[m]];
      } else {
        const uint32_t mask = cf ? mk[u] : 1u;
        KK_UNROLL
        for (int m = 0; m < AV; ++m) {
          const int q = c + 8 * m;
          ra[u][m] = vb[((mask >> q) & 1u) ? __popc(mask & ((1u << q) - 1u)) : 0];
        }
      }
    }
  };
  // Three register sets in rotation (the plane loop is unrolled by three so that every index is static): at the top of plane k
  // the values of plane k + 2, the row words of plane k + 3 and the x plane k + 3 are requested; what plane k computes with was
  // requested two planes ago.  Row words are moved from their stage into the set of their plane one plane after their load.
  OffT w[3][2], w_stage[2];
  uint32_t mk[3][2], m_stage[2];
  AT ra[3][2][AV];
  double xs[3];
  load_words(kbeg, w[0], mk[0]); load_words(kbeg + 1, w[1], mk[1]); load_words(kbeg + 2, w_stage, m_stage);
  load_values(kbeg, w[0], mk[0], ra[0]);
  load_values(kbeg + 1, w[1], mk[1], ra[1]);
  for (int kp = kbeg - 1; kp <= kbeg + 1; ++kp) put_slab(kp, xpt[plane_clamped(kp) * S2]);
  xs[2] = xpt[plane_clamped(kbeg + 2) * S2];            // set 2 plays "requested in plane kbeg - 1"
  __syncthreads();
  for (int kk = kbeg; kk < kend; kk += 3) {
    KK_UNROLL
    for (int P = 0; P < 3; ++P) {
      const int k = kk + P;
      if (k >= kend) break;
      const int N2 = (P + 2) % 3;                      // the set of plane k + 2 (it held plane k - 1)
      KK_UNROLL
      for (int u = 0; u < 2; ++u) { w[N2][u] = w_stage[u]; mk[N2][u] = m_stage[u]; }
      load_words(k + 3, w_stage, m_stage);
      xs[P] = xpt[plane_clamped(k + 3) * S2];
      load_values(k + 2, w[N2], mk[N2], ra[N2]);
      double acc[2] = {0.0, 0.0};
      KK_UNROLL
      for (int u = 0; u < 2; ++u) {
        if (conforms(k, u, w[P][u])) {
          const int own = (2 * line + u + 1) * W + ii + 1;
          KK_UNROLL
          for (int m = 0; m < AV; ++m) {
            const int q = c + 8 * m;
            if (q < G.n && ((mk[P][u] >> q) & 1u)) acc[u] = __builtin_fma((double)ra[P][u][m], ring[(k + edk[m] - 1) & 3][own + erel[m]], acc[u]);
          }
        }
      }
      KK_UNROLL
      for (int u = 0; u < 2; ++u) {
        acc[u] += __shfl_xor(acc[u], 1, 64);
        acc[u] += __shfl_xor(acc[u], 2, 64);
        acc[u] += __shfl_xor(acc[u], 4, 64);
      }
      put_slab(k + 2, xs[N2]);                          // requested in plane k - 1; its slot held plane k - 2, read for the last time before the previous barrier
      // lane 0 of a pair writes its first row, lane 1 the second: the wave's 8 pairs are neighbours in i, 64 contiguous bytes each
      if (c < 2) {
        const OffT wu = c == 0 ? w[P][0] : w[P][1];
        if (conforms(k, c, wu)) {
          double* yp = y + (c == 0 ? r0[0] : r0[1]) + (int64_t)k * S2;
          const double sres = alpha * (c == 0 ? acc[0] : acc[1]);
          *yp = BETA0 ? sres : beta * (*yp) + sres;
        }
      }
      __syncthreads();
    }
  }
}

// rows outside the stencil pattern, rank 1: 16 lanes per row
template <class OffT, class AT>
__global__ __launch_bounds__(kBlock) void march1_rows_kernel(int64_t n_list, const int32_t* __restrict__ list, const OffT* __restrict__ row_map,
                                                             const int32_t* __restrict__ entries, const AT* __restrict__ values,
                                                             const double* __restrict__ x, double* __restrict__ y, double alpha, double beta) {
  int64_t idx = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / 16;
  const int j = threadIdx.x & 15;
  const bool live = idx < n_list;
  if (!live) idx = n_list - 1;
  const int64_t r = list[idx];
  double acc = 0.0;
  for (int64_t a = (int64_t)row_map[r] + j; a < (int64_t)row_map[r + 1]; a += 16) acc += (double)values[a] * x[entries[a]];
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 16);
  if (live && j == 0) y[r] = (beta == 0.0) ? alpha * acc : beta * y[r] + alpha * acc;
}

template <class OffT, class AT>
static int march1_launch(kkamd_spmv_plan* plan, const kkamd_crs_t* A, const double* x, double* y, double alpha, double beta, hipStream_t st) {
  kkamd_mv4_plan* m = plan->mv4;
  if (m->kc1 != plan->tune.march_planes || m->nchunk1 == 0) {       // the k-chunks of this kernel: about march_planes planes each
    int kc1 = plan->tune.march_planes < m->nz ? plan->tune.march_planes : m->nz;
    m->nchunk1 = ceil_div((int64_t)m->nz, (int64_t)kc1);
    m->kc1 = plan->tune.march_planes;
    m->kc1_planes = (int)ceil_div((int64_t)m->nz, m->nchunk1);
  }
  const unsigned grid = (unsigned)(m->npi * m->npj * m->nchunk1);
  if (beta == 0.0)
    KK_LAUNCH((spmv_march1_kernel<OffT, AT, true>), grid, kMv4Threads, 0, st, (const OffT*)m->d_arow, (const uint32_t*)m->d_amask, (const AT*)A->d_values,
              m->grp, x, y, alpha, beta, m->nx, m->ny, m->nz, m->S1, m->S2, m->npi, m->npj, m->kc1_planes);
  else
    KK_LAUNCH((spmv_march1_kernel<OffT, AT, false>), grid, kMv4Threads, 0, st, (const OffT*)m->d_arow, (const uint32_t*)m->d_amask, (const AT*)A->d_values,
              m->grp, x, y, alpha, beta, m->nx, m->ny, m->nz, m->S1, m->S2, m->npi, m->npj, m->kc1_planes);
  KK_LAUNCH_CHECK();
  if (m->n_nc > 0) {
    KK_LAUNCH((march1_rows_kernel<OffT, AT>), (unsigned)ceil_div(m->n_nc * 16, (int64_t)kBlock), kBlock, 0, st, m->n_nc, (const int32_t*)m->d_nc,
              (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, x, y, alpha, beta);
    KK_LAUNCH_CHECK();
  }
  return KKAMD_OK;
}

int march_spmv(kkamd_spmv_plan* plan, const kkamd_crs_t* A, const double* x, double* y, double alpha, double beta, hipStream_t st, int* ran) {
  *ran = 0;
  if (!plan->mv4 && !plan->mv4_tried) {
    const int rc = A->offset_type == KKAMD_I64 ? mv4_plan_build<int64_t>(plan, A, st) : mv4_plan_build<int32_t>(plan, A, st);
    if (rc) return rc;
  }
  if (!plan->mv4) return KKAMD_OK;
  *ran = 1;
  if (A->offset_type == KKAMD_I64)
    return A->value_type == KKAMD_F64 ? march1_launch<int64_t, double>(plan, A, x, y, alpha, beta, st) : march1_launch<int64_t, float>(plan, A, x, y, alpha, beta, st);
  return A->value_type == KKAMD_F64 ? march1_launch<int32_t, double>(plan, A, x, y, alpha, beta, st) : march1_launch<int32_t, float>(plan, A, x, y, alpha, beta, st);
}

// the rows the wave-per-16-rows gather kernel leaves to spmv_mv_long_kernel: found once per plan
template <class OffT>
static int mv_find_long_rows(kkamd_spmv_plan* plan, const kkamd_crs_t* A, hipStream_t st) {
  plan->mv_long_known = true; plan->n_mv_long = 0; plan->mv_long_T = 0; plan->mv_long_nnz = 0;
  if (A->num_rows == 0 || A->nnz == 0) return KKAMD_OK;
  int64_t T = plan->tune.mv_long_T;
  if (T <= 0) { T = 4 * (A->nnz / A->num_rows); if (T < 64) T = 64; }     // R-MAT scale 22 x 16: 37.7 ms without, 9.5 at 1024, 4.8 at 256, 3.0-3.1 at 64-128
  DevBuf cnt;
  KK_HIP(cnt.alloc(2 * sizeof(unsigned long long)));
  unsigned long long* d_cnt = cnt.as<unsigned long long>();
  KK_HIP(hipMemsetAsync(d_cnt, 0, 2 * sizeof(unsigned long long), st));
  const unsigned grid = (unsigned)ceil_div(A->num_rows, kBlock);
  KK_LAUNCH((mv_long_list_kernel<OffT>), grid, kBlock, 0, st, A->num_rows, (const OffT*)A->d_row_map, T, (int32_t*)nullptr, d_cnt);
  unsigned long long h_c[2] = {0, 0};
  KK_HIP(hipMemcpyAsync(h_c, d_cnt, sizeof h_c, hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  const unsigned long long h_n = h_c[0];
  plan->mv_long_nnz = (int64_t)h_c[1];
  if (h_n == 0) return KKAMD_OK;
  KK_HIP(hipMalloc((void**)&plan->d_mv_long, sizeof(int32_t) * (size_t)h_n));
  KK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), st));
  int32_t* d_list = plan->d_mv_long;
  KK_LAUNCH((mv_long_list_kernel<OffT>), grid, kBlock, 0, st, A->num_rows, (const OffT*)A->d_row_map, T, d_list, d_cnt);
  KK_HIP(hipStreamSynchronize(st));
  plan->n_mv_long = (int64_t)h_n; plan->mv_long_T = T;
  return KKAMD_OK;
}

template <class OffT, class AT, class YT>
static int spmv_mv_typed(kkamd_spmv_plan* plan, const kkamd_crs_t* A, bool trans, double alpha_d, const void* dX,
                         int64_t xs0, int64_t xs1, double beta_d, void* dY, int64_t ys0, int64_t ys1, int64_t nvec,
                         hipStream_t st) {
  const YT alpha = (YT)alpha_d, beta = (YT)beta_d;
  const YT* X    = (const YT*)dX;
  YT* Y          = (YT*)dY;
  const int remap = (plan ? plan->tune.xcd_remap : g_spmv_default.xcd_remap) == 1;
  if (trans) {
    // an analysed handle runs the rank-2 dispatch for mode N on its cached transpose (built on first use, values kept current under the
    // "values_tracking" policy; kk_spmv.hip): no atomics, deterministic.  Otherwise the reference's scatter (spmv_impl.hpp:547-632)
    if (plan) {
      kkamd_crs_t At{}; kkamd_spmv_plan* tplan = nullptr;
      const int rc = transpose_view(plan, A, st, &At, &tplan);
      if (rc) return rc;
      if (tplan) return spmv_mv_typed<OffT, AT, YT>(tplan, &At, false, alpha_d, dX, xs0, xs1, beta_d, dY, ys0, ys1, nvec, st);
    }
    int rc = launch_scale<YT>(Y, A->num_cols, ys0, nvec, ys1, beta, st);
    if (rc) return rc;
    KK_LAUNCH((spmv_mv_transpose_kernel<OffT, AT, YT>), (unsigned)ceil_div(A->num_rows, kBlock / 16), kBlock, 0, st,
              A->num_rows, (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, X, xs0, xs1,
              Y, ys0, ys1, nvec, alpha);
    KK_LAUNCH_CHECK();
    return KKAMD_OK;
  }
  const int mvk = plan ? plan->tune.mv_kernel : g_spmv_default.mv_kernel;      // 0 auto, 1 generic, 2 wave-private row-major, 3 LDS-staged X tiles, 4 plane marching
  const bool a_aligned = ((uintptr_t)A->d_values % 16 == 0) && ((uintptr_t)A->d_entries % 16 == 0);
  if (mvk != 1 && a_aligned) {
    // plane-marching kernel (mv_kernel 0 = auto, or 4): analysed handles, fp64 vectors, right-hand sides in blocks of 16 (a
    // remainder of fewer than 16 columns goes to the gather kernel),
    // matrices that verify as a radius-1 lattice stencil; the analysis happens on the first such call.  X and Y keep their
    // strides (row-major X is read with 16-byte loads, anything else with two 8-byte loads per piece): nothing is packed
    if constexpr (sizeof(YT) == 8) {
      if (plan && plan->tile != 0 && (mvk == 0 || mvk == 4) && nvec >= plan->tune.mv4_min_nvec && plan->entries == A->d_entries) {
        if (!plan->mv4 && !plan->mv4_tried) {
          int rc = mv4_plan_build<OffT>(plan, A, st);
          if (rc) return rc;
        }
        if (plan->mv4) {
          // the full blocks of 16 columns in ONE launch (their workgroups side by side: the values cross HBM once), at most what a grid holds
          int64_t c0 = 0;
          const int64_t wgs = plan->mv4->npi * plan->mv4->npj * plan->mv4->nchunk;
          // a width that is no multiple of 16 (17 .. 128 columns, an even remainder or an X that is not row-major): all its blocks in one
          // launch of the partial-block form -- the values once, and the lines of X the blocks share once
          {
            const int64_t nb = ceil_div(nvec, (int64_t)16), rem = nvec % 16;
            const bool xrow = xs1 == 1 && (xs0 % 2 == 0) && ((uintptr_t)X % 16 == 0);
            if (rem != 0 && nb >= 2 && nb <= 8 && (rem % 2 == 0 || !xrow) && wgs * nb <= (int64_t)INT32_MAX && plan->tune.mv4_min_nvec <= 16)
              return launch_mv4<OffT, AT>(plan, A, (const double*)X, xs0, xs1, (double*)Y, ys0, ys1, (double)alpha, (double)beta, st, (int)rem, (int)nb);
          }
          while (c0 + 16 <= nvec) {
            int64_t ncb = (nvec - c0) / 16;
            if (ncb > 8) ncb = 8;
            while (ncb > 1 && wgs * ncb > (int64_t)INT32_MAX) --ncb;
            int rc = launch_mv4<OffT, AT>(plan, A, (const double*)X + c0 * xs1, xs0, xs1, (double*)Y + c0 * ys1, ys0, ys1, (double)alpha, (double)beta, st, 16, (int)ncb);
            if (rc) return rc;
            c0 += 16 * ncb;
          }
          if (c0 == nvec) return KKAMD_OK;
          // the last nvec % 16 columns (or a multivector narrower than 16): one pass of the partial-block form of the kernel
          return launch_mv4<OffT, AT>(plan, A, (const double*)X + c0 * xs1, xs0, xs1, (double*)Y + c0 * ys1, ys0, ys1, (double)alpha, (double)beta, st, (int)(nvec - c0));
        }
      }
    }
    // matrix-core kernel (mv_kernel 0 = auto, or 5; kk_spmv_mvblk.hip): analysed handles, fp64 vectors, matrices whose 16-row tiles share
    // enough columns (block-structured, multi-dof finite elements); any width, any strides.  The analysis happens on the first such call
    if constexpr (sizeof(YT) == 8) {
      if (plan && plan->tile != 0 && plan->tune.mv5 != 0 && (mvk == 0 || mvk == 5) && !plan->mv4 && plan->entries == A->d_entries) {
        if (!plan->mv5 && !plan->mv5_tried) {
          int rc = mv5_plan_build(plan, A, st);
          if (rc) return rc;
        }
        if (plan->mv5) return mv5_spmv(plan, A, (const double*)X, xs0, xs1, (double*)Y, ys0, ys1, nvec, (double)alpha, (double)beta, st);
      }
    }
    const YT* Xr = nullptr; int64_t ldx = 0;
    if (xs1 == 1 && (xs0 % 2 == 0) && ((uintptr_t)X % 16 == 0)) { Xr = X; ldx = xs0; }
    else if (plan && nvec >= 2) {
      // pack X into a row-major workspace owned by the plan (the reference's rank-2 sub-handle tpl_rank2 plays this role)
      const int64_t ldp = (nvec + 1) & ~(int64_t)1;
      const size_t need = (size_t)A->num_cols * (size_t)ldp * sizeof(YT);
      if (plan->xpack_bytes < need) {
        if (plan->d_xpack) { KK_HIP(hipStreamSynchronize(st)); KK_HIP(hipFree(plan->d_xpack)); plan->d_xpack = nullptr; plan->xpack_bytes = 0; }
        KK_HIP(hipMalloc(&plan->d_xpack, need));
        plan->xpack_bytes = need;
      }
      KK_LAUNCH((pack_rows_kernel<YT>), (unsigned)ceil_div(A->num_cols, 32), kBlock, 0, st, A->num_cols, nvec, X, xs0, xs1,
                (YT*)plan->d_xpack, ldp);
      KK_LAUNCH_CHECK();
      Xr = (const YT*)plan->d_xpack; ldx = ldp;
    }
    // LDS-staged X tiles (knob mv_kernel = 3; measured slower than the wave-private kernel on C3): analysed
    // handles, fp64 vectors, 8 or 16 right-hand sides per strip; the analysis happens on the first such call
    if constexpr (sizeof(YT) == 8) {
      if (Xr && plan && plan->tile != 0 && mvk == 3 && nvec >= 8 && nvec % 8 == 0 && !plan->mv_failed && plan->entries == A->d_entries) {
        const int nv = (nvec % 16 == 0) ? 16 : 8;
        if (!plan->mv || plan->mv->nv != nv) {
          int rc = mv_plan_build<OffT>(plan, A, nv, st);
          if (rc) return rc;
        }
        if (plan->mv) {
          if (nv == 16) return launch_mv3<OffT, AT, 16>(plan, A, (const double*)Xr, ldx, (double*)Y, ys0, ys1, nvec, (double)alpha, (double)beta, st);
          return launch_mv3<OffT, AT, 8>(plan, A, (const double*)Xr, ldx, (double*)Y, ys0, ys1, nvec, (double)alpha, (double)beta, st);
        }
      }
    }
    if (Xr) {
      const int yv = (ys1 == 1 && (ys0 % 2 == 0) && ((uintptr_t)Y % 16 == 0)) ? 1 : 0;
      const int mv_remap = plan ? plan->tune.mv_remap : g_spmv_default.mv_remap;
      const bool mv_nt = (plan ? plan->tune.mv_nt : g_spmv_default.mv_nt) != 0;
      int64_t long_T = 0;
      if (plan && plan->row_map == A->d_row_map) {
        if (!plan->mv_long_known) { const int rc = mv_find_long_rows<OffT>(plan, A, st); if (rc) return rc; }
        long_T = plan->n_mv_long > 0 ? plan->mv_long_T : 0;
      }
      // nonzero-split kernel (mv_kernel 0 = auto, or 6; kk_spmv_mvnnz.hip): analysed handles, fp64 vectors; by default on matrices with a
      // share of their nonzeros in long rows (power-law graphs), where a row-based assignment leaves most lanes idle
      if constexpr (sizeof(YT) == 8) {
        if (plan && plan->tile != 0 && plan->tune.mv6 != 0 && (mvk == 0 || mvk == 6) && plan->entries == A->d_entries && plan->row_map == A->d_row_map &&
            (plan->tune.mv6 == 2 || plan->mv_long_nnz * 100 >= (int64_t)plan->tune.mv6_min_long_pct * A->nnz)) {
          if (!plan->mv6 && !plan->mv6_tried) { const int rc = mv6_plan_build(plan, A, st); if (rc) return rc; }
          if (plan->mv6) return mv6_spmv(plan, A, (const double*)Xr, ldx, (double*)Y, ys0, ys1, nvec, (double)alpha, (double)beta, st);
        }
      }
      // rows above long_T entries, after the gather kernel has written beta * Y there (called by the launch macro below)
      auto long_rows = [&]() -> int {
        if (long_T <= 0) return KKAMD_OK;
        const int32_t* d_list = plan->d_mv_long;
        KK_LAUNCH((spmv_mv_long_kernel<OffT, AT, YT>), dim3((unsigned)plan->n_mv_long, (unsigned)ceil_div(nvec, (int64_t)16)), kBlock, 0, st, d_list,
                  (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, Xr, ldx, (int64_t)1, Y, ys0, ys1, nvec, alpha);
        KK_LAUNCH_CHECK();
        return KKAMD_OK;
      };
#define KK_MV2L(L, R, C, N)                                                                                              \
        KK_LAUNCH((spmv_mv2_kernel<OffT, AT, YT, L, R, C, N>), (unsigned)ceil_div(A->num_rows, (kBlock / kWave) * (kWave / L)), kBlock, \
                  0, st, A->num_rows, A->nnz, (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, \
                  Xr, ldx, Y, ys0, ys1, nvec, alpha, beta, yv, mv_remap, mv2_order(plan, A, (kBlock / kWave) * (kWave / L), (int)(nvec < 16 ? nvec : 16) * (int)sizeof(YT), st), long_T)
#define KK_MV2C(L, R, C)                                                                                                 \
      do {                                                                                                               \
        if (mv_nt) KK_MV2L(L, R, C, true); else KK_MV2L(L, R, C, false);                                                  \
        KK_LAUNCH_CHECK();                                                                                               \
        return long_rows();                                                                                              \
      } while (0)
      // staging window: the nnz of the wave's kWave/L rows (+15 % and the 4-alignment slack), rounded up to 256 / 512 / 1024
#define KK_MV2(L, R)                                                                                                     \
      do {                                                                                                               \
        const int64_t need = (int64_t)(1.15 * (double)(kWave / L) * (double)A->nnz / (double)A->num_rows) + 4;            \
        if (need <= 256) KK_MV2C(L, R, 256);                                                                              \
        if (need <= 512) KK_MV2C(L, R, 512);                                                                              \
        KK_MV2C(L, R, 1024);                                                                                              \
      } while (0)
      if (nvec >= 12) KK_MV2(4, 4);
      if (nvec >= 6) KK_MV2(2, 4);
      if (nvec >= 3) KK_MV2(2, 2);
      KK_MV2(1, 2);
#undef KK_MV2C
#undef KK_MV2L
#undef KK_MV2
    }
  }
  if (nvec >= 12) return launch_mv<OffT, AT, YT, 16>(A, X, xs0, xs1, Y, ys0, ys1, nvec, alpha, beta, remap, st);
  if (nvec >= 6)  return launch_mv<OffT, AT, YT, 8>(A, X, xs0, xs1, Y, ys0, ys1, nvec, alpha, beta, remap, st);
  if (nvec >= 3)  return launch_mv<OffT, AT, YT, 4>(A, X, xs0, xs1, Y, ys0, ys1, nvec, alpha, beta, remap, st);
  return launch_mv<OffT, AT, YT, 2>(A, X, xs0, xs1, Y, ys0, ys1, nvec, alpha, beta, remap, st);
}

}  // namespace kk

extern "C" {

int kkamd_spmv_mv(kkamd_spmv_plan_t* plan, const kkamd_crs_t* A, char mode, double alpha, const void* d_X,
                  int64_t x_stride0, int64_t x_stride1, double beta, void* d_Y, int64_t y_stride0, int64_t y_stride1,
                  int64_t nvec, int vector_type, kkamd_stream_t stream) {
  int rc = kk::check_crs(A);
  if (rc) return rc;
  bool trans = false;
  if ((rc = kk::parse_mode(mode, &trans))) return rc;
  if ((rc = kk::check_plan(plan, A))) return rc;
  if (nvec < 0) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_mv: negative number of vectors");
  if (vector_type != KKAMD_F32 && vector_type != KKAMD_F64)
    return kk::fail(KKAMD_ERR_UNSUPPORTED, "kkamd_spmv_mv: unsupported vector_type %d", vector_type);
  hipStream_t st     = kk::to_hip(stream);
  const int64_t ylen = trans ? A->num_cols : A->num_rows;
  if (nvec == 0 || ylen == 0) return KKAMD_OK;
  if (!d_Y) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_mv: null Y");
  if (alpha == 0.0 || A->num_rows == 0 || A->num_cols == 0 || A->nnz == 0) {
    if (vector_type == KKAMD_F64) return kk::launch_scale<double>((double*)d_Y, ylen, y_stride0, nvec, y_stride1, beta, st);
    return kk::launch_scale<float>((float*)d_Y, ylen, y_stride0, nvec, y_stride1, (float)beta, st);
  }
  if (!d_X) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_mv: null X");
  if ((rc = kk::bind_stream(plan, st))) return rc;
  if ((rc = kk::check_entries_content(plan, A, st))) return rc;
  kk::TraceRange range(A->value_type == KKAMD_F64 ? "KokkosSparse::spmv[TPL_KKAMD,double]" : "KokkosSparse::spmv[TPL_KKAMD,float]");
  // one contiguous column: the rank-1 path (sparse/src/KokkosSparse_spmv.hpp:203-217)
  if (nvec == 1 && x_stride0 == 1 && y_stride0 == 1) return kkamd_spmv(plan, A, mode, alpha, d_X, beta, d_Y, vector_type, stream);
  KK_DISPATCH_TYPES(kk::spmv_mv_typed, plan, A, trans, alpha, d_X, x_stride0, x_stride1, beta, d_Y, y_stride0,
                    y_stride1, nvec, st);
}

}  // extern "C"
