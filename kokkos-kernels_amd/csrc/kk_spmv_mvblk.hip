// kk_spmv_mvblk.hip -- rank-2 CSR SpMV (SpMV_MV) on the matrix cores of gfx950, for matrices whose neighbouring rows share columns
// (block-structured and multi-degree-of-freedom finite-element matrices, dense bands).
//
// Reference: sparse/impl/KokkosSparse_spmv_impl.hpp:634-1004 (SPMV_MV_LayoutLeft_Functor: every row gathers the X row of each of its
// entries); the reference's own matrix-core path exists for BSR only (sparse/impl/KokkosSparse_spmv_bsrmatrix_impl.hpp:74-89,
// 16 x 16 x 16 wmma tiles over dense blocks).  Here the CSR matrix keeps its arrays and the ANALYSIS finds the blocks:
//
//   * a TILE is 16 consecutive rows (M of v_mfma_f64_16x16x4f64).  The plan keeps the sorted union U of the tile's columns in BLOCKS
//     of four (K of the instruction; the four columns of a block need not be adjacent) and, per block, a 64-bit mask: bit 4 i + k
//     says that row i of the tile holds column U[4 b + k].  24 bytes per block replace the 4 bytes per entry of `entries`: with every
//     slot filled that is 0.375 B per nonzero, at a quarter filled 1.5 B.
//   * the KERNEL gives a wave a tile: the tile's values (one contiguous piece of the caller's value array -- nothing is copied into the
//     plan, so values may change between calls) are laid down in wave-private LDS with coalesced 16-byte loads; per block, lane
//     (k = lane / 16, j = lane % 16) loads X(U[4 b + k], j) -- the 16 lanes of a k cover one 128-byte X row, four rows per
//     instruction -- and, as lane (i = lane % 16, k), picks A(i, U[4 b + k]) out of LDS: a row's entries ascend, so the entry's place in
//     the row is the number of mask bits of the row seen so far (a running count per lane and a 4-bit popcount).  One MFMA does the
//     16 x 4 x 16 contraction.  An X row is fetched once per TILE instead of once per entry: on a block-diagonal 32 x 32 matrix that is
//     32 gathers instead of 512 per tile, which is what bounds the gather kernel (the texture path).
//   * what the instruction multiplies that the reference does not -- an absent entry (operand 0) with the X value of a column some
//     other row of the tile holds -- is harmless unless that X value is Inf or NaN (0 * Inf).  The lanes watch the exponent of every
//     X value they load; a tile that saw a non-finite one recomputes its rows entry by entry from `entries` (exactly the reference's
//     products), so Inf / NaN propagate to the rows that hold the column and to no other.
//   * tiles the plan cannot describe (a row that does not ascend strictly, more than 2048 entries, fewer than `mv5_min_fill_pct` of the
//     operand slots filled, no entries) leave their rows to a 16-lanes-per-row gather kernel.
//   * LayoutLeft Y: the operands swap roles (D^T = X^T A^T), so that a lane's accumulator registers hold one ROW of the tile in four
//     columns and a store instruction writes four whole 128-byte lines of Y; X is read where it lies for any strides.
#include "kk_spmv_plan.h"
#include "kk_scan.h"
#include <new>
#include <climits>

namespace kk {
typedef int kk_i32x4v __attribute__((vector_size(16)));
constexpr int kMv5Rows = 16;     // rows per tile
constexpr int kMv5Cap  = 2048;   // entries of a described tile (16 KB of values per wave at most)
}  // namespace kk

struct kkamd_mv5_plan {
  int64_t ntiles = 0, tiles_on = 0, nblocks = 0, n_other = 0, nnz_on = 0, rows_on = 0;
  int cap = 0;                                   // LDS values per wave (the largest described tile, rounded up)
  int64_t* d_blk_off = nullptr;                  // [ntiles + 1] first block of every tile (a tile without blocks is not described)
  int32_t* d_cols = nullptr;                     // [4 * nblocks] the union columns, -1 past the end of a tile's union
  unsigned long long* d_masks = nullptr;         // [nblocks]
  int32_t* d_other = nullptr;                    // [n_other] rows of the tiles that are not described
  size_t bytes = 0;
};

namespace kk {

void mv5_plan_destroy(kkamd_mv5_plan* p) {
  if (!p) return;
  if (p->d_blk_off) (void)hipFree(p->d_blk_off);
  if (p->d_cols) (void)hipFree(p->d_cols);
  if (p->d_masks) (void)hipFree(p->d_masks);
  if (p->d_other) (void)hipFree(p->d_other);
  delete p;
}
int64_t mv5_plan_query(const kkamd_mv5_plan* p, int what) {
  if (!p) return 0;
  switch (what) {
    case 0: return p->tiles_on;
    case 1: return p->n_other;
    case 2: return p->nblocks;
    case 3: return (int64_t)p->bytes;
    case 4: return p->nblocks ? (1000 * p->nnz_on) / (64 * p->nblocks) : 0;
    case 5: return p->cap;
    default: return 0;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Analysis, one workgroup per tile.  FILL = false: blk[tile] = number of column blocks (0: the tile is not described) and the
// statistics; FILL = true (blk holds the offsets by then): the union columns and the masks.
// stats: [0] blocks, [1] entries, [2] tiles, [3] rows of the described tiles, [4] most entries in one of them
template <class OffT, bool FILL>
__global__ __launch_bounds__(kBlock) void mv5_tile_kernel(int64_t nrows, const OffT* __restrict__ row_map, const int32_t* __restrict__ entries,
                                                          int min_fill_pct, int64_t* __restrict__ blk, int32_t* __restrict__ cols,
                                                          unsigned long long* __restrict__ masks, unsigned long long* __restrict__ stats, int64_t tile_stride) {
  constexpr int PER = kMv5Cap / kBlock;
  __shared__ int s_key[kMv5Cap];
  __shared__ int s_uni[kMv5Cap];
  __shared__ unsigned s_mask[kMv5Cap / 2];
  __shared__ long long s_rm[kMv5Rows + 1];
  __shared__ int s_bad;
  __shared__ int s_wave[kBlock / 64];
  const int t = threadIdx.x;
  // (stride > 1: the sampling pass of the analysis.  The sample's tiles are JITTERED inside their strides -- a hash of the sample
  // index -- so that a matrix with periodic structure, e.g. interface rows every N tiles of a multi-dof lattice, is not sampled in phase)
  const int64_t tile = tile_stride > 1 ? (int64_t)blockIdx.x * tile_stride + (int64_t)(((unsigned)blockIdx.x * 2654435761u) >> 8) % tile_stride
                                        : (int64_t)blockIdx.x;
  const int64_t row0 = tile * kMv5Rows, rowN = (row0 + kMv5Rows < nrows) ? row0 + kMv5Rows : nrows;
  if (t <= kMv5Rows) s_rm[t] = (long long)row_map[(row0 + t < rowN) ? row0 + t : rowN];
  if (t == 0) s_bad = 0;
  __syncthreads();
  const long long a0 = s_rm[0];
  const long long n64 = s_rm[kMv5Rows] - a0;
  int64_t b0 = 0;
  if (FILL) {
    b0 = blk[tile];
    if (blk[tile + 1] == b0) return;                             // workgroup-uniform
  } else if (n64 == 0 || n64 > kMv5Cap) {
    if (t == 0) blk[tile] = 0;
    return;
  }
  const int n = (int)n64;
  for (int p = t; p < n; p += kBlock) {
    const int key = entries[a0 + p];
    if (!FILL && p > 0) {                                        // a row must ascend strictly: an entry's place in its row is its rank in the union
      bool start = false;
      for (int q = 1; q < kMv5Rows; ++q) start |= (s_rm[q] - a0 == p);
      if (!start && key <= entries[a0 + p - 1]) s_bad = 1;
    }
    if (!FILL && (key < 0 || key == INT_MAX)) s_bad = 1;
    s_key[p] = key;
  }
  __syncthreads();
  // The tile's columns in ascending order.  Its (up to) 16 rows ARE ascending runs, so four rounds of pairwise MERGES do it: an entry's
  // place in the merged pair is its place in its own run plus the number of smaller entries (for the second run: not larger) of the
  // partner, a binary search in LDS -- two barriers per round, eight in all.  (The bitonic network this replaces needed 45 - 66
  // barrier-separated stages for 512 - 2048 keys; a tile whose rows do not ascend is refused above / was refused by the counting
  // pass, so the merge never sees one that matters: its result is not used.)
  {
    int* src = s_key; int* dst = s_uni;
    for (int w = 1; w < kMv5Rows; w <<= 1) {
      int np[PER], kv[PER];
      KK_UNROLL
      for (int q = 0; q < PER; ++q) {
        const int p = t + q * kBlock;
        np[q] = -1; kv[q] = 0;
        if (p < n) {
          int row = 0;                                           // the row that holds position p (rows may be empty): largest r with start(r) <= p
          KK_UNROLL
          for (int r = 1; r < kMv5Rows; ++r) row += ((int)(s_rm[r] - a0) <= p) ? 1 : 0;
          const int pair = row / (2 * w);
          const int ra = pair * 2 * w, rb = ra + w < kMv5Rows ? ra + w : kMv5Rows, rc = ra + 2 * w < kMv5Rows ? ra + 2 * w : kMv5Rows;
          const int A0 = (int)(s_rm[ra] - a0), B0 = (int)(s_rm[rb] - a0), C0 = (int)(s_rm[rc] - a0);
          const int key = src[p];
          int lo, hi;
          if (p < B0) {                                          // first run: entries of the second run that are smaller
            lo = B0; hi = C0;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (src[mid] < key) lo = mid + 1; else hi = mid; }
            np[q] = p + (lo - B0);
          } else {                                               // second run: entries of the first run that are not larger
            lo = A0; hi = B0;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (src[mid] <= key) lo = mid + 1; else hi = mid; }
            np[q] = A0 + (p - B0) + (lo - A0);
          }
          kv[q] = key;
        }
      }
      __syncthreads();
      KK_UNROLL
      for (int q = 0; q < PER; ++q) if (np[q] >= 0) dst[np[q]] = kv[q];
      __syncthreads();
      int* tmp = src; src = dst; dst = tmp;
    }
    // log2(16) = 4 rounds: the result is back in s_key
  }
  // the distinct columns, in order
  int flag[PER], cnt = 0;
  KK_UNROLL
  for (int q = 0; q < PER; ++q) {
    const int p = t * PER + q;
    flag[q] = 0;
    if (p < n) { const int key = s_key[p]; flag[q] = (p == 0 || key != s_key[p - 1]) ? 1 : 0; }
    cnt += flag[q];
  }
  int nU = 0;
  int at = block_exclusive_scan<int>(cnt, &nU, s_wave);
  KK_UNROLL
  for (int q = 0; q < PER; ++q) if (flag[q]) s_uni[at++] = s_key[t * PER + q];
  const int nb = (nU + 3) / 4;
  if (!FILL) {
    __syncthreads();
    if (t == 0) {
      const bool on = !s_bad && (long long)n * 100 >= (long long)min_fill_pct * nb * 64;
      blk[tile] = on ? nb : 0;
      // the totals: here for the sample only (1024 tiles); the full pass leaves them to mv5_stats_kernel -- five atomics per tile on one
      // cache line were 6 of the 7.5 ms this pass took on 125,000 described tiles
      if (on && tile_stride > 1) {
        atomicAdd(&stats[0], (unsigned long long)nb); atomicAdd(&stats[1], (unsigned long long)n); atomicAdd(&stats[2], 1ull);
        atomicAdd(&stats[3], (unsigned long long)(rowN - row0)); atomicMax(&stats[4], (unsigned long long)n);
      }
    }
    return;
  }
  for (int q = t; q < 2 * nb; q += kBlock) s_mask[q] = 0u;
  __syncthreads();
  for (int p = t; p < n; p += kBlock) {
    const int c = entries[a0 + p];
    int i = 0;
    for (int q = 1; q < kMv5Rows; ++q) i += (s_rm[q] - a0 <= p) ? 1 : 0;
    int lo = 0, hi = nU;                                         // the column's place in the union
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_uni[mid] < c) lo = mid + 1; else hi = mid; }
    const int bit = 4 * i + (lo & 3);
    atomicOr(&s_mask[2 * (lo >> 2) + (bit >> 5)], 1u << (bit & 31));
  }
  __syncthreads();
  for (int u = t; u < 4 * nb; u += kBlock) cols[4 * b0 + u] = u < nU ? s_uni[u] : -1;
  for (int b = t; b < nb; b += kBlock) masks[b0 + b] = (unsigned long long)s_mask[2 * b] | ((unsigned long long)s_mask[2 * b + 1] << 32);
}

// totals over the described tiles of the full counting pass (blk[tile] = column blocks of the tile, 0 = not described): [0] blocks,
// [1] entries, [2] tiles, [3] rows, [4] the largest tile's entries -- one set of atomics per workgroup of 256 tiles
template <class OffT>
__global__ __launch_bounds__(kBlock) void mv5_stats_kernel(int64_t ntiles, int64_t nrows, const OffT* __restrict__ row_map, const int64_t* __restrict__ blk,
                                                           unsigned long long* __restrict__ stats) {
  __shared__ unsigned long long s_acc[5];
  const int t = threadIdx.x;
  if (t < 5) s_acc[t] = 0ull;
  __syncthreads();
  const int64_t tile = (int64_t)blockIdx.x * kBlock + t;
  unsigned long long nb = 0, n = 0, on = 0, rows = 0;
  if (tile < ntiles) {
    nb = (unsigned long long)blk[tile];
    if (nb) {
      const int64_t row0 = tile * kMv5Rows, rowN = (row0 + kMv5Rows < nrows) ? row0 + kMv5Rows : nrows;
      n = (unsigned long long)(row_map[rowN] - row_map[row0]); on = 1ull; rows = (unsigned long long)(rowN - row0);
    }
  }
  unsigned long long mx = n;
  for (int o = 32; o > 0; o >>= 1) {
    nb += __shfl_xor(nb, o, 64); on += __shfl_xor(on, o, 64); rows += __shfl_xor(rows, o, 64);
    const unsigned long long m2 = __shfl_xor(mx, o, 64); mx = m2 > mx ? m2 : mx;
    n += __shfl_xor(n, o, 64);
  }
  if ((t & 63) == 0 && on) {
    atomicAdd(&s_acc[0], nb); atomicAdd(&s_acc[1], n); atomicAdd(&s_acc[2], on); atomicAdd(&s_acc[3], rows); atomicMax(&s_acc[4], mx);
  }
  __syncthreads();
  if (t < 4 && s_acc[2]) atomicAdd(&stats[t], s_acc[t]);
  if (t == 4 && s_acc[2]) atomicMax(&stats[4], s_acc[4]);
}

// rows of the tiles that are not described (any order)
__global__ __launch_bounds__(kBlock) void mv5_other_kernel(int64_t nrows, const int64_t* __restrict__ blk_off, int32_t* __restrict__ list,
                                                           unsigned long long* __restrict__ cursor) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= nrows) return;
  const int64_t tile = r / kMv5Rows;
  if (blk_off[tile + 1] == blk_off[tile]) list[atomicAdd(cursor, 1ull)] = (int32_t)r;
}

// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool mv5_nonfinite(double v) {
  return ((unsigned long long)__double_as_longlong(v) & 0x7ff0000000000000ull) == 0x7ff0000000000000ull;
}

// One wave per tile and one wave per workgroup (LDS is then granted per wave: capv values + one chunk of block descriptors, sized by the
// plan's largest described tile, so a CU holds as many tiles in flight as its LDS allows -- the kernel lives on memory latency).
// Per chunk of kMv5Chunk column blocks the descriptors (four columns and a mask per block, contiguous in the plan) arrive with one
// 16-byte and one 8-byte load per lane; per batch of UB blocks all X loads are issued before the first MFMA consumes one.
// NC = blocks of 16 right-hand sides per pass (2: A, the descriptors and the masks are read once for 32 columns);
// ncv = valid right-hand sides of this pass (the spare lanes of a narrower block read its last column and store nothing).
// SWAP: D^T = X^T A^T (column-major Y, see the file header).
constexpr int kMv5Chunk = 64;
template <class OffT, class AT, int NC, bool SWAP>
__global__ __launch_bounds__(kWave) void spmv_mv5_kernel(int64_t nrows, int64_t nnz, const OffT* __restrict__ row_map,
                                                         const int32_t* __restrict__ entries, const AT* __restrict__ values,
                                                         const int64_t* __restrict__ blk_off, const int32_t* __restrict__ cols,
                                                         const unsigned long long* __restrict__ masks, const double* __restrict__ X, int64_t xs0,
                                                         int64_t xs1, double* __restrict__ Y, int64_t ys0, int64_t ys1, double alpha, double beta,
                                                         int ncv, int remap, int capv) {
  using AV = typename vec2<AT>::type;
  constexpr int UB = 16 / NC;                                 // blocks per batch: their X loads are all in flight before the first MFMA (32: no gain on 54-block tiles, 0.20 -> 0.27 ms on 8-block ones)
  KK_DYN_SMEM(char, smem);
  AT* s_val = reinterpret_cast<AT*>(smem);
  int32_t* s_cols = reinterpret_cast<int32_t*>(smem + (size_t)capv * sizeof(double));
  unsigned long long* s_masks = reinterpret_cast<unsigned long long*>(smem + (size_t)capv * sizeof(double) + kMv5Chunk * 16);
  const int lane = threadIdx.x;
  const int64_t tile = xcd_order(blockIdx.x, gridDim.x, remap);
  const int64_t b0 = blk_off[tile], b1 = blk_off[tile + 1];
  if (b0 == b1) return;                                          // not described: its rows are on the gather list
  const int i = lane & 15, kq = lane >> 4;
  const int64_t row0 = tile * kMv5Rows, rowN = (row0 + kMv5Rows < nrows) ? row0 + kMv5Rows : nrows;
  const int64_t rs = (int64_t)row_map[(row0 + i < rowN) ? row0 + i : rowN];
  const int64_t v0 = (int64_t)row_map[row0], v1 = (int64_t)row_map[rowN];
  const int64_t a = v0 & ~(int64_t)1;                            // the value array is 16-byte aligned: pairs start at even entries
  for (int64_t p0 = a + 2 * lane; p0 < v1; p0 += 8 * kWave) {    // four 16-byte loads per lane in flight
    AV v[4];
    KK_UNROLL
    for (int u = 0; u < 4; ++u) {
      const int64_t p = p0 + 2 * kWave * u;
      if (p < v1) {
        if (p + 1 < nnz) v[u] = *reinterpret_cast<const AV*>(values + p);
        else { v[u][0] = values[p]; v[u][1] = AT(0); }             // the array's last entry: no pair to read
      }
    }
    KK_UNROLL
    for (int u = 0; u < 4; ++u) {
      const int64_t p = p0 + 2 * kWave * u;
      if (p < v1) { s_val[p - a] = v[u][0]; s_val[p - a + 1] = v[u][1]; }
    }
  }
  int cur = (int)(rs - a);                                       // where the next entry of the lane's row sits in the wave's LDS
  const double* __restrict__ xcol[NC];
  KK_UNROLL
  for (int q = 0; q < NC; ++q) { const int col = 16 * q + i; xcol[q] = X + (int64_t)(col < ncv ? col : ncv - 1) * xs1; }
  kk_f64x4 acc[NC];
  KK_UNROLL
  for (int q = 0; q < NC; ++q) acc[q] = kk_f64x4{0.0, 0.0, 0.0, 0.0};
  bool bad = false;
  for (int64_t bc = b0; bc < b1; bc += kMv5Chunk) {
    const int nbc = (int)(b1 - bc < kMv5Chunk ? b1 - bc : kMv5Chunk);
    KK_WAVE_SYNC();                                              // the previous chunk's descriptors have been consumed
    {
      const int64_t bl = bc + (lane < nbc ? lane : nbc - 1);
      const kk_i32x4v c4 = *reinterpret_cast<const kk_i32x4v*>(cols + 4 * bl);
      const unsigned long long mk = masks[bl];
      s_cols[4 * lane] = c4[0]; s_cols[4 * lane + 1] = c4[1]; s_cols[4 * lane + 2] = c4[2]; s_cols[4 * lane + 3] = c4[3];
      s_masks[lane] = mk;
    }
    KK_WAVE_SYNC();                                              // ... and, in the first chunk, the tile's values are in place
    for (int bl = 0; bl < nbc; bl += UB) {
      double xv[UB][NC];
      KK_UNROLL
      for (int u = 0; u < UB; ++u) {
        KK_UNROLL
        for (int q = 0; q < NC; ++q) xv[u][q] = 0.0;
        if (bl + u < nbc) {                                      // wave-uniform: nothing is loaded for the blocks past the chunk's end
          const int c = s_cols[4 * (bl + u) + kq];
          KK_UNROLL
          for (int q = 0; q < NC; ++q) if (c >= 0) xv[u][q] = xcol[q][(int64_t)c * xs0];
        }
      }
      KK_UNROLL
      for (int u = 0; u < UB; ++u) {
        if (bl + u < nbc) {                                      // wave-uniform
          const unsigned nib = (unsigned)(s_masks[bl + u] >> (4 * i)) & 15u;
          double av = 0.0;
          if ((nib >> kq) & 1u) av = (double)s_val[cur + __popc(nib & ((1u << kq) - 1u))];
          cur += __popc(nib);
          KK_UNROLL
          for (int q = 0; q < NC; ++q) {
            bad |= mv5_nonfinite(xv[u][q]);
            if (SWAP) acc[q] = KK_MFMA_F64_16X16X4(xv[u][q], av, acc[q]);
            else      acc[q] = KK_MFMA_F64_16X16X4(av, xv[u][q], acc[q]);
          }
        }
      }
    }
  }
  // accumulator register r of lane (i, kq), block q: row 4 r + kq, column 16 q + i -- SWAP: row i, column 16 q + 4 r + kq
  if (__ballot(bad) != 0ull) {
    // an Inf / NaN among the tile's X values: absent entries met it as 0 * Inf.  The reference's products, entry by entry:
    KK_UNROLL
    for (int q = 0; q < NC; ++q) {
      KK_UNROLL
      for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + (SWAP ? i : 4 * r + kq);
        const int col = 16 * q + (SWAP ? 4 * r + kq : i);
        const double* xc = X + (int64_t)(col < ncv ? col : ncv - 1) * xs1;
        double s = 0.0;
        if (row < nrows)
          for (int64_t p = (int64_t)row_map[row]; p < (int64_t)row_map[row + 1]; ++p) s += (double)values[p] * xc[(int64_t)entries[p] * xs0];
        acc[q][r] = s;
      }
    }
  }
  KK_UNROLL
  for (int q = 0; q < NC; ++q) {
    KK_UNROLL
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + (SWAP ? i : 4 * r + kq);
      const int col = 16 * q + (SWAP ? 4 * r + kq : i);
      if (row < nrows && col < ncv) {
        double* yp = Y + row * ys0 + col * ys1;
        const double out = alpha * acc[q][r];
        *yp = (beta == 0.0) ? out : beta * (*yp) + out;
      }
    }
  }
}

// rows of the tiles that are not described: 16 lanes per row (one right-hand side each).  The lanes fetch 16 entries of the row at a
// time (one each), then every lane walks all 16; a product is only added where the row has an entry
template <class OffT, class AT>
__global__ __launch_bounds__(kBlock) void mv5_rows_kernel(int64_t n_list, const int32_t* __restrict__ list, const OffT* __restrict__ row_map,
                                                          const int32_t* __restrict__ entries, const AT* __restrict__ values,
                                                          const double* __restrict__ X, int64_t xs0, int64_t xs1, double* __restrict__ Y,
                                                          int64_t ys0, int64_t ys1, double alpha, double beta, int ncv) {
  int64_t idx = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / 16;
  const int j  = threadIdx.x & 15;
  const int jc = j < ncv ? j : ncv - 1;
  const bool live = idx < n_list;                                // no early return: the shuffles below want whole waves
  if (!live) idx = n_list - 1;
  const int64_t r = list[idx];
  const int64_t b = (int64_t)row_map[r], e = (int64_t)row_map[r + 1];
  // the longest row of the wave decides the trip count (the shuffles are wave-wide rendezvous under the emulator)
  int64_t len = e - b;
  for (int o = 16; o < 64; o <<= 1) { const long long other = __shfl_xor((long long)len, o, 64); len = other > len ? other : len; }
  double acc = 0.0;
  for (int64_t a = b; a < b + len; a += 16) {
    const bool in = a + j < e;
    const int32_t my_col = in ? entries[a + j] : 0;
    const double my_val  = in ? (double)values[a + j] : 0.0;
    KK_UNROLL
    for (int q = 0; q < 16; ++q) {
      const int32_t col = __shfl(my_col, q, 16);
      const double v    = __shfl(my_val, q, 16);
      if (a + q < e) acc += v * X[(int64_t)col * xs0 + jc * xs1];
    }
  }
  if (!live || j >= ncv) return;
  double* yp = Y + r * ys0 + j * ys1;
  *yp = (beta == 0.0) ? alpha * acc : beta * (*yp) + alpha * acc;
}

// ------------------------------------------------------------------------------------------------------------------------------
template <class OffT>
static int mv5_plan_build_t(kkamd_spmv_plan* plan, const kkamd_crs_t* A, hipStream_t st) {
  plan->mv5_tried = true;
  if (A->num_rows < kMv5Rows || A->nnz == 0 || A->num_rows > (int64_t)INT_MAX) return KKAMD_OK;
  const int mode = plan->tune.mv5;
  const int64_t ntiles = ceil_div(A->num_rows, kMv5Rows);
  if (ntiles > (int64_t)INT_MAX) return KKAMD_OK;
  kkamd_mv5_plan* p = new (std::nothrow) kkamd_mv5_plan();
  if (!p) return fail(KKAMD_ERR_ALLOC, "kkamd_spmv_mv: out of host memory");
  struct Guard { kkamd_mv5_plan* p; ~Guard() { if (p) mv5_plan_destroy(p); } } guard{p};
  p->ntiles = ntiles;
  DevBuf stats;
  if (hipMalloc((void**)&p->d_blk_off, sizeof(int64_t) * (size_t)(ntiles + 1)) != hipSuccess || stats.alloc(8 * sizeof(unsigned long long)) != hipSuccess) {
    (void)hipGetLastError();
    return KKAMD_OK;                                             // no memory for the analysis: the gather kernel serves the matrix
  }
  unsigned long long* d_stats = stats.as<unsigned long long>();
  KK_HIP(hipMemsetAsync(d_stats, 0, 8 * sizeof(unsigned long long), st));
  KK_HIP(hipMemsetAsync(p->d_blk_off + ntiles, 0, sizeof(int64_t), st));
  const int min_fill = mode == 2 ? 0 : plan->tune.mv5_min_fill_pct;
  if (mode != 2 && ntiles > 4096) {
    // a sample first (1024 tiles spread over the matrix, no fill threshold): when the tiles that can be described at all fill less than
    // half of what the threshold asks for, the matrix is not for this kernel and the full pass (a sort per tile: 3 ms on 5e6 rows x 20) is skipped
    const int64_t stride = ntiles / 1024;
    KK_LAUNCH((mv5_tile_kernel<OffT, false>), 1024u, kBlock, 0, st, A->num_rows, (const OffT*)A->d_row_map, (const int32_t*)A->d_entries,
              0, p->d_blk_off, (int32_t*)nullptr, (unsigned long long*)nullptr, d_stats, stride);
    KK_LAUNCH_CHECK();
    unsigned long long hs[8];
    KK_HIP(hipMemcpyAsync(hs, d_stats, sizeof hs, hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
    const double fill = hs[0] ? (double)hs[1] / (64.0 * (double)hs[0]) : 0.0;
    if (g_verbose) printf("kkamd_spmv_mv: matrix-core analysis, sample of 1024 tiles: %llu can be described, fill %.3f\n", hs[2], fill);
    if (hs[2] < 256 || fill * 100.0 < 0.5 * (double)min_fill) return KKAMD_OK;
    KK_HIP(hipMemsetAsync(d_stats, 0, 8 * sizeof(unsigned long long), st));
  }
  KK_LAUNCH((mv5_tile_kernel<OffT, false>), (unsigned)ntiles, kBlock, 0, st, A->num_rows, (const OffT*)A->d_row_map, (const int32_t*)A->d_entries,
            min_fill, p->d_blk_off, (int32_t*)nullptr, (unsigned long long*)nullptr, d_stats, (int64_t)1);
  KK_LAUNCH_CHECK();
  KK_LAUNCH((mv5_stats_kernel<OffT>), (unsigned)ceil_div(ntiles, kBlock), kBlock, 0, st, ntiles, A->num_rows, (const OffT*)A->d_row_map,
            (const int64_t*)p->d_blk_off, d_stats);
  KK_LAUNCH_CHECK();
  int rc = exclusive_scan_inplace<int64_t>(p->d_blk_off, ntiles + 1, st);
  if (rc) return rc;
  unsigned long long h[8];
  KK_HIP(hipMemcpyAsync(h, d_stats, sizeof h, hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  p->nblocks = (int64_t)h[0]; p->nnz_on = (int64_t)h[1]; p->tiles_on = (int64_t)h[2]; p->rows_on = (int64_t)h[3];
  p->n_other = A->num_rows - p->rows_on;
  if (g_verbose)
    printf("kkamd_spmv_mv: matrix-core analysis: %lld of %lld tiles described, %lld column blocks, fill %.3f, %lld rows left to the gather rows\n",
           (long long)p->tiles_on, (long long)ntiles, (long long)p->nblocks, p->nblocks ? (double)p->nnz_on / (64.0 * (double)p->nblocks) : 0.0,
           (long long)p->n_other);
  if (p->tiles_on == 0) return KKAMD_OK;
  if (mode != 2 && p->n_other * 100 > (int64_t)plan->tune.mv5_max_other_pct * A->num_rows) return KKAMD_OK;
  p->cap = (int)((h[4] + 2 + 63) / 64 * 64);                     // LDS values per wave: the largest described tile (+ the pair that starts one entry early)
  if (hipMalloc((void**)&p->d_cols, sizeof(int32_t) * 4 * (size_t)p->nblocks) != hipSuccess ||
      hipMalloc((void**)&p->d_masks, sizeof(unsigned long long) * (size_t)p->nblocks) != hipSuccess ||
      (p->n_other > 0 && hipMalloc((void**)&p->d_other, sizeof(int32_t) * (size_t)p->n_other) != hipSuccess)) {
    (void)hipGetLastError();
    return KKAMD_OK;
  }
  KK_LAUNCH((mv5_tile_kernel<OffT, true>), (unsigned)ntiles, kBlock, 0, st, A->num_rows, (const OffT*)A->d_row_map, (const int32_t*)A->d_entries,
            min_fill, p->d_blk_off, p->d_cols, p->d_masks, d_stats, (int64_t)1);
  KK_LAUNCH_CHECK();
  if (p->n_other > 0) {
    KK_HIP(hipMemsetAsync(d_stats, 0, sizeof(unsigned long long), st));
    KK_LAUNCH((mv5_other_kernel), (unsigned)ceil_div(A->num_rows, kBlock), kBlock, 0, st, A->num_rows, (const int64_t*)p->d_blk_off, p->d_other, d_stats);
    KK_LAUNCH_CHECK();
  }
  KK_HIP(hipStreamSynchronize(st));
  p->bytes = sizeof(int64_t) * (size_t)(ntiles + 1) + 24 * (size_t)p->nblocks + sizeof(int32_t) * (size_t)p->n_other;
  plan->mv5 = p;
  guard.p = nullptr;
  return KKAMD_OK;
}

int mv5_plan_build(kkamd_spmv_plan* plan, const kkamd_crs_t* A, hipStream_t st) {
  return A->offset_type == KKAMD_I64 ? mv5_plan_build_t<int64_t>(plan, A, st) : mv5_plan_build_t<int32_t>(plan, A, st);
}

template <class OffT, class AT>
static int mv5_launch(const kkamd_spmv_plan* plan, const kkamd_crs_t* A, const double* X, int64_t xs0, int64_t xs1, double* Y, int64_t ys0, int64_t ys1,
                      int64_t nvec, double alpha, double beta, hipStream_t st) {
  const kkamd_mv5_plan* p = plan->mv5;
  const unsigned grid = (unsigned)p->ntiles;
  const bool swap = ys0 < ys1;                                   // column-major Y
  const int remap = plan->tune.mv_remap;
  const size_t lds = (size_t)p->cap * sizeof(double) + kMv5Chunk * 24;
  for (int64_t c0 = 0; c0 < nvec;) {
    const int nc = (nvec - c0 > 16) ? 2 : 1;                     // 17..32 columns left: one pass over A for two blocks of 16
    const int ncv = (int)(nvec - c0 < 16 * nc ? nvec - c0 : 16 * nc);
    const double* Xb = X + c0 * xs1;
    double* Yb = Y + c0 * ys1;
#define KK_MV5(NC_, SW)                                                                                                             \
    KK_LAUNCH((spmv_mv5_kernel<OffT, AT, NC_, SW>), grid, kWave, lds, st, A->num_rows, A->nnz, (const OffT*)A->d_row_map,             \
              (const int32_t*)A->d_entries, (const AT*)A->d_values, (const int64_t*)p->d_blk_off, (const int32_t*)p->d_cols,          \
              (const unsigned long long*)p->d_masks, Xb, xs0, xs1, Yb, ys0, ys1, alpha, beta, ncv, remap, p->cap)
    if (nc == 2) { if (swap) KK_MV5(2, true); else KK_MV5(2, false); }
    else         { if (swap) KK_MV5(1, true); else KK_MV5(1, false); }
#undef KK_MV5
    KK_LAUNCH_CHECK();
    if (p->n_other > 0) {
      for (int q = 0; q < nc && c0 + 16 * q < nvec; ++q) {
        const int nq = (int)(nvec - c0 - 16 * q < 16 ? nvec - c0 - 16 * q : 16);
        KK_LAUNCH((mv5_rows_kernel<OffT, AT>), (unsigned)ceil_div(p->n_other * 16, kBlock), kBlock, 0, st, p->n_other, (const int32_t*)p->d_other,
                  (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, Xb + 16 * q * xs1, xs0, xs1, Yb + 16 * q * ys1, ys0,
                  ys1, alpha, beta, nq);
        KK_LAUNCH_CHECK();
      }
    }
    c0 += 16 * nc;
  }
  return KKAMD_OK;
}

int mv5_spmv(const kkamd_spmv_plan* plan, const kkamd_crs_t* A, const double* X, int64_t xs0, int64_t xs1, double* Y, int64_t ys0, int64_t ys1,
             int64_t nvec, double alpha, double beta, hipStream_t st) {
  const bool o64 = A->offset_type == KKAMD_I64;
  if (A->value_type == KKAMD_F64)
    return o64 ? mv5_launch<int64_t, double>(plan, A, X, xs0, xs1, Y, ys0, ys1, nvec, alpha, beta, st)
               : mv5_launch<int32_t, double>(plan, A, X, xs0, xs1, Y, ys0, ys1, nvec, alpha, beta, st);
  return o64 ? mv5_launch<int64_t, float>(plan, A, X, xs0, xs1, Y, ys0, ys1, nvec, alpha, beta, st)
             : mv5_launch<int32_t, float>(plan, A, X, xs0, xs1, Y, ys0, ys1, nvec, alpha, beta, st);
}

}  // namespace kk
