// kk_spmv.hip -- CSR SpMV for gfx950 (MI355X), rank-1 and rank-2.
//
// What the reference does on a GPU (sparse/impl/KokkosSparse_spmv_impl.hpp:86-166,337-379): one
// Kokkos "thread" of vl lanes per row, 256/vl rows per team, x gathered through the texture path, no
// LDS.  For the 27-pt Laplacian that is 8 lanes x 4 strided trips per row and 843,750 tiny teams.
//
// What this file does instead (both memory-bound; roofline = HBM):
//   spmv_stream_kernel  -- the planned path.  The nnz stream is cut into fixed TILE-sized pieces, one
//       workgroup each, so col_idx/values are read with perfectly aligned, fully coalesced 16 B / 8 B
//       per-lane loads no matter how rows fall; val*x[col] products are staged in LDS; rows are then
//       reduced out of LDS by lane groups sized to the tile's row count (wave shuffles for the
//       cross-lane part).  Rows that straddle tiles leave a head/tail partial in a carry array and a
//       tiny fix-up kernel finishes them -- no fp64 atomics, no beta pre-pass over y
//       (the reference's merge path needs both: sparse/impl/KokkosSparse_spmv_impl_merge.hpp:278-282).
//       Workgroup -> tile mapping is XCD-aware so each XCD's 4 MiB L2 keeps its own window of x.
//   spmv_vector_kernel  -- the no-analysis path (handle-less calls / SPMV_FAST_SETUP): LPR lanes per
//       row, LPR chosen from nnz/row.
//   spmv_transpose_kernel -- op(A) = A^T via fp64/fp32 hardware atomics after a beta pass (same
//       structure as the reference's SPMV_Transpose_Functor, :36-84; "functional, not performant").
//   spmv_mv_kernel      -- rank-2: A rows staged through LDS once per 16-column strip, one lane per
//       right-hand side, X rows read as contiguous 128 B when X is row-major.
#include "kk_common.h"
#include "kk_scan.h"
#include <new>
#include <cstring>
#include <climits>

#include "kk_spmv_plan.h"

namespace kk {

// ------------------------------------------------------------------------------------------------
// vector kernel: LPR lanes cooperate on a row (K1 analogue).
template <class OffT, class AT, class YT, int LPR>
__global__ __launch_bounds__(kBlock) void spmv_vector_kernel(int64_t nrows, const OffT* __restrict__ row_map,
                                                             const int32_t* __restrict__ entries,
                                                             const AT* __restrict__ values, const YT* __restrict__ x,
                                                             YT* __restrict__ y, YT alpha, YT beta, int remap) {
  constexpr int RPB = kBlock / LPR;
  const int64_t wg  = remap ? xcd_remap(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
  const int64_t row = wg * RPB + threadIdx.x / LPR;
  const int lane    = threadIdx.x % LPR;
  YT sum            = YT(0);
  if (row < nrows) {
    const OffT s = row_map[row], e = row_map[row + 1];
    for (OffT j = s + lane; j < e; j += LPR) sum += (YT)values[j] * x[entries[j]];
  }
  sum = group_sum(sum, LPR);
  if (row < nrows && lane == 0) {
    sum *= alpha;
    y[row] = (beta == YT(0)) ? sum : beta * y[row] + sum;
  }
}

// ------------------------------------------------------------------------------------------------
// plan analysis: blk_row[b] = first row whose start offset is >= b*tile (lower bound over row_map).
template <class OffT>
__global__ void spmv_plan_kernel(int64_t nrows, const OffT* __restrict__ row_map, int64_t nblocks, int64_t tile,
                                 int32_t* __restrict__ blk_row) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nblocks) return;
  if (b == nblocks) { blk_row[b] = (int32_t)nrows; return; }
  const int64_t target = b * tile;
  int64_t lo = 0, hi = nrows + 1;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)row_map[mid] < target) lo = mid + 1; else hi = mid;
  }
  // bit 31: the tile starts inside a row (row_map[lo] > target), i.e. it has a head segment
  const bool head = (int64_t)row_map[lo] > target;
  blk_row[b] = (int32_t)lo | (head ? (int32_t)0x80000000 : 0);
}

// Streaming loads of one tile into registers and staging of the val*x products in LDS.  FULL (the whole
// tile lies inside [0, nnz): every tile but the last) is a workgroup-uniform property; its code path has
// no per-lane guards, so all STEPS independent 16 B + 8 B loads are issued back to back and stay in
// flight together (guarded loads made the compiler drain vmcnt between steps -- 4x fewer bytes in flight).
template <class AT, int STEPS, bool NT, bool FULL>
__device__ __forceinline__ void load_tile(const AT* __restrict__ values, const int32_t* __restrict__ entries, int64_t ts,
                                          int64_t te, int t, AT (&v0)[STEPS], AT (&v1)[STEPS], int (&c0)[STEPS],
                                          int (&c1)[STEPS]) {
  using AV = typename vec2<AT>::type;
  constexpr int SPAN = kBlock * 2;
  KK_UNROLL
  for (int k = 0; k < STEPS; ++k) {
    const int64_t idx = ts + (int64_t)k * SPAN + t * 2;
    if (FULL || idx + 1 < te) {
      const AV* vp       = reinterpret_cast<const AV*>(values + idx);
      const kk_i32x2* cp = reinterpret_cast<const kk_i32x2*>(entries + idx);
      const AV vv        = NT ? KK_NT_LOAD(vp) : *vp;
      const kk_i32x2 cc  = NT ? KK_NT_LOAD(cp) : *cp;
      v0[k] = vv[0]; v1[k] = vv[1]; c0[k] = cc[0]; c1[k] = cc[1];
    } else if (idx < te) {
      v0[k] = values[idx]; c0[k] = entries[idx]; v1[k] = AT(0); c1[k] = c0[k];
    } else {
      v0[k] = v1[k] = AT(0); c0[k] = c1[k] = -1;
    }
  }
}

// Window codes (stream_variant 6).  The column indices are 4 of the 12 bytes per nonzero the kernel streams.  On
// matrices whose tiles touch few column neighbourhoods (stencils, banded and block-structured matrices) a tile's columns
// fit into <= 16 windows of 4096 consecutive columns; the plan then keeps, per nonzero, a 16-bit code = window << 12 |
// (column - window base) and per tile the 16 window bases: 10 instead of 12 bytes per nonzero.  Codes are stored in the
// order work-item t consumes them (its 2*STEPS codes are contiguous: one or two 16-byte loads instead of STEPS 8-byte
// ones), the bases sit in lanes 0-15 of every wave and are fetched with one lane permute per nonzero.
constexpr int kWinBits = 12, kWinCount = 16;
// Per tile the plan keeps 64 ints ("window meta", one coalesced load per wave): [0,16) window bases, [16,32) the LDS slot
// of each window's first used column, [32,64) the first column of every 64-slot chunk of the staged x window (-1 = unused).
// Staged x (stream_variant 6/1, every tile's used column ranges -- padded to 64 -- fit the tile's LDS): the x entries a
// tile needs are contiguous ranges, so they are fetched with coalesced loads issued TOGETHER with the value loads and the
// per-nonzero gather reads LDS: no dependent trip to memory, and ~10x fewer cache-line look-ups than the gather.
constexpr int kWinMeta = 64, kWinChunks = 32;

template <int NPT>
__global__ __launch_bounds__(kBlock) void win_build_kernel(int64_t nnz, const int32_t* __restrict__ entries,
                                                           uint16_t* __restrict__ wcode, int32_t* __restrict__ wmeta,
                                                           int32_t* __restrict__ tinfo, int allow_stage) {
  // Per tile (tinfo[b]): kTilePlain when the tile needs more than 16 windows (it keeps reading entries), kTileCodes when
  // its used column ranges exceed the LDS x window, kTileStaged otherwise (tile_mode_hist_kernel counts the modes afterwards).
  constexpr int TILE = kBlock * NPT, STEPS = NPT / 2, SPAN = kBlock * 2, NW = kBlock / 64;
  __shared__ int s_base[kWinCount];
  __shared__ int s_nch[kWinCount];                           // 64-column chunks of every window up to its last used one
  __shared__ int s_off[kWinCount + 1];
  __shared__ int s_wmin[NW];                                 // per wave: smallest uncovered column / used chunks of the window being formed
  __shared__ unsigned long long s_wmask[NW];
  __shared__ int s_flag;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t b = blockIdx.x, s = b * TILE;
  int c[NPT];
  KK_UNROLL
  for (int k = 0; k < STEPS; ++k) {
    const int64_t idx = s + (int64_t)k * SPAN + t * 2;
    c[2 * k]     = idx < nnz ? entries[idx] : -1;
    c[2 * k + 1] = idx + 1 < nnz ? entries[idx + 1] : -1;
  }
  // Greedy cover of the tile's columns, left to right: the next window starts at the smallest column not covered yet.
  // Pass 0 ends a window at the first run of 64 unused columns (64-aligned from its base), so that windows hug the
  // contiguous column runs a tile really touches (27-pt: nine runs of ~80) and the staged x window stays small; if that
  // needs more than 16 windows, pass 1 takes full 4096-column windows.
  // Both questions -- the smallest uncovered column, the used 64-column chunks of the window (a 64-bit mask: 4096 / 64) -- are answered
  // in registers, reduced across the wave with lane exchanges and across the four waves through four LDS words: two barriers per
  // window.  (Round 4 marked every used column in an LDS bitmap and took the window lengths with an LDS atomic per entry: a tile's
  // entries sit in a few words, so the 64 lanes of every ds_or / ds_max went through the LDS one after the other -- 4 of the 6.7 ms
  // this kernel took on C2; and since a window's slots are padded to whole chunks, only the chunk count was ever used.)
  bool uncovered = false;
  for (int pass = 0; pass < 2; ++pass) {                       // workgroup-uniform control flow throughout
    long long bound = 0;                                       // columns < bound are covered
    for (int w = 0; w < kWinCount; ++w) {
      int m = INT_MAX;
      KK_UNROLL
      for (int k = 0; k < NPT; ++k) if (c[k] >= 0 && (long long)c[k] >= bound && c[k] < m) m = c[k];
      for (int o = 32; o > 0; o >>= 1) { const int m2 = __shfl_xor(m, o, 64); m = m2 < m ? m2 : m; }
      if (lane == 0) s_wmin[wave] = m;
      __syncthreads();
      int base = s_wmin[0];
      KK_UNROLL
      for (int q = 1; q < NW; ++q) base = s_wmin[q] < base ? s_wmin[q] : base;
      if (base == INT_MAX) {                                   // everything is covered: the unused windows repeat the last base
        if (t == 0) for (int q = w; q < kWinCount; ++q) { s_base[q] = q ? s_base[q - 1] : 0; s_nch[q] = 0; }
        break;
      }
      unsigned long long used = 0ull;
      KK_UNROLL
      for (int k = 0; k < NPT; ++k) {
        const long long d = (long long)c[k] - base;
        if (c[k] >= 0 && d >= 0 && d < (1 << kWinBits)) used |= 1ull << (d >> 6);
      }
      for (int o = 32; o > 0; o >>= 1) used |= __shfl_xor(used, o, 64);
      if (lane == 0) s_wmask[wave] = used;
      __syncthreads();
      used = s_wmask[0];
      KK_UNROLL
      for (int q = 1; q < NW; ++q) used |= s_wmask[q];
      int nch;                                                 // chunk 0 holds the base itself: nch >= 1
      if (pass == 0) nch = (~used == 0ull) ? 64 : __ffsll(~used) - 1;                 // up to the first unused chunk
      else           nch = 64 - __clzll((long long)used);                             // up to the last used one
      if (t == 0) { s_base[w] = base; s_nch[w] = nch; }
      bound = (long long)base + (pass == 0 ? 64ll * nch : (long long)(1 << kWinBits));
    }
    uncovered = false;
    KK_UNROLL
    for (int k = 0; k < NPT; ++k) uncovered |= (c[k] >= 0 && (long long)c[k] >= bound);
    __syncthreads();                                           // (also: s_base / s_nch of the last window, and the words of the break above)
    if (t == 0) s_flag = 0;
    __syncthreads();
    if (__ballot(uncovered) != 0ull && lane == 0) s_flag = 1;
    __syncthreads();
    const int any = s_flag;
    if (!any) break;
  }
  if (s_flag != 0) {                                           // workgroup-uniform: the last pass left columns uncovered
    if (t == 0) tinfo[b] = kTilePlain;
    return;
  }
  int sb[kWinCount];
  KK_UNROLL
  for (int q = 0; q < kWinCount; ++q) sb[q] = s_base[q];
  unsigned code[NPT];
  KK_UNROLL
  for (int k = 0; k < NPT; ++k) {
    int w = 0;
    KK_UNROLL
    for (int q = 1; q < kWinCount; ++q) if (sb[q] <= c[k] && sb[q] > sb[q - 1]) w = q;
    int bw = sb[0];
    KK_UNROLL
    for (int q = 1; q < kWinCount; ++q) bw = (w == q) ? sb[q] : bw;
    const int d = c[k] - bw;
    const bool ok = (c[k] >= 0 && d >= 0 && d < (1 << kWinBits));
    code[k] = ok ? (unsigned)((w << kWinBits) | d) : 0u;
  }
  // work-item t's NPT codes are contiguous (2 NPT bytes): 16-byte stores where the count allows
  uint16_t* out = wcode + s + (int64_t)t * NPT;
  if constexpr (NPT % 8 == 0) {
    KK_UNROLL
    for (int k = 0; k < NPT; k += 8) {
      kk_u32x4 v;
      v[0] = code[k] | (code[k + 1] << 16); v[1] = code[k + 2] | (code[k + 3] << 16);
      v[2] = code[k + 4] | (code[k + 5] << 16); v[3] = code[k + 6] | (code[k + 7] << 16);
      *reinterpret_cast<kk_u32x4*>(out + k) = v;
    }
  } else {
    KK_UNROLL
    for (int k = 0; k < NPT; k += 2) *reinterpret_cast<unsigned*>(out + k) = code[k] | (code[k + 1] << 16);
  }
  // LDS slots of the used part of every window, in whole 64-slot chunks
  if (t == 0) {
    int off = 0;
    for (int w = 0; w < kWinCount; ++w) { s_off[w] = off; off += 64 * s_nch[w]; }
    s_off[kWinCount] = off;
    constexpr int CAP = TILE < kWinChunks * 64 ? TILE : kWinChunks * 64;
    const int mode = (off > CAP || !allow_stage) ? kTileCodes : kTileStaged;
    tinfo[b] = mode;
  }
  __syncthreads();
  int32_t* meta = wmeta + b * kWinMeta;
  if (t < kWinCount) { meta[t] = s_base[t]; meta[kWinCount + t] = s_off[t]; }
  if (t < kWinChunks) {
    const int slot = t * 64;
    int col = -1;
    for (int w = 0; w < kWinCount; ++w) if (slot >= s_off[w] && slot < s_off[w] + 64 * s_nch[w]) col = s_base[w] + (slot - s_off[w]);
    meta[2 * kWinCount + t] = col;
  }
}

// Row-pattern codes (on top of the staged x window).  On locally Toeplitz matrices -- stencils: column = row + a constant
// per diagonal -- the LDS slot of entry k of a row is the slot of entry k of the row before, plus one.  A tile then
// decomposes into a few SEGMENTS of consecutive rows of equal length L that share one slot table T[0..L): nonzero i of the
// tile belongs to segment g (its start sb_g <= i), j = i - (start of the segment's first row), row = j / L, k = j % L and
// its x entry sits in LDS slot T_g[k] + row.  Per tile that is kPatW ints instead of 2 bytes per nonzero.  Tiles that do not
// decompose into <= kPatSeg segments of rows with 1..kPatLen entries keep their 16-bit codes (nseg = 0).
// Tile record (ints): [0] nseg, [1..kPatSeg-1] starts of segments 1.. (INT_MAX when unused; segment 0 starts at 0), [8 + 4g ..] {start,
// -first row start, L, float 1 / L}, then the slot tables as signed 16-bit values (-1 <= slot < 4096): T_g[k] = short[32 g + k] behind
// int kPatTab.  672 bytes per tile.
constexpr int kPatSeg = 8, kPatLen = 32, kPatRec = 8, kPatTab = kPatRec + 4 * kPatSeg, kPatW = kPatTab + kPatSeg * kPatLen / 2;

template <class OffT, int NPT>
__global__ __launch_bounds__(kBlock) void pat_build_kernel(int64_t nnz, const OffT* __restrict__ row_map,
                                                           const int32_t* __restrict__ blk_info,
                                                           const uint16_t* __restrict__ wcode, const int32_t* __restrict__ wmeta,
                                                           int32_t* __restrict__ tinfo, int32_t* __restrict__ pmeta) {
  constexpr int TILE = kBlock * NPT, STEPS = NPT / 2, SPAN = kBlock * 2;
  __shared__ unsigned short s_slot[TILE];
  __shared__ unsigned char s_head[TILE + 2];
  __shared__ int s_segq[kPatSeg];
  __shared__ int s_nseg, s_bad, s_chunks;
  const int t = threadIdx.x;
  const int64_t b = blockIdx.x, s = b * TILE;
  int32_t* out = pmeta + b * kPatW;
  if ((tinfo[b] & 3) != kTileStaged) return;                   // workgroup-uniform: only staged-x tiles can carry a record
  if (t == 0) { s_nseg = 0; s_bad = (s + TILE <= nnz) ? 0 : 1; s_chunks = 0; }    // the ragged last tile keeps its codes
  __syncthreads();
  // LDS slot of every nonzero, in tile order (the codes are stored in work-item order)
  const int ldsoff = wmeta[b * kWinMeta + kWinCount + (t & (kWinCount - 1))];
  KK_UNROLL
  for (int k = 0; k < STEPS; ++k) {
    KK_UNROLL
    for (int j = 0; j < 2; ++j) {
      const unsigned code = wcode[s + (int64_t)t * NPT + 2 * k + j];
      const int off = __shfl(ldsoff, (int)(code >> kWinBits), 64);
      s_slot[k * SPAN + 2 * t + j] = (unsigned short)(off + (int)(code & ((1u << kWinBits) - 1)));
    }
  }
  if (t < kWinChunks && wmeta[b * kWinMeta + 2 * kWinCount + t] >= 0) atomicAdd(&s_chunks, 1);
  const int32_t info0 = blk_info[b], info1 = blk_info[b + 1];
  const int64_t ra = info0 & 0x7fffffff, rb = info1 & 0x7fffffff;
  const int has_head = (info0 >> 31) & 1;
  const int64_t nv = (rb - ra) + has_head;                   // rows with at least their start or their end in this tile
  __syncthreads();
  // the tail of the product array holds this record at run time: the x window must leave it free (float products: 4 B slots)
  if (t == 0 && s_chunks * 64 > TILE - kPatW) s_bad = 1;
  for (int64_t q = t; q < nv; q += kBlock) {
    const int64_t r  = ra + q - has_head;
    const int64_t rs = (int64_t)row_map[r] - s, re = (int64_t)row_map[r + 1] - s, L = re - rs;
    bool head = true;
    if (L < 1 || L > kPatLen) atomicOr(&s_bad, 1);
    else if (q > 0) {
      const int64_t ps = (int64_t)row_map[r - 1] - s;
      if (rs - ps == L) {
        head = false;
        for (int k = 0; k < (int)L; ++k) {
          const int64_t i0 = ps + k, i1 = rs + k;
          if (i0 >= 0 && i1 < TILE && (int)s_slot[i1] != (int)s_slot[i0] + 1) head = true;
        }
      }
    }
    s_head[q] = head ? 1 : 0;
    if (head) { const int idx = atomicAdd(&s_nseg, 1); if (idx < kPatSeg) s_segq[idx] = (int)q; }
  }
  __syncthreads();
  const int nseg = s_nseg;
  if (s_bad || nseg > kPatSeg || nseg < 1) return;            // workgroup-uniform: the tile keeps its codes
  if (t == 0) {                                               // segment heads in row order
    for (int a = 1; a < nseg; ++a) { const int v = s_segq[a]; int c = a - 1; while (c >= 0 && s_segq[c] > v) { s_segq[c + 1] = s_segq[c]; --c; } s_segq[c + 1] = v; }
  }
  __syncthreads();
  if (t < kPatSeg) {
    int sb = INT_MAX, rs32 = 0, L32 = 1; float M = 1.0f;
    if (t < nseg) {
      const int64_t r  = ra + s_segq[t] - has_head;
      const int64_t rs = (int64_t)row_map[r] - s, L = (int64_t)row_map[r + 1] - s - rs;
      sb = rs > 0 ? (int)rs : 0; rs32 = (int)rs; L32 = (int)L;
      M  = 1.0f / (float)L;                                    // floor(j / L) = (int)((j + 0.5f) * M), exact for j < 2^16, L <= 32
    }
    if (t >= 1) out[t] = sb;                                  // starts of segments 1..7 (segment 0 starts at 0)
    out[kPatRec + 4 * t + 0] = sb; out[kPatRec + 4 * t + 1] = -rs32; out[kPatRec + 4 * t + 2] = L32; out[kPatRec + 4 * t + 3] = __float_as_int(M);
  }
  if (t == 0) { out[0] = nseg; tinfo[b] = kTilePattern; }
  if (t < kPatSeg * kPatLen) {
    const int g = t / kPatLen, k = t % kPatLen;
    int val = 0;
    if (g < nseg) {
      const int q = s_segq[g];
      const int64_t r  = ra + q - has_head;
      const int64_t rs = (int64_t)row_map[r] - s, re = (int64_t)row_map[r + 1] - s;
      if (k < re - rs) {
        const int64_t i = rs + k;
        if (i >= 0 && i < TILE) val = (int)s_slot[i];
        else if (i < 0 && q + 1 < nv && !s_head[q + 1] && re + k < TILE) val = (int)s_slot[re + k] - 1;   // from the next row of the segment
      }
    }
    reinterpret_cast<short*>(out + kPatTab)[t] = (short)val;     // may be -1: the slot before a row that starts in an earlier tile
  }
}

// The row-pattern record of a tile, DIRECTLY from the matrix (round 6).  On a stencil matrix nearly every tile ends as a pattern tile, and
// the way there -- win_build_kernel (greedy window cover over the tile's 4096 entries by all four waves, two barriers per window, then a code
// per entry: 10,000 vector instructions per tile, 4.3 of the 6.2 ms a plan of C2 took) followed by pat_build_kernel on the codes -- computes
// 2 bytes per nonzero that a pattern tile never reads.  Here the order is turned round: the rows are compared first (entry k of a row = entry
// k of the row before + 1: one comparison per nonzero out of LDS), which leaves <= kPatSeg segments of <= kPatLen entries; the columns a tile
// touches are then <= 256 INTERVALS (entry k of a segment's first row, one column further per row), and the same greedy cover runs over those,
// on one wave, four intervals per lane.  The window meta and the record are what the two kernels would have produced (column adjacency implies
// slot adjacency, so a tile accepted here is accepted there; the converse fails only where two windows abut in LDS by coincidence).
// tinfo[b] = kTilePattern, or kTilePlain where the tile has no record: the caller keeps this analysis when at most one tile in a hundred
// is left plain, otherwise it starts over with the window codes.
template <class OffT, int NPT>
__global__ __launch_bounds__(kBlock) void pat_direct_kernel(int64_t nnz, const OffT* __restrict__ row_map, const int32_t* __restrict__ entries,
                                                            const int32_t* __restrict__ blk_info, int32_t* __restrict__ wmeta,
                                                            int32_t* __restrict__ tinfo, int32_t* __restrict__ pmeta) {
  constexpr int TILE = kBlock * NPT, NI = kPatSeg * kPatLen;
  static_assert(NI == kBlock, "one interval per work-item");
  __shared__ int s_col[TILE];
  __shared__ int s_segq[kPatSeg];
  __shared__ int s_lo[NI], s_hi[NI];
  __shared__ int s_base[kWinCount], s_nch[kWinCount], s_off[kWinCount + 1];
  __shared__ int s_nseg, s_bad;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t b = blockIdx.x, s = b * TILE;
  int32_t* out = pmeta + b * kPatW;
  if (s + TILE > nnz) { if (t == 0) tinfo[b] = kTilePlain; return; }        // (uniform) the ragged last tile
  KK_UNROLL
  for (int k = 0; k < NPT; ++k) s_col[k * kBlock + t] = entries[s + (int64_t)k * kBlock + t];
  if (t == 0) { s_nseg = 0; s_bad = 0; }
  const int32_t info0 = blk_info[b], info1 = blk_info[b + 1];
  const int64_t ra = info0 & 0x7fffffff, rb = info1 & 0x7fffffff;
  const int has_head = (info0 >> 31) & 1;
  const int64_t nv = (rb - ra) + has_head;                   // rows with at least their start or their end in this tile
  __syncthreads();
  for (int64_t q = t; q < nv; q += kBlock) {
    const int64_t r  = ra + q - has_head;
    const int64_t rs = (int64_t)row_map[r] - s, re = (int64_t)row_map[r + 1] - s, L = re - rs;
    bool head = true;
    if (L < 1 || L > kPatLen) atomicOr(&s_bad, 1);
    else if (q > 0) {
      const int64_t ps = (int64_t)row_map[r - 1] - s;
      if (rs - ps == L) {
        head = false;
        for (int k = 0; k < (int)L; ++k) {
          const int64_t i0 = ps + k, i1 = rs + k;
          if (i0 >= 0 && i1 < TILE && s_col[i1] != s_col[i0] + 1) head = true;
        }
      }
    }
    if (head) { const int idx = atomicAdd(&s_nseg, 1); if (idx < kPatSeg) s_segq[idx] = (int)q; }
  }
  __syncthreads();
  const int nseg = s_nseg;
  if (s_bad || nseg > kPatSeg || nseg < 1) { if (t == 0) tinfo[b] = kTilePlain; return; }      // (uniform)
  if (t == 0) {                                               // segment heads in row order
    for (int a = 1; a < nseg; ++a) { const int v = s_segq[a]; int c = a - 1; while (c >= 0 && s_segq[c] > v) { s_segq[c + 1] = s_segq[c]; --c; } s_segq[c + 1] = v; }
  }
  __syncthreads();
  // interval t = (segment g, entry k): the columns entry k takes over the segment's rows that have it inside the tile; for the record:
  // the column the entry's slot is derived from (csrc) and the correction (-1 when it is the second row's)
  int csrc = -1, cadj = 0;
  {
    const int g = t / kPatLen, k = t % kPatLen;
    int lo = INT_MAX, hi = -1;
    if (g < nseg) {
      const int q0 = s_segq[g], q1 = (g + 1 < nseg) ? s_segq[g + 1] : (int)nv;
      const int64_t r0 = ra + q0 - has_head;
      const int rs0 = (int)((int64_t)row_map[r0] - s), L = (int)((int64_t)row_map[r0 + 1] - s) - rs0;
      if (k < L) {
        const int i0 = rs0 + k;                               // row j of the segment has the entry at i0 + j L
        const int jlo = i0 < 0 ? 1 : 0;
        int jhi = (TILE - 1 - i0) / L;                        // (i0 < TILE: floor division of a non-negative number)
        if (i0 >= TILE) jhi = -1;
        if (jhi > q1 - q0 - 1) jhi = q1 - q0 - 1;
        if (jlo <= jhi) {
          const int c = s_col[i0 + jlo * L] - jlo;            // the column entry k would have in row 0
          lo = c + jlo; hi = c + jhi;
          csrc = s_col[i0 + jlo * L]; cadj = -jlo;
        }
      }
    }
    s_lo[t] = lo; s_hi[t] = hi;
  }
  __syncthreads();
  // greedy cover, as in win_build_kernel: pass 0 ends a window at its first unused 64-column chunk, pass 1 takes whole windows
  if (wave == 0) {
    int ilo[NI / 64], ihi[NI / 64];
    KK_UNROLL
    for (int u = 0; u < NI / 64; ++u) { ilo[u] = s_lo[lane + 64 * u]; ihi[u] = s_hi[lane + 64 * u]; }
    bool failed = true;
    for (int pass = 0; pass < 2 && failed; ++pass) {
      long long bound = 0;
      int wn = 0;
      for (; wn < kWinCount; ++wn) {
        long long m = LLONG_MAX;
        KK_UNROLL
        for (int u = 0; u < NI / 64; ++u) if ((long long)ihi[u] >= bound) { const long long c = (long long)ilo[u] > bound ? (long long)ilo[u] : bound; m = c < m ? c : m; }
        for (int o = 32; o > 0; o >>= 1) { const long long m2 = __shfl_xor(m, o, 64); m = m2 < m ? m2 : m; }
        if (m == LLONG_MAX) break;                            // (uniform) everything is covered
        const long long base = m;
        unsigned long long used = 0ull;
        KK_UNROLL
        for (int u = 0; u < NI / 64; ++u) {
          const long long cl = (long long)ilo[u] > base ? (long long)ilo[u] : base;
          const long long ch = (long long)ihi[u] < base + (1 << kWinBits) - 1 ? (long long)ihi[u] : base + (1 << kWinBits) - 1;
          if (cl <= ch) {
            const int a = (int)((cl - base) >> 6), z = (int)((ch - base) >> 6);
            used |= ((2ull << z) - 1ull) & ~((1ull << a) - 1ull);
          }
        }
        for (int o = 32; o > 0; o >>= 1) used |= __shfl_xor(used, o, 64);
        int nch;
        if (pass == 0) nch = (~used == 0ull) ? 64 : __ffsll(~used) - 1;
        else           nch = 64 - __clzll((long long)used);
        if (lane == 0) { s_base[wn] = (int)base; s_nch[wn] = nch; }
        bound = base + (pass == 0 ? 64ll * nch : (long long)(1 << kWinBits));
      }
      bool unc = false;
      KK_UNROLL
      for (int u = 0; u < NI / 64; ++u) unc |= ((long long)ihi[u] >= bound);
      failed = __ballot(unc) != 0ull;
      if (!failed && lane == 0) {
        for (int q = wn; q < kWinCount; ++q) { s_base[q] = q ? s_base[q - 1] : 0; s_nch[q] = 0; }
        int off = 0;
        for (int w = 0; w < kWinCount; ++w) { s_off[w] = off; off += 64 * s_nch[w]; }
        s_off[kWinCount] = off;
      }
    }
    if (lane == 0 && failed) s_bad = 1;
  }
  __syncthreads();
  constexpr int CAP = TILE < kWinChunks * 64 ? TILE : kWinChunks * 64;
  // the x window must fit the LDS window and leave the tail of the product array free for the record (float products: 4 B slots)
  if (s_bad || s_off[kWinCount] > CAP || s_off[kWinCount] > TILE - kPatW) { if (t == 0) tinfo[b] = kTilePlain; return; }      // (uniform)
  int32_t* meta = wmeta + b * kWinMeta;
  if (t < kWinCount) { meta[t] = s_base[t]; meta[kWinCount + t] = s_off[t]; }
  if (t < kWinChunks) {
    const int slot = t * 64;
    int col = -1;
    for (int w = 0; w < kWinCount; ++w) if (slot >= s_off[w] && slot < s_off[w] + 64 * s_nch[w]) col = s_base[w] + (slot - s_off[w]);
    meta[2 * kWinCount + t] = col;
  }
  if (t < kPatSeg) {
    int sb = INT_MAX, rs32 = 0, L32 = 1; float M = 1.0f;
    if (t < nseg) {
      const int64_t r  = ra + s_segq[t] - has_head;
      const int64_t rs = (int64_t)row_map[r] - s, L = (int64_t)row_map[r + 1] - s - rs;
      sb = rs > 0 ? (int)rs : 0; rs32 = (int)rs; L32 = (int)L;
      M  = 1.0f / (float)L;
    }
    if (t >= 1) out[t] = sb;
    out[kPatRec + 4 * t + 0] = sb; out[kPatRec + 4 * t + 1] = -rs32; out[kPatRec + 4 * t + 2] = L32; out[kPatRec + 4 * t + 3] = __float_as_int(M);
  }
  if (t == 0) { out[0] = nseg; tinfo[b] = kTilePattern; }
  {
    int val = 0;
    if (csrc >= 0) {
      for (int w = 0; w < kWinCount; ++w) if (s_nch[w] > 0 && csrc >= s_base[w] && csrc < s_base[w] + 64 * s_nch[w]) val = s_off[w] + (csrc - s_base[w]);
      val += cadj;
    }
    reinterpret_cast<short*>(out + kPatTab)[t] = (short)val;     // may be -1: the slot before a row that starts in an earlier tile
  }
}

// Which tiles keep per-nonzero codes (modes 1, 2, and 3 when the records are not used) -> flags for the scan; demotes unused records.
// counts[m] = tiles whose mode (tinfo & 3) is m.  (The build kernels used to count with one atomic per tile: 177,000 atomics on one
// address are 2 ms on C2, the read of the tile table is microseconds.)
__global__ __launch_bounds__(kBlock) void tile_mode_hist_kernel(int64_t nblocks, const int32_t* __restrict__ tinfo, int* __restrict__ counts) {
  __shared__ int s_c[4];
  const int t = threadIdx.x;
  if (t < 4) s_c[t] = 0;
  __syncthreads();
  const int64_t b = (int64_t)blockIdx.x * kBlock + t;
  const int mode = b < nblocks ? (tinfo[b] & 3) : -1;
  for (int m = 0; m < 4; ++m) {
    const int c = __popcll(__ballot(mode == m));
    if ((t & 63) == 0 && c) atomicAdd(&s_c[m], c);
  }
  __syncthreads();
  if (t < 4 && s_c[t]) atomicAdd(counts + t, s_c[t]);
}

__global__ void code_flag_kernel(int64_t nblocks, int32_t* __restrict__ tinfo, int32_t* __restrict__ flag, int use_pat) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nblocks) return;
  if (b == nblocks) { flag[b] = 0; return; }
  int mode = tinfo[b] & 3;
  if (mode == kTilePattern && !use_pat) { mode = kTileStaged; tinfo[b] = mode; }
  flag[b] = (mode == kTileCodes || mode == kTileStaged) ? 1 : 0;
}
// Tile lists per mode: flag, scan, scatter.
__global__ void mode_flag_kernel(int64_t nblocks, const int32_t* __restrict__ tinfo, int32_t* __restrict__ flag, int mode) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nblocks) return;
  flag[b] = (b < nblocks && (tinfo[b] & 3) == mode) ? 1 : 0;
}
__global__ void mode_scatter_kernel(int64_t nblocks, const int32_t* __restrict__ tinfo, const int32_t* __restrict__ idx, int32_t* __restrict__ list, int mode) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nblocks && (tinfo[b] & 3) == mode) list[idx[b]] = (int32_t)b;
}
// Moves the codes of the flagged tiles to their compact position (idx = exclusive scan of the flags) and completes tinfo.
template <int TILE>
__global__ __launch_bounds__(kBlock) void code_compact_kernel(const uint16_t* __restrict__ full, uint16_t* __restrict__ compact,
                                                              int32_t* __restrict__ tinfo, const int32_t* __restrict__ idx) {
  const int64_t b = blockIdx.x;
  const int mode = tinfo[b] & 3;
  if (mode == kTilePattern) { if (threadIdx.x == 0) tinfo[b] = mode | (int32_t)((unsigned)b << 2); return; }   // records stay in place
  if (mode != kTileCodes && mode != kTileStaged) return;
  const int64_t d = idx[b];
  const kk_u32x4* src = reinterpret_cast<const kk_u32x4*>(full + b * TILE);
  kk_u32x4* dst       = reinterpret_cast<kk_u32x4*>(compact + d * TILE);
  for (int i = threadIdx.x; i < TILE / 8; i += kBlock) dst[i] = src[i];
  if (threadIdx.x == 0) tinfo[b] = mode | (int32_t)((unsigned)d << 2);
}

template <class AT, int STEPS, bool FULL, bool NT = false>
__device__ __forceinline__ void load_tile_values(const AT* __restrict__ values, int64_t ts, int64_t te, int t, AT (&v0)[STEPS],
                                                 AT (&v1)[STEPS]) {
  using AV = typename vec2<AT>::type;
  constexpr int SPAN = kBlock * 2;
  KK_UNROLL
  for (int k = 0; k < STEPS; ++k) {
    const int64_t idx = ts + (int64_t)k * SPAN + t * 2;
    if (FULL || idx + 1 < te) {
      const AV* vp = reinterpret_cast<const AV*>(values + idx);
      const AV vv  = NT ? KK_NT_LOAD(vp) : *vp;
      v0[k] = vv[0]; v1[k] = vv[1];
    } else if (idx < te) {
      v0[k] = values[idx]; v1[k] = AT(0);
    } else {
      v0[k] = v1[k] = AT(0);
    }
  }
}

// the 2*STEPS codes of work-item t, two per 32-bit word (nonzero 2k of the item in the low half of word k)
template <int STEPS, bool NT = false>
__device__ __forceinline__ void load_tile_codes(const uint16_t* __restrict__ wtile, int t, unsigned (&w)[STEPS]) {
  // wtile: the tile's codes (the plan stores codes only for the tiles that use them)
  constexpr int NPT = 2 * STEPS;
  const unsigned* cw = reinterpret_cast<const unsigned*>(wtile + t * NPT);   // 4*STEPS bytes, aligned
  if (STEPS % 4 == 0) {
    KK_UNROLL
    for (int k = 0; k < STEPS; k += 4) {
      const kk_u32x4* qp = reinterpret_cast<const kk_u32x4*>(cw + k);
      const kk_u32x4 q   = NT ? KK_NT_LOAD(qp) : *qp;
      w[k] = q[0]; w[k + 1] = q[1]; w[k + 2] = q[2]; w[k + 3] = q[3];
    }
  } else {
    KK_UNROLL
    for (int k = 0; k < STEPS; k += 2) {
      const kk_u32x2* qp = reinterpret_cast<const kk_u32x2*>(cw + k);
      const kk_u32x2 q   = NT ? KK_NT_LOAD(qp) : *qp;
      w[k] = q[0]; w[k + 1] = q[1];
    }
  }
}

template <class AT, int STEPS, bool FULL>
__device__ __forceinline__ void load_tile_win(const AT* __restrict__ values, const uint16_t* __restrict__ wtile,
                                              const int32_t* __restrict__ wmeta, int64_t b, int64_t ts, int64_t te, int t,
                                              AT (&v0)[STEPS], AT (&v1)[STEPS], int (&c0)[STEPS], int (&c1)[STEPS]) {
  constexpr int SPAN = kBlock * 2;
  unsigned w[STEPS];
  load_tile_codes<STEPS>(wtile, t, w);
  const int meta = wmeta[b * kWinMeta + (t & (kWinMeta - 1))];
  load_tile_values<AT, STEPS, FULL>(values, ts, te, t, v0, v1);
  KK_UNROLL
  for (int k = 0; k < STEPS; ++k) {
    const int64_t idx = ts + (int64_t)k * SPAN + t * 2;
    const unsigned lo = w[k] & 0xffffu, hi = w[k] >> 16;
    const int b0 = __shfl(meta, (int)(lo >> kWinBits), 64), b1 = __shfl(meta, (int)(hi >> kWinBits), 64);
    c0[k] = b0 + (int)(lo & ((1u << kWinBits) - 1));
    c1[k] = b1 + (int)(hi & ((1u << kWinBits) - 1));
    if (!FULL) {
      if (idx >= te) c0[k] = c1[k] = -1;
      else if (idx + 1 >= te) c1[k] = c0[k];
    }
  }
}

// Staged-x tile: values, codes, meta and the x chunks are all requested before anything is waited for; the x chunks go to
// LDS (aliasing the product array), every work-item then picks its x entries out of LDS and the products replace them.
template <class AT, class YT, int STEPS, bool FULL, bool NT, bool PAT>
__device__ __forceinline__ void stage_products_win(const AT* __restrict__ values, const uint16_t* __restrict__ wtile,
                                                   const int32_t* __restrict__ wmeta, const YT* __restrict__ x, int64_t ncols,
                                                   YT* prod, int64_t b, int64_t ts, int64_t te, int t,
                                                   const int32_t* __restrict__ pmeta) {
  // PAT (workgroup-uniform template choice made by the caller from the tile's mode): the tile has a row-pattern record
  constexpr int SPAN = kBlock * 2, NPT = 2 * STEPS, TILE = kBlock * NPT;
  constexpr int CAPC = (TILE < kWinChunks * 64 ? TILE : kWinChunks * 64) / 64;     // chunks the LDS window can hold
  constexpr int CPW  = (CAPC + kBlock / 64 - 1) / (kBlock / 64);                     // chunks per wave
  AT v0[STEPS], v1[STEPS];
  unsigned w[STEPS];
  const int lane = t & 63, wave = t >> 6;
  const int meta = wmeta[b * kWinMeta + lane];
  // PAT: a tile with a row-pattern record needs no per-nonzero codes at all
  const int32_t* pm = PAT ? pmeta + b * kPatW : nullptr;
  const int nseg    = PAT ? pm[0] : 0;                         // workgroup-uniform, >= 1
  int prec0 = 0;                                               // kPatW <= kBlock
  if (PAT && t < kPatW) prec0 = pm[t];
  load_tile_values<AT, STEPS, FULL, NT>(values, ts, te, t, v0, v1);
  YT xv[CPW];
  bool used[CPW];                                              // unused chunks are not written: the pattern record may sit there
  KK_UNROLL
  for (int i = 0; i < CPW; ++i) {
    const int c   = wave + i * (kBlock / 64);
    const int col = c < CAPC ? __shfl(meta, 2 * kWinCount + c, 64) : -1;            // wave-uniform
    int64_t xi    = (int64_t)col + lane;
    xi            = xi < ncols ? xi : ncols - 1;
    used[i]       = col >= 0;
    xv[i]         = used[i] ? x[xi] : YT(0);
  }
  if (!PAT) load_tile_codes<STEPS, NT>(wtile, t, w);
  KK_UNROLL
  for (int i = 0; i < CPW; ++i) {
    const int c = wave + i * (kBlock / 64);
    if (used[i]) prod[c * 64 + lane] = xv[i];
  }
  int* sseg = reinterpret_cast<int*>(prod + TILE) - kPatW;     // the record sits behind the x window (the analysis leaves room)
  const short* stab = reinterpret_cast<const short*>(sseg + kPatTab);   // 16-bit (signed) slot tables
  if (PAT && t < kPatW) sseg[t] = prec0;
  __syncthreads();
  YT x0[STEPS], x1[STEPS];
  if (PAT) {
    // nonzero li -> (segment g, row, k): row = floor((li - first row start) / L) by a float reciprocal (exact: (j + 0.5) / L
    // stays 1/64 away from every integer), k by a full-rate 24-bit multiply; slot = T_g[k] + row.  One-segment tiles (no grid
    // line boundary inside) need no search and keep the segment's constants in scalar registers.
    if (nseg == 1) {
      const int jadd = pm[kPatRec + 1];
      const unsigned L = (unsigned)pm[kPatRec + 2];
      const float rcp  = __int_as_float(pm[kPatRec + 3]);
      KK_UNROLL
      for (int k = 0; k < STEPS; ++k) {
        KK_UNROLL
        for (int h = 0; h < 2; ++h) {
          const unsigned j   = (unsigned)(k * SPAN + t * 2 + h + jadd);
          const unsigned row = (unsigned)(((float)j + 0.5f) * rcp);
          const int slot     = (int)stab[(int)(j - KK_UMUL24(row, L))] + (int)row;
          if (h == 0) x0[k] = prod[slot]; else x1[k] = prod[slot];
        }
      }
    } else {
      const int sb1 = pm[1], sb2 = pm[2], sb3 = pm[3], sb4 = pm[4], sb5 = pm[5], sb6 = pm[6], sb7 = pm[7];
      KK_UNROLL
      for (int k = 0; k < STEPS; ++k) {
        KK_UNROLL
        for (int h = 0; h < 2; ++h) {
          const int li = k * SPAN + t * 2 + h;
          int g = (li >= sb1) + (li >= sb2) + (li >= sb3);
          if (nseg > 4) g += (li >= sb4) + (li >= sb5) + (li >= sb6) + (li >= sb7);      // workgroup-uniform
          const int* rec     = sseg + kPatRec + 4 * g;
          const unsigned j   = (unsigned)(li + rec[1]);
          const unsigned L   = (unsigned)rec[2];
          const unsigned row = (unsigned)(((float)j + 0.5f) * __int_as_float(rec[3]));
          const int slot     = (int)stab[g * kPatLen + (int)(j - KK_UMUL24(row, L))] + (int)row;
          if (h == 0) x0[k] = prod[slot]; else x1[k] = prod[slot];
        }
      }
    }
  } else {
    KK_UNROLL
    for (int k = 0; k < STEPS; ++k) {
      const unsigned lo = w[k] & 0xffffu, hi = w[k] >> 16;
      const int s0 = __shfl(meta, kWinCount + (int)(lo >> kWinBits), 64) + (int)(lo & ((1u << kWinBits) - 1));
      const int s1 = __shfl(meta, kWinCount + (int)(hi >> kWinBits), 64) + (int)(hi & ((1u << kWinBits) - 1));
      x0[k] = prod[s0]; x1[k] = prod[s1];
    }
  }
  __syncthreads();                       // every x entry is in registers: the products may overwrite the window
  KK_UNROLL
  for (int k = 0; k < STEPS; ++k) {
    const int li = k * SPAN + t * 2;
    prod[li]     = (YT)v0[k] * x0[k];
    prod[li + 1] = (YT)v1[k] * x1[k];
  }
}

// 32-bit halves of a scalar for lane permutes
__device__ __forceinline__ int  bits_lo(double v) { return (int)(__double_as_longlong(v) & 0xffffffffll); }
__device__ __forceinline__ int  bits_hi(double v) { return (int)(__double_as_longlong(v) >> 32); }
__device__ __forceinline__ double from_bits(int lo, int hi, double) { return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo); }
__device__ __forceinline__ int  bits_lo(float v) { return __float_as_int(v); }
__device__ __forceinline__ int  bits_hi(float) { return 0; }
__device__ __forceinline__ float from_bits(int lo, int, float) { return __int_as_float(lo); }

// value of nnz (8q + off + j) of the quad's eight consecutive nnz, delivered to lane 4q+j, given that lane
// 4q+i holds nnz 8q+2i (a0) and 8q+2i+1 (a1).  ctrl = 0x50 (off 0: lanes 0,0,1,1) or 0xFA (off 4: lanes 2,2,3,3).
template <int CTRL, class T> __device__ __forceinline__ T quad_pick(T a0, T a1, bool odd) {
  // both permutes are executed by every lane (DPP reads the SOURCE lane's operand), then one is selected
  const int l0 = KK_QUAD_PERM(bits_lo(a0), CTRL), l1 = KK_QUAD_PERM(bits_lo(a1), CTRL);
  if (sizeof(T) == 8) {
    const int h0 = KK_QUAD_PERM(bits_hi(a0), CTRL), h1 = KK_QUAD_PERM(bits_hi(a1), CTRL);
    return from_bits(odd ? l1 : l0, odd ? h1 : h0, T());
  }
  return from_bits(odd ? l1 : l0, 0, T());
}
template <int CTRL> __device__ __forceinline__ int quad_pick_i(int a0, int a1, bool odd) {
  const int p0 = KK_QUAD_PERM(a0, CTRL), p1 = KK_QUAD_PERM(a1, CTRL);
  return odd ? p1 : p0;
}

// Stages val*x products of one tile in LDS.  The x gather is what saturates a CU here: the texture
// addresser (TA/TCP) spends one cycle per DISTINCT cache line per 4-lane quad (rocprof: TA_BUSY ~ 100%,
// TCP tag lookups ~ 1/cycle/CU, 3.2 lines per quad with the natural layout).  With QP the (value, column)
// pairs are first re-dealt inside each quad with DPP moves so the four lanes of a quad gather four
// CONSECUTIVE nnz -- columns of consecutive nnz cluster (27-pt: runs of three), ~2.0 lines per quad.
template <class AT, class YT, int STEPS, bool FULL, bool QP>
__device__ __forceinline__ void stage_products(const YT* __restrict__ x, YT* prod, int t, const AT (&v0)[STEPS],
                                               const AT (&v1)[STEPS], const int (&c0)[STEPS], const int (&c1)[STEPS]) {
  constexpr int SPAN = kBlock * 2;
  if (FULL && QP) {
    const bool odd = (t & 1) != 0;
    const int j    = t & 3;
    YT xa[STEPS], xb[STEPS];
    AT va[STEPS], vb[STEPS];
    KK_UNROLL
    for (int k = 0; k < STEPS; ++k) {
      const int ca = quad_pick_i<0x50>(c0[k], c1[k], odd);
      const int cb = quad_pick_i<0xFA>(c0[k], c1[k], odd);
      xa[k] = x[ca]; xb[k] = x[cb];
      va[k] = quad_pick<0x50, AT>(v0[k], v1[k], odd);
      vb[k] = quad_pick<0xFA, AT>(v0[k], v1[k], odd);
    }
    KK_UNROLL
    for (int k = 0; k < STEPS; ++k) {
      const int li = k * SPAN + t * 2 - j;      // = k*SPAN + 8*(t/4) + j
      prod[li]     = (YT)va[k] * xa[k];
      prod[li + 4] = (YT)vb[k] * xb[k];
    }
    return;
  }
  YT x0[STEPS], x1[STEPS];
  KK_UNROLL
  for (int k = 0; k < STEPS; ++k) {
    x0[k] = (FULL || c0[k] >= 0) ? x[FULL ? c0[k] : (c0[k] >= 0 ? c0[k] : 0)] : YT(0);
    x1[k] = (FULL || c1[k] >= 0) ? x[FULL ? c1[k] : (c1[k] >= 0 ? c1[k] : 0)] : YT(0);
  }
  KK_UNROLL
  for (int k = 0; k < STEPS; ++k) {
    const int li = k * SPAN + t * 2;
    prod[li]     = (YT)v0[k] * x0[k];
    prod[li + 1] = (YT)v1[k] * x1[k];
  }
}

// sum of prod[i0+lane], prod[i0+lane+G], ... below i1: four independent partial sums so that four LDS reads are
// in flight per lane (a plain loop waits out one LDS round trip per element: ~1.3 us per 4096-nnz tile)
template <class YT> __device__ __forceinline__ YT strided_lds_sum(const YT* prod, int i0, int i1, int lane, int G) {
  YT s0 = YT(0), s1 = YT(0), s2 = YT(0), s3 = YT(0);
  int i = i0 + lane;
  const int G2 = 2 * G, G3 = 3 * G, G4 = 4 * G;
  for (; i + G3 < i1; i += G4) {
    const YT a = prod[i], b = prod[i + G], c = prod[i + G2], d = prod[i + G3];
    s0 += a; s1 += b; s2 += c; s3 += d;
  }
  for (; i < i1; i += G) s0 += prod[i];
  return (s0 + s1) + (s2 + s3);
}

// The planned kernel (nnz-split tiles).  What bounds a kernel that needs ~100 KB in flight per CU is the DEPENDENCY CHAIN
// each tile goes through: the tile descriptor (first row + "starts inside a row" flag: one scalar load) is requested first,
// the streaming loads do not depend on it, and the per-lane row bounds row_map[r], row_map[r+1] are requested together
// with the x gathers -- so a tile sees two memory latencies (stream, then gather + bounds).  After the barrier the row
// reduction touches only LDS and registers.
// The column analysis is PER TILE (tinfo[b]): kTilePlain reads entries and gathers x with quad-dealt loads, kTileCodes takes
// its columns from the plan's 16-bit window codes, kTileStaged also stages the tile's x ranges in LDS (stage_products_win),
// kTilePattern decodes a row-pattern record and reads no per-nonzero code at all.  One tile the codes cannot cover costs
// that tile its codes, not the matrix.  Each mode is its own instantiation (MODE) launched over the plan's LIST of the
// tiles of that mode (null = every tile): one kernel with all four paths needs 135-152 registers per work-item where the
// pattern path alone needs 88, i.e. three instead of five workgroups per CU for the tiles that matter.
template <class OffT, class AT, class YT, int NPT, int MODE>
__global__ __launch_bounds__(kBlock) void spmv_stream3_kernel(int64_t nnz, const OffT* __restrict__ row_map,
                                                              const int32_t* __restrict__ entries,
                                                              const AT* __restrict__ values, const YT* __restrict__ x,
                                                              YT* __restrict__ y, YT alpha, YT beta,
                                                              const int32_t* __restrict__ blk_info,
                                                              YT* __restrict__ carry_head, YT* __restrict__ carry_tail,
                                                              int remap, const int32_t* __restrict__ list,
                                                              const int32_t* __restrict__ tinfo,
                                                              const uint16_t* __restrict__ wcode,
                                                              const int32_t* __restrict__ wmeta, int64_t ncols,
                                                              const int32_t* __restrict__ pmeta KK_ABL_PARAM) {
  // KK_ABL bits (measurement build): 4 = no y stores, 8 = no LDS reduction loop, 16 = synthetic row bounds (no row_map
  // loads), 32 = no barrier, 64 / 128 = y-store experiments (see the store)
  constexpr int TILE  = kBlock * NPT;
  constexpr int STEPS = NPT / 2;
  __shared__ YT prod[TILE];
  constexpr bool NT = false;
  const int t       = threadIdx.x;
  const int64_t pos = xcd_order(blockIdx.x, gridDim.x, remap);
  const int64_t b   = list ? (int64_t)list[pos] : pos;         // the plan's list of the tiles of this mode (ascending), or every tile
  const int64_t s = b * TILE;
  const bool full = (s + TILE <= nnz);
  const int64_t e = full ? s + TILE : nnz;

  // tile descriptor: bit 31 = the tile starts inside a row (a "head" segment exists)
  const int32_t info0 = blk_info[b], info1 = blk_info[b + 1];
  const uint16_t* wtile = nullptr;
  if (MODE == kTileCodes || MODE == kTileStaged) wtile = wcode + (int64_t)((unsigned)tinfo[b] >> 2) * TILE;   // the tile's codes

  AT v0[STEPS], v1[STEPS];
  int c0[STEPS], c1[STEPS];
  if (MODE >= kTileStaged) {
    // nothing to load here: stage_products_win requests values, codes and x chunks together
  } else if (MODE == kTileCodes) {
    if (full) load_tile_win<AT, STEPS, true>(values, wtile, wmeta, b, s, e, t, v0, v1, c0, c1);
    else      load_tile_win<AT, STEPS, false>(values, wtile, wmeta, b, s, e, t, v0, v1, c0, c1);
  } else {
    if (full) load_tile<AT, STEPS, NT, true>(values, entries, s, e, t, v0, v1, c0, c1);
    else      load_tile<AT, STEPS, NT, false>(values, entries, s, e, t, v0, v1, c0, c1);
  }

  const int64_t ra    = info0 & 0x7fffffff;
  const int64_t rb    = info1 & 0x7fffffff;
  const int has_head  = (info0 >> 31) & 1;
  const int64_t nv    = (rb - ra) + has_head;          // virtual rows: [head] + rows starting in this tile
  int G = 1;
  while (G < kWave && nv * (G * 2) <= kBlock) G *= 2;
  const int lane = t & (G - 1), grp = t / G, ngrp = kBlock / G;
  // virtual row j -> matrix row r = ra + j - has_head (r = ra-1 is the head's row); bounds for the first pass
  bool valid = grp < nv;
  int64_t r  = ra + grp - has_head;
  int64_t rs = 0, re = 0;
  if (valid) {
    if (KK_ABL(16)) { rs = s + (int64_t)grp * 27; re = rs + 27; }
    else { rs = (int64_t)row_map[r]; re = (int64_t)row_map[r + 1]; }
  }
  // beta != 0: the old y of the first pass' rows is requested now, with everything else, instead of right before the store
  YT yold = YT(0);
  if (beta != YT(0) && valid && lane == 0 && r >= 0) yold = y[r];

  if (MODE == kTilePattern) {                                   // records exist for full tiles only (the ragged last tile keeps its codes)
    stage_products_win<AT, YT, STEPS, true, NT, true>(values, wtile, wmeta, x, ncols, prod, b, s, e, t, pmeta);
  } else if (MODE == kTileStaged) {
    if (full) stage_products_win<AT, YT, STEPS, true, NT, false>(values, wtile, wmeta, x, ncols, prod, b, s, e, t, nullptr);
    else      stage_products_win<AT, YT, STEPS, false, NT, false>(values, wtile, wmeta, x, ncols, prod, b, s, e, t, nullptr);
  } else {
    if (full) stage_products<AT, YT, STEPS, true, true>(x, prod, t, v0, v1, c0, c1);
    else      stage_products<AT, YT, STEPS, false, false>(x, prod, t, v0, v1, c0, c1);
  }
  if (!KK_ABL(32)) __syncthreads();

  for (int64_t base = 0; base < nv; base += ngrp) {
    if (base > 0) {
      valid = (base + grp) < nv;
      r     = ra + base + grp - has_head;
      if (valid) { rs = (int64_t)row_map[r]; re = (int64_t)row_map[r + 1]; }
    }
    const bool is_head  = r < ra;
    const bool complete = !is_head && re <= e;
    const int i0 = (int)((rs > s ? rs : s) - s), i1 = (int)((re < e ? re : e) - s);
    YT sum = valid ? (KK_ABL(8) ? prod[i0 < TILE ? i0 : 0] : strided_lds_sum<YT>(prod, i0, i1, lane, G)) : YT(0);
    sum = group_sum(sum, G);
    if (valid && lane == 0) {
      if (is_head) carry_head[b] = sum;
      else if (!complete) carry_tail[b] = sum;
      else if (!KK_ABL(4) && !(KK_ABL(64) && (b & 7))) {          // 64: only one tile in eight stores its rows
        sum *= alpha;
        const YT out = (beta == YT(0)) ? sum : beta * (base == 0 ? yold : y[r]) + sum;
        y[KK_ABL(128) ? (r & 255) : r] = out;                      // 128: every store lands in the same 2 KB
      }
    }
  }
}

// finishes the rows cut by tile boundaries: thread b owns the row that starts in tile b and ends later.
template <class OffT, class YT>
__global__ void spmv_stream_fixup_kernel(int64_t nblocks, int64_t nnz, int64_t tile, const OffT* __restrict__ row_map,
                                         const int32_t* __restrict__ blk_row, const YT* __restrict__ carry_head,
                                         const YT* __restrict__ carry_tail, YT* __restrict__ y, YT alpha, YT beta) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  const int64_t ra = blk_row[b] & 0x7fffffff, rb = blk_row[b + 1] & 0x7fffffff;
  if (rb == ra) return;
  const int64_t e  = ((b + 1) * tile < nnz) ? (b + 1) * tile : nnz;
  const int64_t R  = rb - 1;
  const int64_t re = (int64_t)row_map[R + 1];
  if (re <= e) return;
  YT total = carry_tail[b];
  for (int64_t b2 = b + 1; b2 < nblocks && b2 * tile < re; ++b2) total += carry_head[b2];
  total *= alpha;
  y[R] = (beta == YT(0)) ? total : beta * y[R] + total;
}

// ------------------------------------------------------------------------------------------------
// y += alpha * A^T x after y := beta*y; LPR lanes per row, hardware float atomics.
template <class OffT, class AT, class YT, int LPR>
__global__ __launch_bounds__(kBlock) void spmv_transpose_kernel(int64_t nrows, const OffT* __restrict__ row_map,
                                                                const int32_t* __restrict__ entries,
                                                                const AT* __restrict__ values,
                                                                const YT* __restrict__ x, YT* __restrict__ y, YT alpha) {
  constexpr int RPB = kBlock / LPR;
  const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;
  const int lane    = threadIdx.x % LPR;
  if (row >= nrows) return;
  const YT xv  = alpha * x[row];
  const OffT s = row_map[row], e = row_map[row + 1];
  for (OffT j = s + lane; j < e; j += LPR) atomicAdd(&y[entries[j]], (YT)values[j] * xv);
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch
static int pick_lpr(int64_t nrows, int64_t nnz, int forced) {
  if (forced > 0) { int l = 1; while (l < forced && l < 64) l *= 2; return l; }
  const int64_t avg = nrows > 0 ? nnz / nrows : 1;
  int l = 1;
  while (l < 64 && l * 2 <= avg) l *= 2;   // largest power of two <= nnz/row: one trip for most rows
  return l;
}

template <class OffT, class AT, class YT, int LPR>
static int launch_vector(const kkamd_crs_t* A, const YT* x, YT* y, YT alpha, YT beta, int remap, hipStream_t st) {
  const int64_t nwg = ceil_div(A->num_rows, kBlock / LPR);
  KK_LAUNCH((spmv_vector_kernel<OffT, AT, YT, LPR>), (unsigned)nwg, kBlock, 0, st, A->num_rows,
            (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, x, y, alpha, beta, remap);
  KK_LAUNCH_CHECK();
  return KKAMD_OK;
}

template <class OffT, class AT, class YT>
static int run_vector(const kkamd_crs_t* A, const YT* x, YT* y, YT alpha, YT beta, const SpmvTuning& tn, hipStream_t st) {
  const int remap = tn.xcd_remap == 1;                      // the grouped orders (>= 2) belong to the nnz-split kernel
  switch (pick_lpr(A->num_rows, A->nnz, tn.lanes_per_row)) {
    case 1:  return launch_vector<OffT, AT, YT, 1>(A, x, y, alpha, beta, remap, st);
    case 2:  return launch_vector<OffT, AT, YT, 2>(A, x, y, alpha, beta, remap, st);
    case 4:  return launch_vector<OffT, AT, YT, 4>(A, x, y, alpha, beta, remap, st);
    case 8:  return launch_vector<OffT, AT, YT, 8>(A, x, y, alpha, beta, remap, st);
    case 16: return launch_vector<OffT, AT, YT, 16>(A, x, y, alpha, beta, remap, st);
    case 32: return launch_vector<OffT, AT, YT, 32>(A, x, y, alpha, beta, remap, st);
    default: return launch_vector<OffT, AT, YT, 64>(A, x, y, alpha, beta, remap, st);
  }
}

template <class OffT, class AT, class YT, int NPT>
static int launch_stream(const kkamd_spmv_plan* p, const kkamd_crs_t* A, const YT* x, YT* y, YT alpha, YT beta,
                         hipStream_t st) {
  YT* ch = reinterpret_cast<YT*>(p->d_carry);
  YT* ct = reinterpret_cast<YT*>(reinterpret_cast<char*>(p->d_carry) + 8 * p->nblocks);
  // one launch per tile mode the plan holds, each over the list of its tiles (no column analysis: every tile is plain)
#define KK_STREAM3(MODE, NTILES, LIST)                                                                                    \
  KK_LAUNCH((spmv_stream3_kernel<OffT, AT, YT, NPT, MODE>), (unsigned)(NTILES), kBlock, KK_LDS_PAD(p), st, A->nnz,        \
            (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, x, y, alpha, beta,          \
            (const int32_t*)p->d_blk_row, ch, ct, p->tune.xcd_remap, (const int32_t*)(LIST), (const int32_t*)p->d_tinfo, \
            (const uint16_t*)p->d_wcode, (const int32_t*)p->d_wbase, A->num_cols, (const int32_t*)p->d_pmeta KK_ABL_ARG(p))
  if (!p->d_tinfo) {
    KK_STREAM3(kTilePlain, p->nblocks, nullptr);
  } else {
    if (NPT != 4 && p->n_mode[kTilePattern] > 0) KK_STREAM3((NPT == 4 ? kTileStaged : kTilePattern), p->n_mode[kTilePattern], p->d_list[kTilePattern]);
    if (p->n_mode[kTileStaged] > 0) KK_STREAM3(kTileStaged, p->n_mode[kTileStaged], p->d_list[kTileStaged]);
    if (p->n_mode[kTileCodes] > 0)  KK_STREAM3(kTileCodes, p->n_mode[kTileCodes], p->d_list[kTileCodes]);
    if (p->n_mode[kTilePlain] > 0)  KK_STREAM3(kTilePlain, p->n_mode[kTilePlain], p->d_list[kTilePlain]);
  }
#undef KK_STREAM3
  KK_LAUNCH_CHECK();
  KK_LAUNCH((spmv_stream_fixup_kernel<OffT, YT>), (unsigned)ceil_div(p->nblocks, kBlock), kBlock, 0, st, p->nblocks,
            A->nnz, (int64_t)p->tile, (const OffT*)A->d_row_map, (const int32_t*)p->d_blk_row, (const YT*)ch,
            (const YT*)ct, y, alpha, beta);
  KK_LAUNCH_CHECK();
  return KKAMD_OK;
}

// tile sizes the kernels are instantiated for: 2048 and 4096 nnz, plus 1024 for fp64 values
template <class OffT, class AT, class YT>
static int run_stream(const kkamd_spmv_plan* p, const kkamd_crs_t* A, const YT* x, YT* y, YT alpha, YT beta, hipStream_t st) {
  const int npt = p->tile / kBlock;
  if (npt * kBlock != p->tile) return fail(KKAMD_ERR_STATE, "kkamd_spmv: plan tile %d is not a multiple of the workgroup size", p->tile);
  if (npt == 16) return launch_stream<OffT, AT, YT, 16>(p, A, x, y, alpha, beta, st);
  if (npt == 8)  return launch_stream<OffT, AT, YT, 8>(p, A, x, y, alpha, beta, st);
  if (npt == 4 && sizeof(AT) == 8) return launch_stream<OffT, AT, YT, (sizeof(AT) == 8 ? 4 : 8)>(p, A, x, y, alpha, beta, st);
  return fail(KKAMD_ERR_STATE, "kkamd_spmv: no kernel for %d nonzeros per work-item with %d-byte values", npt, (int)sizeof(AT));
}

template <class OffT, class AT, class YT>
static int run_transpose(const kkamd_crs_t* A, const YT* x, YT* y, YT alpha, YT beta, hipStream_t st) {
  int rc = launch_scale<YT>(y, A->num_cols, 1, 1, 1, beta, st);
  if (rc) return rc;
  const int64_t avg = A->nnz / A->num_rows;
  if (avg >= 32) {
    KK_LAUNCH((spmv_transpose_kernel<OffT, AT, YT, 32>), (unsigned)ceil_div(A->num_rows, kBlock / 32), kBlock, 0, st,
              A->num_rows, (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, x, y, alpha);
  } else if (avg >= 6) {
    KK_LAUNCH((spmv_transpose_kernel<OffT, AT, YT, 8>), (unsigned)ceil_div(A->num_rows, kBlock / 8), kBlock, 0, st,
              A->num_rows, (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, x, y, alpha);
  } else {
    KK_LAUNCH((spmv_transpose_kernel<OffT, AT, YT, 1>), (unsigned)ceil_div(A->num_rows, kBlock), kBlock, 0, st,
              A->num_rows, (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, x, y, alpha);
  }
  KK_LAUNCH_CHECK();
  return KKAMD_OK;
}

static bool stream_usable(const kkamd_spmv_plan* p, const kkamd_crs_t* A, int elem_size) {
  if (!p || p->tile == 0 || p->tune.kernel == 1) return false;
  // aligned 2-element vector loads need 2*sizeof(value) / 8-byte alignment of the array bases
  if (((uintptr_t)A->d_values % (uintptr_t)(2 * elem_size)) != 0 || ((uintptr_t)A->d_entries % 8) != 0) return false;
  return true;
}

// ------------------------------------------------------------------------------------------------
// Modes T / H.  The reference scatters with atomics (spmv_impl.hpp:383-513, "functional, not performant"): 11.2 ms on 27-pt
// 300^3.  An analysed handle instead caches the TRANSPOSE -- structure, values in transposed order, and for every
// entry of A its position in A^T -- and runs the planned N kernel on A^T: deterministic, no atomics.  The matrix values may
// change between calls (the structure may not): every call re-fingerprints A's values (128 bits per 4096 values, one coalesced
// stream of 8 bytes per nonzero -- kk_spmv_colslab.hip says what the fingerprints are) and moves the tiles that changed into
// their transposed places;
//   explicit_transpose = 1 (default from explicit_transpose_min_knnz when the plan memory fits an eighth of free HBM): that;
//   explicit_transpose = 2: the caller promises constant values, the comparison is skipped too;
//   explicit_transpose = 0: the reference's atomic scatter.
// Costs nnz * (4 + sizeof(offset) + sizeof(value)) bytes of plan memory and 0.3 s once for the transpose; falls back to the
// atomic kernel if the memory cannot be had.
__global__ void iota_f64_kernel(double* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
// inv[perm[j]] = j: where entry i of A sits in A^T
template <class OffT> __global__ void invert_perm_kernel(const double* __restrict__ perm_f64, OffT* __restrict__ inv, int64_t n) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) inv[(int64_t)perm_f64[j]] = (OffT)j;
}
// the cached transpose is a convenience, not a right: it is only built when it (and the 16 bytes per nonzero its construction
// needs on top) fit an eighth of the HBM that is free at that moment
template <class OffT, class AT>
static bool transpose_fits(kkamd_spmv_plan* p, const kkamd_crs_t* A) {
  if (p->t_ready) return true;
  if (p->t_failed) return false;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return false; }
  // structure, permutation, values, 16 bytes per nonzero while it is built -- and, under exact value tracking (the default), the shadow copy
  // of A.values the comparison needs
  const bool shadow = p->tune.values_tracking == 0 && p->tune.explicit_transpose != 2;
  const double need = (double)A->nnz * (4.0 + sizeof(OffT) + sizeof(AT) + 16.0 + (shadow ? (double)sizeof(AT) : 0.0)) + (double)A->num_cols * sizeof(OffT);
  if (need > (double)free_b / 8.0) { p->t_failed = true; return false; }
  return true;
}
// knobs that belong to the handle's own re-ordered copies and are not passed on to the plan of its cached transpose
static bool knob_stays_with_parent(const std::string& k) {
  return k == "colslab" || k == "colslab_shift" || k == "colslab_rate_pct" || k == "colslab_min_knnz" || k == "colslab_const" || k == "explicit_transpose" ||
         k == "explicit_transpose_min_knnz" || k == "values_tracking";
}
template <class OffT, class AT>
static int ensure_transpose(kkamd_spmv_plan* p, const kkamd_crs_t* A, hipStream_t st) {
  if (p->t_ready) return KKAMD_OK;
  if (p->t_failed) return KKAMD_ERR_ALLOC;
  const size_t nnz = (size_t)A->nnz;
  double *d_iota = nullptr, *d_tmp = nullptr;
  auto fail_clean = [&]() {
    (void)hipGetLastError();
    if (d_iota) (void)hipFree(d_iota);
    if (d_tmp) (void)hipFree(d_tmp);
    if (p->d_t_rm) (void)hipFree(p->d_t_rm);
    if (p->d_t_ent) (void)hipFree(p->d_t_ent);
    if (p->d_t_perm) (void)hipFree(p->d_t_perm);
    if (p->d_t_fp) (void)hipFree(p->d_t_fp);
    if (p->d_t_val) (void)hipFree(p->d_t_val);
    p->d_t_rm = nullptr; p->d_t_ent = nullptr; p->d_t_perm = nullptr; p->d_t_val = nullptr; p->d_t_fp = nullptr;
    p->t_failed = true;
    return KKAMD_ERR_ALLOC;
  };
  if (hipMalloc(&p->d_t_rm, sizeof(OffT) * (size_t)(A->num_cols + 1)) != hipSuccess || hipMalloc((void**)&p->d_t_ent, sizeof(int32_t) * nnz) != hipSuccess ||
      hipMalloc(&p->d_t_perm, sizeof(OffT) * nnz) != hipSuccess || hipMalloc(&p->d_t_val, sizeof(AT) * nnz) != hipSuccess ||
      hipMalloc((void**)&p->d_t_fp, 16 * (size_t)values_fp_tiles(A->nnz)) != hipSuccess ||
      hipMalloc((void**)&d_iota, sizeof(double) * nnz) != hipSuccess || hipMalloc((void**)&d_tmp, sizeof(double) * nnz) != hipSuccess)
    return fail_clean();
  const unsigned grid = (unsigned)(ceil_div(A->nnz, kBlock) < 65536 ? ceil_div(A->nnz, kBlock) : 65536);
  KK_LAUNCH(iota_f64_kernel, grid, kBlock, 0, st, d_iota, A->nnz);
  // positions are exact in fp64 (nnz < 2^53), so the transpose of the "values" 0..nnz-1 IS the permutation
  int rc = kkamd_transpose(A->num_rows, A->num_cols, A->nnz, A->d_row_map, (const int32_t*)A->d_entries, d_iota, A->offset_type, KKAMD_F64,
                           p->d_t_rm, p->d_t_ent, d_tmp, reinterpret_cast<kkamd_stream_t>(st));
  if (rc != KKAMD_OK) { fail_clean(); return rc; }
  KK_LAUNCH((invert_perm_kernel<OffT>), grid, kBlock, 0, st, (const double*)d_tmp, (OffT*)p->d_t_perm, A->nnz);     // d_t_perm: position of A's entry i in A^T
  p->t_fp_valid = false; p->t_shadow_valid = false; p->t_stale = true;          // the first refresh moves every value
  // exact tracking: the shadow copy is part of the transpose (budgeted in transpose_fits), not an allocation on the first tracked call
  if (p->tune.values_tracking == 0 && p->tune.explicit_transpose != 2 && !p->d_t_shadow && !p->t_shadow_failed &&
      hipMalloc(&p->d_t_shadow, sizeof(AT) * nnz) != hipSuccess) { (void)hipGetLastError(); p->d_t_shadow = nullptr; p->t_shadow_failed = true; }
  if (hipStreamSynchronize(st) != hipSuccess) return fail_clean();
  (void)hipFree(d_iota); (void)hipFree(d_tmp); d_iota = d_tmp = nullptr;
  kkamd_crs_t At{A->num_cols, A->num_rows, A->nnz, p->d_t_rm, p->d_t_ent, p->d_t_val, A->offset_type, A->value_type};
  // what the caller set on the handle holds for its transposed modes too (a forced rank-2 kernel, mv5 / mv6 off, check_entries, ...):
  // the transpose's plan is created with the handle's knobs, except those of the handle's own re-ordered copies
  std::vector<const char*> t_keys; std::vector<int> t_vals;
  for (const auto& kv : p->set_log) if (!knob_stays_with_parent(kv.first)) { t_keys.push_back(kv.first.c_str()); t_vals.push_back(kv.second); }
  rc = kkamd_spmv_plan_create_knobs(&p->t_plan, &At, p->algorithm, t_keys.data(), t_vals.data(), (int)t_keys.size(), reinterpret_cast<kkamd_stream_t>(st));
  if (rc != KKAMD_OK) { fail_clean(); return rc; }
  p->t_plan->tune.colslab = 0;            // A^T's values are this plan's own copy: no second re-ordered copy to keep current behind it
  p->t_plan->tune.explicit_transpose = 0;
  p->t_ready = true;
  return KKAMD_OK;
}

// the cached transpose and its plan are given up (a knob the parent took could not be applied to the transpose's plan): the next T / H call
// builds them again from the parent's knobs
static void drop_transpose(kkamd_spmv_plan* p) {
  if (p->used) (void)hipStreamSynchronize(p->last_stream);
  if (p->t_plan) { (void)kkamd_spmv_plan_destroy(p->t_plan); p->t_plan = nullptr; }
  void** bufs[] = {&p->d_t_rm, (void**)&p->d_t_ent, &p->d_t_perm, (void**)&p->d_t_fp, &p->d_t_shadow, &p->d_t_val};
  for (void** b : bufs) if (*b) { (void)hipFree(*b); *b = nullptr; }
  p->t_ready = false; p->t_fp_valid = false; p->t_shadow_valid = false; p->t_stale = true;
}

template <class OffT> static int analyse(kkamd_spmv_plan* p, const kkamd_crs_t* A, hipStream_t st);

// per-host-thread scratch plan for calls without an analysed handle; grown on demand, never shrunk.  Work is stream
// ordered: consecutive calls on one stream may share the scratch; a change of stream (or device) drains the old one.
struct TransientPlan {
  kkamd_spmv_plan plan;
  size_t cap_blk = 0, cap_carry = 0;
  int device = -1;
  hipStream_t last_stream = nullptr;
  bool used = false;
};
static TransientPlan& transient_ref() { static thread_local TransientPlan tp; return tp; }
int release_transient() {
  TransientPlan& tp = transient_ref();
  if (tp.used) (void)hipStreamSynchronize(tp.last_stream);
  if (tp.plan.d_blk_row) (void)hipFree(tp.plan.d_blk_row);
  if (tp.plan.d_carry) (void)hipFree(tp.plan.d_carry);
  tp.plan.d_blk_row = nullptr; tp.plan.d_carry = nullptr; tp.cap_blk = tp.cap_carry = 0; tp.used = false;
  return KKAMD_OK;
}
template <class OffT>
static kkamd_spmv_plan* transient_plan(const kkamd_crs_t* A, const SpmvTuning& tn, int elem_size, hipStream_t st) {
  TransientPlan& tp = transient_ref();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  kkamd_spmv_plan& p = tp.plan;
  if (tp.used && (tp.device != dev || tp.last_stream != st)) {
    (void)hipStreamSynchronize(tp.last_stream);
    if (tp.device != dev) { p.d_blk_row = nullptr; p.d_carry = nullptr; tp.cap_blk = tp.cap_carry = 0; }   // other device's buffers are abandoned
  }
  p.num_rows = A->num_rows; p.num_cols = A->num_cols; p.nnz = A->nnz; p.row_map = A->d_row_map; p.entries = A->d_entries;
  p.offset_type = A->offset_type; p.value_type = A->value_type; p.algorithm = KKAMD_SPMV_FAST_SETUP; p.tune = tn;
  p.tune.window_codes = 0;        // one-shot plans do not pay for the column codes
  int npt = (elem_size == 8 && A->nnz >= 200000000) ? 16 : 8;
  if (elem_size == 8 && (tn.nnz_per_thread == 4 || tn.nnz_per_thread == 8 || tn.nnz_per_thread == 16)) npt = tn.nnz_per_thread;
  p.tile    = kBlock * npt;
  p.nblocks = ceil_div(A->nnz, p.tile);
  const size_t need_blk = sizeof(int32_t) * (size_t)(p.nblocks + 1), need_carry = (size_t)16 * (size_t)p.nblocks;
  if (need_blk > tp.cap_blk || need_carry > tp.cap_carry) {
    if (tp.used) (void)hipStreamSynchronize(st);
    if (p.d_blk_row) (void)hipFree(p.d_blk_row);
    if (p.d_carry) (void)hipFree(p.d_carry);
    p.d_blk_row = nullptr; p.d_carry = nullptr; tp.cap_blk = tp.cap_carry = 0;
    const size_t cb = need_blk + need_blk / 4, cc = need_carry + need_carry / 4;
    if (hipMalloc((void**)&p.d_blk_row, cb) != hipSuccess || hipMalloc(&p.d_carry, cc) != hipSuccess) {
      if (p.d_blk_row) (void)hipFree(p.d_blk_row);
      p.d_blk_row = nullptr; p.d_carry = nullptr; p.tile = 0;
      (void)hipGetLastError();
      return nullptr;                        // out of memory: the no-analysis kernel still works
    }
    tp.cap_blk = cb; tp.cap_carry = cc;
  }
  tp.device = dev; tp.last_stream = st; tp.used = true;
  {
    hipDeviceProp_t prop;
    static thread_local int cus = 0;
    if (!cus) cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    p.num_cus = cus;
  }
  if (analyse<OffT>(&p, A, st) != KKAMD_OK) return nullptr;
  return &p;
}

// The cached transpose of an analysed handle with its values brought up to date under the "values_tracking" policy (exact by default),
// or *tplan = nullptr when the handle has none (not analysed, knob off, matrix too small, no memory).
template <class OffT, class AT>
static int transpose_view_t(kkamd_spmv_plan* plan, const kkamd_crs_t* A, hipStream_t st, kkamd_crs_t* At, kkamd_spmv_plan** tplan) {
  *tplan = nullptr;
  if (!(plan && plan->tile != 0 && plan->tune.explicit_transpose && A->nnz >= (int64_t)plan->tune.explicit_transpose_min_knnz * 1000 &&
        transpose_fits<OffT, AT>(plan, A) && ensure_transpose<OffT, AT>(plan, A, st) == KKAMD_OK))
    return KKAMD_OK;
  const int rc = values_track(plan->tune.values_tracking, plan->tune.explicit_transpose == 2, A->offset_type, A->value_type, A->nnz, A->d_values, plan->d_t_perm,
                              plan->d_t_val, plan->d_t_fp, &plan->d_t_shadow, &plan->t_fp_valid, &plan->t_shadow_valid, &plan->t_shadow_failed, &plan->t_stale, st);
  if (rc) return rc;
  *At = kkamd_crs_t{A->num_cols, A->num_rows, A->nnz, plan->d_t_rm, plan->d_t_ent, plan->d_t_val, A->offset_type, A->value_type};
  *tplan = plan->t_plan;
  return KKAMD_OK;
}
int transpose_view(kkamd_spmv_plan* plan, const kkamd_crs_t* A, hipStream_t st, kkamd_crs_t* At, kkamd_spmv_plan** tplan) {
  const bool o64 = A->offset_type == KKAMD_I64;
  if (A->value_type == KKAMD_F64) return o64 ? transpose_view_t<int64_t, double>(plan, A, st, At, tplan) : transpose_view_t<int32_t, double>(plan, A, st, At, tplan);
  return o64 ? transpose_view_t<int64_t, float>(plan, A, st, At, tplan) : transpose_view_t<int32_t, float>(plan, A, st, At, tplan);
}

// Column-slab copy (kk_spmv_colslab.hip), decided once per plan at the first mode-N call: the analysis must say "gather-bound"
// (most tiles read plain entries, x is several L2s large), then both kernels are timed on the caller's x into a scratch y and the
// copy is kept when it wins by 10 % including its fingerprint pass.  colslab = 2 skips the gates and the timing (tests).
template <class OffT, class AT, class YT>
static int colslab_select(kkamd_spmv_plan* plan, const kkamd_crs_t* A, const YT* x, hipStream_t st) {
  plan->cs_tried = true;
  const int mode = plan->tune.colslab;
  const bool force = mode == 2 || mode == 4, det = mode >= 3;
  if (!force) {
    const int64_t plain = plan->d_tinfo ? plan->plain_tiles : plan->nblocks;
    if (A->nnz < (int64_t)plan->tune.colslab_min_knnz * 1000 || (double)A->num_cols * sizeof(YT) < 16.0 * 1048576.0 || 2 * plain < plan->nblocks) return KKAMD_OK;
    if (det) {
      // the rule (no timing): sampled windows of the matrix name nearly one 128-byte line of x per nonzero
      int rc0 = cs_gather_ratio(A, (int)sizeof(YT), st, &plan->cs_ratio);
      if (rc0) return rc0;
      if (g_verbose) printf("kkamd_spmv: %.3f distinct lines of x per nonzero in the sampled windows: %s\n", plan->cs_ratio, plan->cs_ratio * 100.0 >= (double)plan->tune.colslab_min_pct ? "column-slab form (deterministic)" : "CRS kernel");
      if (plan->cs_ratio * 100.0 < (double)plan->tune.colslab_min_pct) return KKAMD_OK;
      // ... and the slab form must pay by bytes: it streams 16 B per nonzero (+ 16 B under exact value tracking) and writes and reads a
      // partial sum per (slab, row) -- whole lines of a [slabs][rows] array, half empty when the rows are short (weight 1.5) -- against one
      // 128-byte line of x per missing nonzero through the fabric.  Rates fitted on one MI355X (uniform random 5e6 x 20: 1.32 ms slab form
      // against 1.75 ms CRS; 2e7 x 8: 3.85 against 3.27 -- the rule must say no there): 4.24 TB/s and 6.95 TB/s, 10 % margin.
      const int sh = cs_pick_shift(A, (int)sizeof(YT), plan->tune.colslab_shift, true);        // the slabs cs_build will make
      const double nslabs = (double)ceil_div(A->num_cols, (int64_t)1 << sh);
      const bool exact = plan->tune.values_tracking == 0 && !plan->tune.colslab_const;
      const double cost_slab = (double)A->nnz * (16.0 + (exact ? 16.0 : 0.0)) + 1.5 * nslabs * (double)A->num_rows * 2.0 * sizeof(YT);
      const double cost_crs = (double)A->nnz * plan->cs_ratio * 128.0;
      // the two kernels' rates on their own bytes, fitted on one MI355X (4.24 and 6.95 TB/s: the slab form's streams against the lines the CRS
      // kernel's gathers pull); what decides is their RATIO, a knob in percent (default 61) for parts or clocks where it lies elsewhere
      if (cost_slab * 1.1 >= cost_crs * (0.01 * (double)plan->tune.colslab_rate_pct)) {
        if (g_verbose) printf("kkamd_spmv: the column-slab form would move %.2f GB against %.2f GB of x lines: CRS kernel\n", cost_slab / 1e9, cost_crs / 1e9);
        return KKAMD_OK;
      }
    }
  }
  int rc = cs_build(&plan->cs, A, (int)sizeof(YT), plan->tune.colslab_shift, st, det);
  if (rc || !plan->cs || force || det) return rc;
#ifdef KK_EMU
  cs_plan_destroy(plan->cs); plan->cs = nullptr;                 // nothing to time under the emulator
  return KKAMD_OK;
#else
  DevBuf ys;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  auto drop = [&]() { (void)hipGetLastError(); cs_plan_destroy(plan->cs); plan->cs = nullptr; for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); return KKAMD_OK; };
  if (ys.alloc(sizeof(YT) * (size_t)A->num_rows) != hipSuccess) return drop();
  for (hipEvent_t& e : ev) if (hipEventCreate(&e) != hipSuccess) return drop();
  YT* yscr = ys.as<YT>();
  const int check = plan->tune.colslab_const ? -1 : plan->tune.values_tracking;
  constexpr int kReps = 5;
  for (int phase = 0; phase < 2; ++phase) {                     // phase 0 warms both up
    if (phase == 1 && hipEventRecord(ev[0], st) != hipSuccess) return drop();
    for (int i = 0; i < (phase ? kReps : 2); ++i) if ((rc = run_stream<OffT, AT, YT>(plan, A, x, yscr, YT(1), YT(0), st))) { drop(); return rc; }
    if (phase == 1 && hipEventRecord(ev[1], st) != hipSuccess) return drop();
    for (int i = 0; i < (phase ? kReps : 2); ++i) if ((rc = cs_apply(plan->cs, A, scalar_tag<YT>::value, x, yscr, 1.0, 0.0, check, st))) { drop(); return rc; }
    if (phase == 1 && hipEventRecord(ev[2], st) != hipSuccess) return drop();
  }
  float ms_crs = 0.f, ms_cs = 0.f;
  if (hipEventSynchronize(ev[2]) != hipSuccess || hipEventElapsedTime(&ms_crs, ev[0], ev[1]) != hipSuccess || hipEventElapsedTime(&ms_cs, ev[1], ev[2]) != hipSuccess) return drop();
  plan->cs_crs_us = 1e3 * ms_crs / kReps; plan->cs_us = 1e3 * ms_cs / kReps;
  if (g_verbose) printf("kkamd_spmv: column-slab copy %.1f us against %.1f us for the CRS kernel: %s\n", plan->cs_us, plan->cs_crs_us, plan->cs_us < 0.9 * plan->cs_crs_us ? "kept" : "dropped");
  if (!(plan->cs_us < 0.9 * plan->cs_crs_us)) { drop(); return KKAMD_OK; }
  for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  return KKAMD_OK;
#endif
}

static int build_analysis(kkamd_spmv_plan* p, const kkamd_crs_t* A, hipStream_t st, int force_npt);
template <class OffT, class AT, class YT>
static int spmv_typed(kkamd_spmv_plan* plan, const kkamd_crs_t* A, bool trans, double alpha_d, const void* dx,
                      double beta_d, void* dy, hipStream_t st) {
  const YT alpha = (YT)alpha_d, beta = (YT)beta_d;
  const YT* x    = (const YT*)dx;
  YT* y          = (YT*)dy;
  if (trans) {
    kkamd_crs_t At{}; kkamd_spmv_plan* tplan = nullptr;
    const int rc = transpose_view_t<OffT, AT>(plan, A, st, &At, &tplan);
    if (rc) return rc;
    if (tplan) return spmv_typed<OffT, AT, YT>(tplan, &At, false, alpha_d, dx, beta_d, dy, st);
    return run_transpose<OffT, AT, YT>(A, x, y, alpha, beta, st);
  }
  if (plan && plan->rank1_deferred) {                          // knob defer_rank1: the first rank-1 call of a handle that began with rank 2
    plan->rank1_deferred = false;
    plan->tune.defer_rank1 = 0;
    const int rc = build_analysis(plan, A, st, 0);
    if (rc) return rc;
  }
  if constexpr (sizeof(YT) == 8) {
    // lattice stencils: the plane-marching kernel reads no column index and fetches x once per patch plane (knob march)
    if (plan && plan->tile != 0 && plan->tune.march && plan->entries == A->d_entries) {
      int ran = 0;
      const int rc = march_spmv(plan, A, (const double*)x, (double*)y, (double)alpha, (double)beta, st, &ran);
      if (rc || ran) return rc;
    }
  }
  if (stream_usable(plan, A, (int)sizeof(AT))) {
    if (plan->tune.colslab && plan->entries == A->d_entries) {
      if (!plan->cs_tried) { const int rc = colslab_select<OffT, AT, YT>(plan, A, x, st); if (rc) return rc; }
      if (plan->cs) return cs_apply(plan->cs, A, scalar_tag<YT>::value, x, y, (double)alpha, (double)beta, plan->tune.colslab_const ? -1 : plan->tune.values_tracking, st);
    }
    return run_stream<OffT, AT, YT>(plan, A, x, y, alpha, beta, st);
  }
  // No analysed plan (handle-less overloads, SPMV_FAST_SETUP): a large matrix is still worth the nnz-split kernel --
  // its "analysis" is one tiny kernel (a binary search per 4096-nnz tile) into a per-thread scratch that is reused
  // from call to call, so nothing is allocated or kept per matrix and the call stays asynchronous.
  const SpmvTuning& tn = plan ? plan->tune : g_spmv_default;
  if (tn.kernel != 1 && tn.transient_min_knnz > 0 && A->nnz >= (int64_t)tn.transient_min_knnz * 1000 && A->num_rows > 0) {
    kkamd_spmv_plan* tp = transient_plan<OffT>(A, tn, (int)sizeof(AT), st);
    if (tp && stream_usable(tp, A, (int)sizeof(AT))) return run_stream<OffT, AT, YT>(tp, A, x, y, alpha, beta, st);
  }
  return run_vector<OffT, AT, YT>(A, x, y, alpha, beta, tn, st);
}

int check_crs(const kkamd_crs_t* A) {
  if (!A) return fail(KKAMD_ERR_INVALID_ARG, "kkamd: null matrix descriptor");
  if (A->num_rows < 0 || A->num_cols < 0 || A->nnz < 0)
    return fail(KKAMD_ERR_INVALID_ARG, "kkamd: negative matrix dimension");
  if (A->num_rows > INT32_MAX || A->num_cols > INT32_MAX)
    return fail(KKAMD_ERR_INVALID_ARG, "kkamd: dimensions exceed the int32 ordinal range");
  if (A->offset_type != KKAMD_I32 && A->offset_type != KKAMD_I64)
    return fail(KKAMD_ERR_INVALID_ARG, "kkamd: unknown offset_type %d", A->offset_type);
  if (A->value_type != KKAMD_F32 && A->value_type != KKAMD_F64)
    return fail(KKAMD_ERR_UNSUPPORTED, "kkamd: unsupported value_type %d", A->value_type);
  if (A->offset_type == KKAMD_I32 && A->nnz > INT32_MAX)
    return fail(KKAMD_ERR_INVALID_ARG, "kkamd: nnz does not fit 32-bit offsets");
  if (A->num_rows > 0 && !A->d_row_map) return fail(KKAMD_ERR_INVALID_ARG, "kkamd: null row_map");
  if (A->nnz > 0 && (!A->d_entries || !A->d_values)) return fail(KKAMD_ERR_INVALID_ARG, "kkamd: null entries/values");
  return KKAMD_OK;
}

int check_plan(const kkamd_spmv_plan* p, const kkamd_crs_t* A) {
  if (!p) return KKAMD_OK;
  if (p->num_rows != A->num_rows || p->num_cols != A->num_cols || p->nnz != A->nnz || p->row_map != A->d_row_map ||
      p->offset_type != A->offset_type || p->value_type != A->value_type || ((p->d_tinfo || p->mv || p->mv4 || p->mv5 || p->mv6 || p->t_ready || p->cs) && p->entries != A->d_entries))
    return fail(KKAMD_ERR_STATE, "kkamd_spmv: plan was created for a different matrix (a handle is bound to one matrix)");
  return KKAMD_OK;
}

// knob check_entries (debug aid): an order-independent 64-bit hash of (position, column) over the matrix's column array, taken when the
// handle first sees the matrix and again at every call.  The analysis (tile modes, pattern records, lattice plans, cached transpose) is
// only valid for the structure it was made from; check_plan can compare pointers and sizes, this compares content.
__global__ __launch_bounds__(kBlock) void entries_hash_kernel(int64_t nnz, const int32_t* __restrict__ ent, unsigned long long* __restrict__ out) {
  unsigned long long h = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * kBlock) {
    unsigned long long z = ((unsigned long long)(unsigned)ent[i] << 32 | (unsigned long long)(i & 0xffffffffll)) + 0x9E3779B97F4A7C15ull * (unsigned long long)(i >> 32);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    h += z ^ (z >> 31);
  }
  h = group_sum(h, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}
int check_entries_content(kkamd_spmv_plan* p, const kkamd_crs_t* A, hipStream_t st) {
  if (!p || !p->tune.check_entries || A->nnz == 0 || p->entries != A->d_entries) return KKAMD_OK;
  DevBuf buf;
  KK_HIP(buf.alloc(sizeof(unsigned long long)));
  KK_HIP(hipMemsetAsync(buf.p, 0, sizeof(unsigned long long), st));
  const int64_t nb = ceil_div(A->nnz, kBlock);
  unsigned long long* d_h = buf.as<unsigned long long>();
  KK_LAUNCH(entries_hash_kernel, (unsigned)(nb < 4096 ? nb : 4096), kBlock, 0, st, A->nnz, (const int32_t*)A->d_entries, d_h);
  KK_LAUNCH_CHECK();
  unsigned long long h = 0;
  KK_HIP(hipMemcpyAsync(&h, buf.p, sizeof h, hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  if (!p->entries_hash_known) { p->entries_hash = h; p->entries_hash_known = true; return KKAMD_OK; }
  if (h != p->entries_hash)
    return fail(KKAMD_ERR_STATE, "kkamd_spmv: the matrix's column indices changed in place since the handle analysed them (check_entries)");
  return KKAMD_OK;
}

// TPL_SpMV_Data::set_exec_space (sparse/src/KokkosSparse_spmv_handle.hpp:95-104): a handle's scratch (carry slots, packed
// X / Y) is ordered by the stream it was last used on; when the caller switches streams, the old one is fenced first.
int bind_stream(kkamd_spmv_plan* p, hipStream_t st) {
  if (!p) return KKAMD_OK;
  if (p->used && p->last_stream != st) KK_HIP(hipStreamSynchronize(p->last_stream));
  p->last_stream = st; p->used = true;
  return KKAMD_OK;
}

int parse_mode(char mode, bool* trans) {
  switch (mode) {
    case 'N': case 'n': case 'C': case 'c': *trans = false; return KKAMD_OK;
    case 'T': case 't': case 'H': case 'h': *trans = true; return KKAMD_OK;
    default: return fail(KKAMD_ERR_INVALID_ARG, "Invalid transpose mode %c for KokkosSparse::spmv()", mode);
  }
}

static int set_tuning(SpmvTuning& t, const char* key, int value) {
  if (!key) return fail(KKAMD_ERR_INVALID_ARG, "null key");
  const std::string k(key);
  auto bad = [&](const char* what) { return fail(KKAMD_ERR_INVALID_ARG, "tuning key '%s': %d is not %s", key, value, what); };
  if (k == "kernel") { if (value < 0 || value > 2) return bad("0, 1 or 2"); t.kernel = value; }
  else if (k == "lanes_per_row") { if (value < 0 || value > 64) return bad("in 0..64"); t.lanes_per_row = value; }
  else if (k == "nnz_per_thread") { if (value != 0 && value != 4 && value != 8 && value != 16) return bad("0, 4, 8 or 16"); t.nnz_per_thread = value; }
  else if (k == "xcd_remap") { if (!valid_order_knob(value)) return bad("0, 1 or a power of two"); t.xcd_remap = value; }
  else if (k == "mv_kernel") { if (value < 0 || value > 6) return bad("in 0..6"); t.mv_kernel = value; }
  else if (k == "mv6") { if (value < 0 || value > 2) return bad("in 0..2"); t.mv6 = value; }
  else if (k == "mv6_min_long_pct") { if (value < 0 || value > 100) return bad("a percentage"); t.mv6_min_long_pct = value; }
  else if (k == "mv5") { if (value < 0 || value > 2) return bad("in 0..2"); t.mv5 = value; }
  else if (k == "mv5_min_fill_pct") { if (value < 0 || value > 100) return bad("a percentage"); t.mv5_min_fill_pct = value; }
  else if (k == "mv5_max_other_pct") { if (value < 0 || value > 100) return bad("a percentage"); t.mv5_max_other_pct = value; }
  else if (k == "stream_variant") { if (value != 1 && value != 6) return bad("1 or 6"); t.stream_variant = value; }
  else if (k == "mv_remap") { if (!valid_order_knob(value)) return bad("0, 1 or a power of two"); t.mv_remap = value; }
  else if (k == "mv_order") { if (value < 0 || value > 2) return bad("in 0..2"); t.mv_order = value; }
  else if (k == "mv_strip_min_kb") { if (value < 0) return bad("non-negative"); t.mv_strip_min_kb = value; }
  else if (k == "mv_strip_l2_kb") { if (value < 1) return bad("positive"); t.mv_strip_l2_kb = value; }
  else if (k == "mv_nt") { if (value != 0 && value != 1) return bad("0 or 1"); t.mv_nt = value; }
  else if (k == "mv_glds") { if (value != 0 && value != 1) return bad("0 or 1"); t.mv_glds = value; }
  else if (k == "mv_long_T") { if (value < 0) return bad("non-negative"); t.mv_long_T = value; }
  else if (k == "mv4_min_nvec") { if (value < 1 || value > 1024) return bad("in 1..1024"); t.mv4_min_nvec = value; }
  else if (k == "mv4_2d") { if (value != 0 && value != 1) return bad("0 or 1"); t.mv4_2d = value; }
  else if (k == "mv4_xcol") { if (value != 0 && value != 1) return bad("0 or 1"); t.mv4_xcol = value; }
  else if (k == "mv4_wg_per_cu") { if (value < 1 || value > 64) return bad("in 1..64"); t.mv4_wg_per_cu = value; }
  else if (k == "march") { if (value != 0 && value != 1) return bad("0 or 1"); t.march = value; }
  else if (k == "march_planes") { if (value < 1 || value > 4096) return bad("in 1..4096"); t.march_planes = value; }
  else if (k == "window_codes") { if (value < 0 || value > 2) return bad("in 0..2"); t.window_codes = value; }
  else if (k == "window_codes_min_knnz") { if (value < 0) return bad("non-negative"); t.window_codes_min_knnz = value; }
  else if (k == "window_codes_min_pct") { if (value < 0 || value > 100) return bad("a percentage"); t.window_codes_min_pct = value; }
  else if (k == "pattern_codes") { if (value < 0 || value > 2) return bad("in 0..2"); t.pattern_codes = value; }
  else if (k == "pattern_direct") { if (value != 0 && value != 1) return bad("0 or 1"); t.pattern_direct = value; }
  else if (k == "pattern_codes_min_knnz") { if (value < 0) return bad("non-negative"); t.pattern_codes_min_knnz = value; }
  else if (k == "colslab") { if (value < 0 || value > 4) return bad("in 0..4"); t.colslab = value; }
  else if (k == "colslab_min_pct") { if (value < 0 || value > 100) return bad("in 0..100"); t.colslab_min_pct = value; }
  else if (k == "colslab_min_knnz") { if (value < 0) return bad("non-negative"); t.colslab_min_knnz = value; }
  else if (k == "colslab_shift") { if (value != 0 && (value < 2 || value > 30)) return bad("0 or in 2..30"); t.colslab_shift = value; }
  else if (k == "colslab_rate_pct") { if (value < 1 || value > 1000) return bad("in 1..1000"); t.colslab_rate_pct = value; }
  else if (k == "values_tracking") { if (value < 0 || value > 2) return bad("in 0..2"); t.values_tracking = value; }
  else if (k == "check_entries") { if (value != 0 && value != 1) return bad("0 or 1"); t.check_entries = value; }
  else if (k == "defer_rank1") { if (value != 0 && value != 1) return bad("0 or 1"); t.defer_rank1 = value; }
  else if (k == "colslab_const") { if (value != 0 && value != 1) return bad("0 or 1"); t.colslab_const = value; }
  else if (k == "transient_min_knnz") { if (value < 0) return bad("non-negative"); t.transient_min_knnz = value; }
  else if (k == "explicit_transpose") { if (value < 0 || value > 2) return bad("in 0..2"); t.explicit_transpose = value; }
  else if (k == "explicit_transpose_min_knnz") { if (value < 0) return bad("non-negative"); t.explicit_transpose_min_knnz = value; }
#ifdef KK_ABLATE
  else if (k == "ablate") t.ablate = value;
  else if (k == "lds_pad_kb") t.lds_pad_kb = value;
#endif
  else return fail(KKAMD_ERR_INVALID_ARG, "unknown tuning key '%s'", key);
  return KKAMD_OK;
}

template <class OffT>
static int analyse(kkamd_spmv_plan* p, const kkamd_crs_t* A, hipStream_t st) {
  KK_LAUNCH((spmv_plan_kernel<OffT>), (unsigned)ceil_div(p->nblocks + 1, kBlock), kBlock, 0, st, A->num_rows,
            (const OffT*)A->d_row_map, p->nblocks, (int64_t)p->tile, p->d_blk_row);
  KK_LAUNCH_CHECK();
  return KKAMD_OK;
}

static void free_analysis(kkamd_spmv_plan* p) {
  void** bufs[] = {(void**)&p->d_blk_row, &p->d_carry, (void**)&p->d_tinfo, (void**)&p->d_wcode, (void**)&p->d_wbase, (void**)&p->d_pmeta};
  for (void** b : bufs) if (*b) { (void)hipFree(*b); *b = nullptr; }
  for (int m = 0; m < 4; ++m) { if (p->d_list[m]) { (void)hipFree(p->d_list[m]); p->d_list[m] = nullptr; } p->n_mode[m] = 0; }
  p->pat_tiles = p->code_tiles = p->staged_tiles = p->plain_tiles = 0; p->pat_direct = false;
  p->tile = 0; p->nblocks = 0; p->plan_bytes = 0;
}

// Column analysis of the tiling (window codes, staged x, row-pattern records), tile by tile.  Returns KKAMD_OK with
// p->d_tinfo == nullptr when the codes are not used (not worth it, or no memory for them); *redo_npt != 0 asks the caller to
// repeat the whole analysis with that many nonzeros per work-item (the tile size was picked for the codes).
static int build_codes(kkamd_spmv_plan* p, const kkamd_crs_t* A, hipStream_t st, int npt, bool may_retile, int* redo_npt) {
  *redo_npt = 0;
  const size_t nb = (size_t)p->nblocks;
  DevBuf full, counts, flag, tinfo_b, wbase_b;                  // freed on every early return
  auto give_up = [&]() {
    (void)hipGetLastError();
    if (p->d_pmeta) { (void)hipFree(p->d_pmeta); p->d_pmeta = nullptr; }
    if (p->d_wcode) { (void)hipFree(p->d_wcode); p->d_wcode = nullptr; }
    for (int m = 0; m < 4; ++m) { if (p->d_list[m]) { (void)hipFree(p->d_list[m]); p->d_list[m] = nullptr; } p->n_mode[m] = 0; }
    p->d_tinfo = nullptr; p->d_wbase = nullptr;
    p->pat_tiles = p->code_tiles = p->staged_tiles = 0; p->plain_tiles = p->nblocks;
    p->win_failed = true;
    return KKAMD_OK;
  };
  // the codes are an optimisation: if HBM cannot hold them (2 bytes per nonzero while they are built) the plan keeps reading entries
  if (full.alloc(sizeof(uint16_t) * nb * (size_t)p->tile) != hipSuccess || counts.alloc(12 * sizeof(int)) != hipSuccess ||
      flag.alloc(sizeof(int32_t) * (nb + 1)) != hipSuccess || tinfo_b.alloc(sizeof(int32_t) * nb) != hipSuccess ||
      wbase_b.alloc(sizeof(int32_t) * nb * kWinMeta) != hipSuccess)
    return give_up();
  int32_t* tinfo = tinfo_b.as<int32_t>();
  int32_t* wbase = wbase_b.as<int32_t>();
  uint16_t* d_full = full.as<uint16_t>();                      // raw pointers for the launches (the buffers stay owned above)
  int* d_counts    = counts.as<int>();
  int32_t* d_flag  = flag.as<int32_t>();
  KK_HIP(hipMemsetAsync(counts.p, 0, 12 * sizeof(int), st));
  const int allow_stage = p->tune.window_codes != 2;
  const int32_t* ent = (const int32_t*)p->entries;
  // the launch lists: the tiles of every mode, ascending (none needed when one mode has every tile).  known[m] >= 0: the count of mode m is
  // known already (no scan, no copy back for the modes without tiles).  Returns -1 when a launch or an allocation fails.
  auto make_lists = [&](const int64_t* known, size_t* list_bytes) -> int {
    for (int m = 0; m < 4; ++m) {
      if (known && known[m] == 0) { p->n_mode[m] = 0; continue; }
      if (known && known[m] == (int64_t)nb) { p->n_mode[m] = (int64_t)nb; continue; }
      KK_LAUNCH(mode_flag_kernel, (unsigned)ceil_div((int64_t)nb + 1, kBlock), kBlock, 0, st, (int64_t)nb, (const int32_t*)tinfo, d_flag, m);
      if (hipGetLastError() != hipSuccess) return -1;
      int rc2 = exclusive_scan_inplace<int32_t>(d_flag, (int64_t)nb + 1, st);
      if (rc2) { give_up(); return rc2; }
      int32_t cnt = 0;
      if (known) cnt = (int32_t)known[m];
      else {
        KK_HIP(hipMemcpyAsync(&cnt, d_flag + nb, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        KK_HIP(hipStreamSynchronize(st));
      }
      p->n_mode[m] = cnt;
      if (cnt > 0 && (size_t)cnt < nb) {
        if (hipMalloc((void**)&p->d_list[m], sizeof(int32_t) * (size_t)cnt) != hipSuccess) return -1;
        *list_bytes += sizeof(int32_t) * (size_t)cnt;
        int32_t* d_list_m = p->d_list[m];
        KK_LAUNCH(mode_scatter_kernel, (unsigned)ceil_div((int64_t)nb, kBlock), kBlock, 0, st, (int64_t)nb, (const int32_t*)tinfo, (const int32_t*)d_flag, d_list_m, m);
        if (hipGetLastError() != hipSuccess) return -1;
      }
    }
    return 0;
  };
  const bool pat_wanted = p->tune.pattern_codes && (npt == 16 || npt == 8) &&
                          (p->tune.pattern_codes >= 2 || A->nnz >= (int64_t)p->tune.pattern_codes_min_knnz * 1000);
  // Row-pattern records straight from the matrix (pat_direct_kernel): kept when at most one tile in a hundred (one, at least) is left
  // without a record -- those read entries like the tiles of an unanalysed matrix --; otherwise the window codes are built as before.
  if (pat_wanted && allow_stage && p->tune.pattern_direct && hipMalloc((void**)&p->d_pmeta, sizeof(int32_t) * nb * kPatW) == hipSuccess) {
    const bool o64 = A->offset_type == KKAMD_I64;
    int* d_cnt = d_counts + 8;
#define KK_PAT_DIRECT(OT, N)                                                                                                 \
  KK_LAUNCH((pat_direct_kernel<OT, N>), (unsigned)nb, kBlock, 0, st, A->nnz, (const OT*)A->d_row_map, ent,                   \
            (const int32_t*)p->d_blk_row, wbase, tinfo, p->d_pmeta)
    if (npt == 16) { if (o64) { KK_PAT_DIRECT(int64_t, 16); } else { KK_PAT_DIRECT(int32_t, 16); } }
    else           { if (o64) { KK_PAT_DIRECT(int64_t, 8); } else { KK_PAT_DIRECT(int32_t, 8); } }
#undef KK_PAT_DIRECT
    if (hipGetLastError() != hipSuccess) return give_up();
    KK_LAUNCH(tile_mode_hist_kernel, (unsigned)ceil_div((int64_t)nb, kBlock), kBlock, 0, st, (int64_t)nb, (const int32_t*)tinfo, d_cnt);
    if (hipGetLastError() != hipSuccess) return give_up();
    int hd[4] = {0, 0, 0, 0};
    KK_HIP(hipMemcpyAsync(hd, d_cnt, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
    const int64_t left = (int64_t)nb - hd[kTilePattern], allowed = (int64_t)nb / 100 > 1 ? (int64_t)nb / 100 : 1;
    if (hd[kTilePattern] > 0 && left <= allowed) {
      if (hipMalloc((void**)&p->d_wcode, sizeof(uint16_t) * (size_t)p->tile) != hipSuccess) return give_up();     // (no tile reads codes)
      const int64_t known[4] = {left, 0, 0, (int64_t)hd[kTilePattern]};
      size_t list_bytes = 0;
      int rc2 = make_lists(known, &list_bytes);
      if (rc2) return rc2 == -1 ? give_up() : rc2;
      KK_HIP(hipStreamSynchronize(st));
      p->d_tinfo = (int32_t*)tinfo_b.release(); p->d_wbase = (int32_t*)wbase_b.release();
      p->plain_tiles = left; p->pat_tiles = hd[kTilePattern]; p->code_tiles = 0; p->staged_tiles = hd[kTilePattern]; p->pat_direct = true;
      p->plan_bytes += list_bytes + sizeof(int32_t) * nb * (1 + kWinMeta) + sizeof(uint16_t) * (size_t)p->tile + sizeof(int32_t) * nb * kPatW;
      return KKAMD_OK;
    }
    KK_HIP(hipFree(p->d_pmeta)); p->d_pmeta = nullptr;
  } else {
    (void)hipGetLastError();
    if (p->d_pmeta) { (void)hipFree(p->d_pmeta); p->d_pmeta = nullptr; }
  }
  if (npt == 16)     { KK_LAUNCH((win_build_kernel<16>), (unsigned)nb, kBlock, 0, st, A->nnz, ent, d_full, wbase, tinfo, allow_stage); }
  else if (npt == 8) { KK_LAUNCH((win_build_kernel<8>), (unsigned)nb, kBlock, 0, st, A->nnz, ent, d_full, wbase, tinfo, allow_stage); }
  else               { KK_LAUNCH((win_build_kernel<4>), (unsigned)nb, kBlock, 0, st, A->nnz, ent, d_full, wbase, tinfo, allow_stage); }
  if (hipGetLastError() != hipSuccess) return give_up();
  KK_LAUNCH(tile_mode_hist_kernel, (unsigned)ceil_div((int64_t)nb, kBlock), kBlock, 0, st, (int64_t)nb, (const int32_t*)tinfo, d_counts);
  if (hipGetLastError() != hipSuccess) return give_up();
  int h[8] = {0};
  KK_HIP(hipMemcpyAsync(h, counts.p, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  const int64_t n_plain = h[kTilePlain], n_codes = h[kTileCodes], n_staged = h[kTileStaged];
  // worth it?  Tiles that cannot use the codes keep reading entries, so this is a question of how many can.
  if ((double)(n_codes + n_staged) * 100.0 < (double)p->tune.window_codes_min_pct * (double)nb || n_codes + n_staged == 0) return give_up();
  // 4096-nnz tiles were picked in the hope that their x windows stage: when fewer than 90 % do, 2048-nnz tiles serve better
  if (may_retile && npt == 16 && allow_stage && (double)n_staged < 0.9 * (double)nb) { give_up(); p->win_failed = false; *redo_npt = 8; return KKAMD_OK; }
  // row-pattern records for the staged tiles
  int64_t n_pat = 0;
  bool use_pat = false;
  if (n_staged > 0 && pat_wanted &&
      hipMalloc((void**)&p->d_pmeta, sizeof(int32_t) * nb * kPatW) == hipSuccess) {
    const bool o64 = A->offset_type == KKAMD_I64;
    int* d_cnt = d_counts + 4;
#define KK_PAT_BUILD(OT, N)                                                                                                  \
  KK_LAUNCH((pat_build_kernel<OT, N>), (unsigned)nb, kBlock, 0, st, A->nnz, (const OT*)A->d_row_map,                        \
            (const int32_t*)p->d_blk_row, d_full, (const int32_t*)wbase, tinfo, p->d_pmeta)
    if (npt == 16) { if (o64) { KK_PAT_BUILD(int64_t, 16); } else { KK_PAT_BUILD(int32_t, 16); } }
    else           { if (o64) { KK_PAT_BUILD(int64_t, 8); } else { KK_PAT_BUILD(int32_t, 8); } }
#undef KK_PAT_BUILD
    if (hipGetLastError() != hipSuccess) return give_up();
    KK_LAUNCH(tile_mode_hist_kernel, (unsigned)ceil_div((int64_t)nb, kBlock), kBlock, 0, st, (int64_t)nb, (const int32_t*)tinfo, d_cnt);
    if (hipGetLastError() != hipSuccess) return give_up();
    int h_cnt = 0;
    KK_HIP(hipMemcpyAsync(&h_cnt, d_cnt + kTilePattern, sizeof(int), hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
    n_pat   = h_cnt;
    use_pat = p->tune.pattern_codes >= 2 ? n_pat > 0 : (double)n_pat >= 0.9 * (double)nb;
    if (!use_pat) { KK_HIP(hipFree(p->d_pmeta)); p->d_pmeta = nullptr; n_pat = 0; }
  } else {
    (void)hipGetLastError();
  }
  // keep codes only for the tiles that read them: flags -> scan -> compact copy
  KK_LAUNCH(code_flag_kernel, (unsigned)ceil_div((int64_t)nb + 1, kBlock), kBlock, 0, st, (int64_t)nb, tinfo, d_flag, use_pat ? 1 : 0);
  if (hipGetLastError() != hipSuccess) return give_up();
  int rc = exclusive_scan_inplace<int32_t>(d_flag, (int64_t)nb + 1, st);
  if (rc) { give_up(); return rc; }
  int32_t h_keep = 0;
  KK_HIP(hipMemcpyAsync(&h_keep, d_flag + nb, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  if (hipMalloc((void**)&p->d_wcode, sizeof(uint16_t) * (size_t)(h_keep > 0 ? h_keep : 1) * (size_t)p->tile) != hipSuccess) return give_up();
  if (npt == 16)     { KK_LAUNCH((code_compact_kernel<kBlock * 16>), (unsigned)nb, kBlock, 0, st, d_full, p->d_wcode, tinfo, d_flag); }
  else if (npt == 8) { KK_LAUNCH((code_compact_kernel<kBlock * 8>), (unsigned)nb, kBlock, 0, st, d_full, p->d_wcode, tinfo, d_flag); }
  else               { KK_LAUNCH((code_compact_kernel<kBlock * 4>), (unsigned)nb, kBlock, 0, st, d_full, p->d_wcode, tinfo, d_flag); }
  if (hipGetLastError() != hipSuccess) return give_up();
  // the launch lists: the tiles of every mode, ascending (none needed when one mode has every tile)
  size_t list_bytes = 0;
  if ((rc = make_lists(nullptr, &list_bytes))) return rc == -1 ? give_up() : rc;
  KK_HIP(hipStreamSynchronize(st));
  p->d_tinfo = (int32_t*)tinfo_b.release(); p->d_wbase = (int32_t*)wbase_b.release();
  p->plan_bytes += list_bytes;
  p->plain_tiles = n_plain; p->pat_tiles = n_pat; p->code_tiles = h_keep; p->staged_tiles = n_staged;
  p->plan_bytes += sizeof(int32_t) * nb * (1 + kWinMeta) + sizeof(uint16_t) * (size_t)h_keep * (size_t)p->tile +
                   (p->d_pmeta ? sizeof(int32_t) * nb * kPatW : 0);
  return KKAMD_OK;
}

static int build_analysis(kkamd_spmv_plan* p, const kkamd_crs_t* A, hipStream_t st, int force_npt) {
  if (p->d_blk_row) KK_HIP(hipStreamSynchronize(st));
  free_analysis(p);
  p->rank1_deferred = false;
  if (p->algorithm == KKAMD_SPMV_FAST_SETUP || p->tune.kernel == 1 || A->nnz == 0 || A->num_rows == 0) return KKAMD_OK;
  int npt = p->tune.nnz_per_thread;
  const bool f64v = A->value_type == KKAMD_F64;
  const bool want_win = !p->win_failed && p->entries && p->tune.window_codes &&
                        (p->tune.stream_variant == 6 || A->nnz >= (int64_t)p->tune.window_codes_min_knnz * 1000);
  const bool auto_npt = (npt != 4 && npt != 8 && npt != 16);
  // auto: 4096-nnz tiles once there are plenty of them (measured best from ~2e8 nnz up), 2048-nnz tiles below that
  // (5-pt 1000^2: 19.2 vs 22.2 us).  With window codes: 4096-nnz tiles when the matrix is large and their x windows fit
  // LDS (27-pt 300^3: 1.38 ms against 1.44 with 2048-nnz tiles), else 2048-nnz tiles (7-pt 400^3 is only stageable at
  // 2048) -- build_codes asks for the 4096 attempt to be redone at 2048 when too few tiles stage.
  if (force_npt) npt = force_npt;
  else if (auto_npt) {
    if (!f64v) npt = 8;                                        // fp32 values: 2048-nnz tiles unless asked
    else if (want_win) npt = (A->nnz >= 50000000) ? 16 : 8;
    else npt = (A->nnz < 200000000) ? 8 : 16;
  }
  if (npt == 4 && !f64v) npt = 8;                              // 1024-nnz tiles are instantiated for fp64 values only
  p->tile    = kBlock * npt;
  p->nblocks = ceil_div(A->nnz, p->tile);
  if (p->tune.defer_rank1 && !force_npt) {                     // a rank-2 caller: the rank-1 plan is built by the first rank-1 call, if one comes
    p->rank1_deferred = true;
    p->nblocks = 0;
    return KKAMD_OK;
  }
  if (hipMalloc((void**)&p->d_blk_row, sizeof(int32_t) * (size_t)(p->nblocks + 1)) != hipSuccess ||
      hipMalloc(&p->d_carry, (size_t)16 * (size_t)p->nblocks) != hipSuccess) {
    free_analysis(p);
    return fail(KKAMD_ERR_ALLOC, "kkamd_spmv_plan: out of device memory for the tile descriptors");
  }
  p->plan_bytes = (sizeof(int32_t) + 16) * (size_t)p->nblocks + 4;
  int rc = A->offset_type == KKAMD_I64 ? analyse<int64_t>(p, A, st) : analyse<int32_t>(p, A, st);
  if (rc) return rc;
  if (want_win && ((uintptr_t)p->entries % 8 == 0)) {
    int redo = 0;
    if ((rc = build_codes(p, A, st, npt, auto_npt && !force_npt, &redo))) return rc;
    if (redo) return build_analysis(p, A, st, redo);
    if (!p->d_tinfo && auto_npt && !force_npt && f64v && npt != ((A->nnz < 200000000) ? 8 : 16))
      return build_analysis(p, A, st, 0);                      // no codes after all (win_failed is set): the plain kernel's tile size
  }
  KK_HIP(hipStreamSynchronize(st));   // setup is synchronous, like the vendor analysis it replaces
  return KKAMD_OK;
}

SpmvTuning g_spmv_default;
int g_verbose = 0;

#ifdef KK_EMU
void trace_push(const char*) {}
void trace_pop() {}
#else
}  // namespace kk
#include <dlfcn.h>
namespace kk {
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)()             = nullptr;
  Roctx() {
    void* h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    pop  = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
Roctx& roctx() { static Roctx r; return r; }
}  // namespace
void trace_push(const char* label) { Roctx& r = roctx(); if (r.push) (void)r.push(label); }
void trace_pop() { Roctx& r = roctx(); if (r.pop) (void)r.pop(); }
#endif

}  // namespace kk

// ================================================================================================
extern "C" {

const char* kkamd_last_error(void) { return kk::last_error_ref().c_str(); }
int kkamd_version(void) { return KKAMD_VERSION; }

int kkamd_device_info(char* name, int name_len, int* is_gfx950, int* num_cus) {
  int dev = 0;
  KK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  KK_HIP(hipGetDeviceProperties(&prop, dev));
  if (name && name_len > 0) { snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName); }
  if (is_gfx950) *is_gfx950 = (strncmp(prop.gcnArchName, "gfx950", 6) == 0);
  if (num_cus) *num_cus = prop.multiProcessorCount;
  return KKAMD_OK;
}

int kkamd_trace_push(const char* label) { if (!label) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_trace_push: null label"); kk::trace_push(label); return KKAMD_OK; }
int kkamd_trace_pop(void) { kk::trace_pop(); return KKAMD_OK; }

int kkamd_set_default(const char* key, int value) {
  if (key && std::strcmp(key, "verbose") == 0) { kk::g_verbose = value != 0; return KKAMD_OK; }
  if (key && std::strncmp(key, "spgemm_", 7) == 0) return kk::spgemm_set_default(key, value);
  if (key && std::strcmp(key, "struct_remap") == 0) {
    if (value != 0 && value != 1) return kk::fail(KKAMD_ERR_INVALID_ARG, "struct_remap: %d is not 0 or 1", value);
    kk::g_struct_remap = value; return KKAMD_OK;
  }
  if (key && std::strcmp(key, "struct_group") == 0) {
    if (!kk::valid_order_knob(value) || value == 1) return kk::fail(KKAMD_ERR_INVALID_ARG, "struct_group: %d is not 0 or a power of two >= 2", value);
    kk::g_struct_group = value; return KKAMD_OK;
  }
  if (key && std::strcmp(key, "struct_strip") == 0) {
    if (value < 0) return kk::fail(KKAMD_ERR_INVALID_ARG, "struct_strip: %d is negative", value);
    kk::g_struct_strip = value; return KKAMD_OK;
  }
#ifdef KK_ABLATE
  if (key && std::strcmp(key, "struct_lds_pad_kb") == 0) { kk::g_struct_lds_pad_kb = value; return KKAMD_OK; }
#endif
  return kk::set_tuning(kk::g_spmv_default, key, value);
}

int kkamd_spmv_plan_create(kkamd_spmv_plan_t** plan, const kkamd_crs_t* A, int algorithm, kkamd_stream_t stream) {
  return kkamd_spmv_plan_create_knobs(plan, A, algorithm, nullptr, nullptr, 0, stream);
}

int kkamd_spmv_plan_create_knobs(kkamd_spmv_plan_t** plan, const kkamd_crs_t* A, int algorithm, const char* const* keys,
                                 const int* values, int nknobs, kkamd_stream_t stream) {
  if (!plan) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_plan_create: null output pointer");
  *plan = nullptr;
  int rc = kk::check_crs(A);
  if (rc) return rc;
  if (algorithm < KKAMD_SPMV_DEFAULT || algorithm > KKAMD_SPMV_NATIVE_MERGE_PATH)
    return kk::fail(KKAMD_ERR_INVALID_ARG, "SPMVHandle: algorithm %d cannot be used if A is a CrsMatrix", algorithm);
  kkamd_spmv_plan* p = new (std::nothrow) kkamd_spmv_plan();
  if (!p) return kk::fail(KKAMD_ERR_ALLOC, "kkamd_spmv_plan_create: out of host memory");
  p->num_rows = A->num_rows; p->num_cols = A->num_cols; p->nnz = A->nnz; p->row_map = A->d_row_map;
  p->entries = A->d_entries;
  p->offset_type = A->offset_type; p->value_type = A->value_type; p->algorithm = algorithm; p->tune = kk::g_spmv_default;
  for (int i = 0; i < nknobs; ++i) {                           // this plan's knobs, applied before the analysis they shape
    if (!keys || !values) { delete p; return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_plan_create_knobs: null knob arrays"); }
    if ((rc = kk::set_tuning(p->tune, keys[i], values[i]))) { delete p; return rc; }
    p->set_log.emplace_back(std::string(keys[i]), values[i]);   // (what the plan of a cached transpose will be created with)
  }
  {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      p->num_cus = prop.multiProcessorCount;
  }
  rc = kk::build_analysis(p, A, kk::to_hip(stream), 0);
  if (rc) { kkamd_spmv_plan_destroy(p); return rc; }
  *plan = p;
  return KKAMD_OK;
}

int kkamd_spmv_plan_destroy(kkamd_spmv_plan_t* plan) {
  if (!plan) return KKAMD_OK;
  // hipFree synchronises the device, so kernels still using the buffers have finished
  // (the reference's rocSPARSE sub-handle relies on the same property, spmv_handle.hpp:148-152)
  kk::free_analysis(plan);
  if (plan->d_xpack) (void)hipFree(plan->d_xpack);
  if (plan->d_ypack) (void)hipFree(plan->d_ypack);
  if (plan->d_mv2_order) (void)hipFree(plan->d_mv2_order);
  if (plan->d_mv_long) (void)hipFree(plan->d_mv_long);
  if (plan->mv) kk::mv_plan_destroy(plan->mv);
  if (plan->mv4) kk::mv4_plan_destroy(plan->mv4);
  if (plan->mv5) kk::mv5_plan_destroy(plan->mv5);
  if (plan->mv6) kk::mv6_plan_destroy(plan->mv6);
  if (plan->cs) kk::cs_plan_destroy(plan->cs);
  if (plan->d_t_rm) (void)hipFree(plan->d_t_rm);
  if (plan->d_t_ent) (void)hipFree(plan->d_t_ent);
  if (plan->d_t_perm) (void)hipFree(plan->d_t_perm);
  if (plan->d_t_fp) (void)hipFree(plan->d_t_fp);
  if (plan->d_t_shadow) (void)hipFree(plan->d_t_shadow);
  if (plan->d_t_val) (void)hipFree(plan->d_t_val);
  if (plan->t_plan) kkamd_spmv_plan_destroy(plan->t_plan);
  delete plan;
  return KKAMD_OK;
}

static int plan_set_impl(kkamd_spmv_plan_t* plan, const char* key, int value);
int kkamd_spmv_plan_set(kkamd_spmv_plan_t* plan, const char* key, int value) {
  if (!plan) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_plan_set: null plan");
  int rc = plan_set_impl(plan, key, value);
  if (rc) return rc;
  const std::string k(key ? key : "");
  if (!kk::knob_stays_with_parent(k)) {
    // one entry per key, the last value (a caller that toggles a knob every iteration must not grow the log, and the transpose, created later,
    // replays it once per key)
    bool seen = false;
    for (auto& kv : plan->set_log) if (kv.first == k) { kv.second = value; seen = true; break; }
    if (!seen) plan->set_log.emplace_back(k, value);
    // modes T / H run the mode-N dispatch on the cached transpose's own plan.  A knob that the parent took is valid there too; if its
    // side effects fail on the transpose (no memory for a new analysis, ...) the transpose is dropped -- the next T / H call builds it
    // again from the parent's knobs -- rather than left with other knobs than its parent
    if (plan->t_plan && plan_set_impl(plan->t_plan, key, value) != KKAMD_OK) kk::drop_transpose(plan);
  }
  return rc;
}
static int plan_set_impl(kkamd_spmv_plan_t* plan, const char* key, int value) {
  const kk::SpmvTuning old = plan->tune;
  int rc = kk::set_tuning(plan->tune, key, value);
  if (rc) return rc;
  const kk::SpmvTuning& t = plan->tune;
  if (t.nnz_per_thread != old.nnz_per_thread || t.kernel != old.kernel || t.stream_variant != old.stream_variant ||
      t.window_codes != old.window_codes || t.window_codes_min_knnz != old.window_codes_min_knnz ||
      t.window_codes_min_pct != old.window_codes_min_pct || t.pattern_codes != old.pattern_codes || t.pattern_direct != old.pattern_direct ||
      t.pattern_codes_min_knnz != old.pattern_codes_min_knnz) {
    // the tiling or its column analysis changed: redo the analysis from the matrix the plan is bound to
    plan->win_failed = false;
    kkamd_crs_t A{};
    A.num_rows = plan->num_rows; A.num_cols = plan->num_cols; A.nnz = plan->nnz; A.d_row_map = plan->row_map;
    A.d_entries = plan->entries; A.offset_type = plan->offset_type; A.value_type = plan->value_type;
    if (plan->used) KK_HIP(hipStreamSynchronize(plan->last_stream));
    return kk::build_analysis(plan, &A, plan->last_stream, 0);
  }
  if (t.mv_order != old.mv_order || t.mv_strip_min_kb != old.mv_strip_min_kb || t.mv_strip_l2_kb != old.mv_strip_l2_kb) {
    if (plan->d_mv2_order) { if (plan->used) KK_HIP(hipStreamSynchronize(plan->last_stream)); KK_HIP(hipFree(plan->d_mv2_order)); plan->d_mv2_order = nullptr; }
    plan->mv2_tried = false;
  }
  if ((t.mv_order != old.mv_order || t.mv_strip_min_kb != old.mv_strip_min_kb || t.mv_strip_l2_kb != old.mv_strip_l2_kb) && plan->mv) { kk::mv_plan_destroy(plan->mv); plan->mv = nullptr; plan->mv_failed = false; }
  if (t.colslab != old.colslab || t.colslab_shift != old.colslab_shift || t.colslab_min_knnz != old.colslab_min_knnz) {
    if (plan->used) KK_HIP(hipStreamSynchronize(plan->last_stream));
    if (plan->cs) { kk::cs_plan_destroy(plan->cs); plan->cs = nullptr; }
    plan->cs_tried = false; plan->cs_crs_us = plan->cs_us = 0.0;
  }
  if ((t.mv4_wg_per_cu != old.mv4_wg_per_cu || t.mv4_2d != old.mv4_2d) && (plan->mv4 || plan->mv4_tried)) {
    if (plan->used) KK_HIP(hipStreamSynchronize(plan->last_stream));
    kk::mv4_plan_destroy(plan->mv4); plan->mv4 = nullptr; plan->mv4_tried = false;
  }
  if ((t.mv5 != old.mv5 || t.mv5_min_fill_pct != old.mv5_min_fill_pct || t.mv5_max_other_pct != old.mv5_max_other_pct) && (plan->mv5 || plan->mv5_tried)) {
    if (plan->used) KK_HIP(hipStreamSynchronize(plan->last_stream));
    kk::mv5_plan_destroy(plan->mv5); plan->mv5 = nullptr; plan->mv5_tried = false;
  }
  if (t.values_tracking != old.values_tracking || t.explicit_transpose != old.explicit_transpose || t.colslab_const != old.colslab_const) {
    // another policy: whatever the copies recorded under the old one is void, the next call copies every value
    plan->t_stale = true; plan->t_fp_valid = false; plan->t_shadow_valid = false;
    kk::cs_reset_tracking(plan->cs);
  }
  if ((t.mv6 != old.mv6) && (plan->mv6 || plan->mv6_tried) && t.mv6 == 0) {
    if (plan->used) KK_HIP(hipStreamSynchronize(plan->last_stream));
    kk::mv6_plan_destroy(plan->mv6); plan->mv6 = nullptr; plan->mv6_tried = false;
  }
  if (t.mv_long_T != old.mv_long_T && plan->mv_long_known) {          // the list of long rows was made for the old threshold
    if (plan->used) KK_HIP(hipStreamSynchronize(plan->last_stream));
    if (plan->d_mv_long) { KK_HIP(hipFree(plan->d_mv_long)); plan->d_mv_long = nullptr; }
    plan->mv_long_known = false; plan->n_mv_long = 0; plan->mv_long_T = 0; plan->mv_long_nnz = 0;
  }
  return KKAMD_OK;
}

/* the caller changed A.values: re-ordered copies the plan keeps (cached transpose, column-slab copy) copy them again at the next call */
int kkamd_spmv_plan_values_changed(kkamd_spmv_plan_t* plan) {
  if (!plan) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_plan_values_changed: null plan");
  plan->t_stale = true;
  kk::cs_mark_stale(plan->cs);
  return KKAMD_OK;
}

int kkamd_spmv_plan_query(const kkamd_spmv_plan_t* plan, const char* key, int64_t* value) {
  if (!plan || !key || !value) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_plan_query: null argument");
  const std::string k(key);
  if (k.rfind("transpose_plan_", 0) == 0) {        // "transpose_plan_<key>": the same query on the plan of the cached transpose (0 while there is none)
    if (!plan->t_plan) { *value = 0; return KKAMD_OK; }
    return kkamd_spmv_plan_query(plan->t_plan, k.c_str() + 15, value);
  }
  if (k == "tile") *value = plan->tile;
  else if (k == "tiles") *value = plan->nblocks;
  else if (k == "window_codes") *value = plan->d_tinfo ? 1 : 0;
  else if (k == "window_staged_x") *value = (plan->d_tinfo && plan->staged_tiles > 0) ? 1 : 0;
  else if (k == "plain_tiles") *value = plan->d_tinfo ? plan->plain_tiles : plan->nblocks;
  else if (k == "code_tiles") *value = plan->code_tiles;
  else if (k == "staged_tiles") *value = plan->d_tinfo ? plan->staged_tiles : 0;
  else if (k == "pattern_tiles") *value = plan->d_pmeta ? plan->pat_tiles : 0;
  else if (k == "plan_bytes") *value = (int64_t)plan->plan_bytes;
  else if (k == "transpose_bytes") {            // the cached transpose of modes T / H: structure, permutation, values, fingerprints and the shadow copy of exact tracking
    const int64_t ob = plan->offset_type == KKAMD_I64 ? 8 : 4, vb = plan->value_type == KKAMD_F64 ? 8 : 4;
    *value = plan->t_ready ? plan->nnz * (4 + ob + vb) + (plan->num_cols + 1) * ob + 16 * kk::values_fp_tiles(plan->nnz) + (plan->d_t_shadow ? plan->nnz * vb : 0) : 0;
  }
  else if (k == "transpose_cached") *value = plan->t_ready ? 1 : 0;
  else if (k == "mv_tiles") *value = kk::mv_plan_query(plan->mv, 0);
  else if (k == "mv_staged_tiles") *value = kk::mv_plan_query(plan->mv, 1);
  else if (k == "mv_long_rows") *value = plan->n_mv_long;
  else if (k == "mv_order") *value = plan->mv ? kk::mv_plan_query(plan->mv, 2) : (plan->d_mv2_order ? 2 : 0);
  else if (k == "mv_period") *value = plan->mv_period;
  else if (k == "mv_plan_bytes") *value = kk::mv_plan_query(plan->mv, 3) + kk::mv4_plan_query(plan->mv4, 3) + kk::mv5_plan_query(plan->mv5, 3) + kk::mv6_plan_query(plan->mv6, 2);
  else if (k == "mv4_workgroups") *value = kk::mv4_plan_query(plan->mv4, 0);
  else if (k == "mv4_other_rows") *value = kk::mv4_plan_query(plan->mv4, 1);
  else if (k == "mv4_stencil") *value = kk::mv4_plan_query(plan->mv4, 2);
  else if (k == "mv4_near_stride") *value = kk::mv4_plan_query(plan->mv4, 4);
  else if (k == "mv6_chunks") *value = kk::mv6_plan_query(plan->mv6, 0);
  else if (k == "mv6_empty_rows") *value = kk::mv6_plan_query(plan->mv6, 1);
  else if (k == "mv_long_nnz") *value = plan->mv_long_nnz;
  else if (k == "mv5_tiles") *value = kk::mv5_plan_query(plan->mv5, 0);
  else if (k == "mv5_other_rows") *value = kk::mv5_plan_query(plan->mv5, 1);
  else if (k == "mv5_blocks") *value = kk::mv5_plan_query(plan->mv5, 2);
  else if (k == "mv5_fill_permille") *value = kk::mv5_plan_query(plan->mv5, 4);
  else if (k == "pattern_direct") *value = plan->pat_direct ? 1 : 0;
  else if (k == "colslab") *value = plan->cs ? 1 : 0;
  else if (k == "colslab_tried") *value = plan->cs_tried ? 1 : 0;
  else if (k == "colslab_slabs") *value = kk::cs_plan_query(plan->cs, 0);
  else if (k == "colslab_shift") *value = kk::cs_plan_query(plan->cs, 1);
  else if (k == "colslab_bytes") *value = kk::cs_plan_query(plan->cs, 2);
  else if (k == "colslab_deterministic") *value = kk::cs_plan_query(plan->cs, 3);
  else if (k == "colslab_lines_permille") *value = (int64_t)(plan->cs_ratio * 1000.0 + 0.5);
  else if (k == "colslab_crs_us") *value = (int64_t)(plan->cs_crs_us + 0.5);
  else if (k == "colslab_us") *value = (int64_t)(plan->cs_us + 0.5);
  else if (k == "march_workgroups") *value = plan->tune.march ? kk::mv4_plan_query(plan->mv4, 5) : 0;
  else return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_plan_query: unknown key '%s'", key);
  return KKAMD_OK;
}

/* copies a per-tile array of the plan to the host (tests pin the tile -> row search against the reference's merge-path vectors) */
int kkamd_spmv_plan_export(const kkamd_spmv_plan_t* plan, const char* what, void* h_out, int64_t count) {
  if (!plan || !what || !h_out) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_plan_export: null argument");
  const std::string k(what);
  const void* src = nullptr; int64_t have = 0; size_t esz = 4;
  if (k == "tile_first_row") { src = plan->d_blk_row; have = plan->d_blk_row ? plan->nblocks + 1 : 0; }     // bit 31: the tile starts inside a row
  else if (k == "tile_mode") { src = plan->d_tinfo; have = plan->d_tinfo ? plan->nblocks : 0; }             // low two bits: kTilePlain .. kTilePattern
  else return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_plan_export: unknown array '%s'", what);
  if (count > have) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_plan_export: '%s' has %lld entries, %lld asked for", what, (long long)have, (long long)count);
  if (count > 0) KK_HIP(hipMemcpy(h_out, src, esz * (size_t)count, hipMemcpyDeviceToHost));
  return KKAMD_OK;
}

/* frees the calling host thread's scratch of the handle-less route (tile descriptors + carries, grown on demand) */
int kkamd_release_scratch(void) { const int rc = kk::release_transient(); const int rc2 = kk::release_bitmap_pool(); return rc ? rc : rc2; }

int kkamd_spmv(kkamd_spmv_plan_t* plan, const kkamd_crs_t* A, char mode, double alpha, const void* d_x, double beta,
               void* d_y, int vector_type, kkamd_stream_t stream) {
  int rc = kk::check_crs(A);
  if (rc) return rc;
  bool trans = false;
  if ((rc = kk::parse_mode(mode, &trans))) return rc;
  if ((rc = kk::check_plan(plan, A))) return rc;
  if (vector_type != KKAMD_F32 && vector_type != KKAMD_F64)
    return kk::fail(KKAMD_ERR_UNSUPPORTED, "kkamd_spmv: unsupported vector_type %d", vector_type);
  hipStream_t st     = kk::to_hip(stream);
  const int64_t ylen = trans ? A->num_cols : A->num_rows;
  const int64_t xlen = trans ? A->num_rows : A->num_cols;
  if (ylen > 0 && !d_y) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv: null y");
  // alpha*op(A) == 0: only the y scaling happens (sparse/src/KokkosSparse_spmv.hpp:145-154)
  if (alpha == 0.0 || A->num_rows == 0 || A->num_cols == 0 || A->nnz == 0) {
    if (vector_type == KKAMD_F64) return kk::launch_scale<double>((double*)d_y, ylen, 1, 1, 1, beta, st);
    return kk::launch_scale<float>((float*)d_y, ylen, 1, 1, 1, (float)beta, st);
  }
  if (xlen > 0 && !d_x) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv: null x");
  if ((rc = kk::bind_stream(plan, st))) return rc;
  if ((rc = kk::check_entries_content(plan, A, st))) return rc;
  kk::TraceRange range(A->value_type == KKAMD_F64 ? "KokkosSparse::spmv[TPL_KKAMD,double]" : "KokkosSparse::spmv[TPL_KKAMD,float]");
  KK_DISPATCH_TYPES(kk::spmv_typed, plan, A, trans, alpha, d_x, beta, d_y, st);
}

}  // extern "C"
