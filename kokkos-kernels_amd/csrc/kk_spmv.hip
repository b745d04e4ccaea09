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

namespace kk {

struct SpmvTuning {
  int kernel         = 0;  // 0 auto, 1 vector, 2 stream
  int lanes_per_row  = 0;  // vector kernel, 0 = auto
  int nnz_per_thread = 0;  // stream kernel: 4, 8 or 16 (0 = 16 for fp64, 8 otherwise)
  int xcd_remap      = 16; // tile order of the nnz-split kernel: 0 dispatch order (tile b on XCD b % 8), 1 XCD-contiguous (3-8 % slower than 0),
                           // G >= 2 grouped (G consecutive tiles per XCD inside blocks of 8G tiles; 16: 1.37 -> 1.29 ms on C2, 7-pt 400^3 -7 %)
  int nontemporal    = 0;  // measured: no consistent gain from nt loads on the value/column streams
  int mv_kernel      = 0;  // reserved for rank-2 variants
  int stream_variant = 1;  // 1 lean kernel + quad-dealt gathers (default), 3 same with natural gather layout,
                           // 0 first-generation, 2 wave-private tiles, 4 tile-local column structure (kept for A/B)
  int wg_per_cu      = 0;  // unused (persistent variant measured slower and was removed)
  int ablate         = 0;  // bench-only: 1 no x gather, 2 no LDS/reduce, 3 both
  int lds_pad_kb     = 0;  // bench-only: extra dynamic LDS per workgroup (caps workgroups per CU)
  int mv_remap       = 16; // rank-2: 0 dispatch order, 1 XCD-contiguous (no faster than 0), 4 / 8 / 16 / ... grouped (16: 5.60 -> 4.76 ms on C3)
  int explicit_transpose = 0;   // modes T/H with an analysed handle: 1 = cache A^T (structure + permutation) in the plan, refresh its
                                // values every call and run the N kernel on it; 2 = same, the caller promises constant values (no refresh)
  int explicit_transpose_min_knnz = 1000;   // ... from this many thousand nnz
  int transient_min_knnz = 10000;  // handle-less / FAST_SETUP calls analyse on the fly from this many thousand nnz (0 = never)
  int window_codes = 1;            // analysed handles with the default kernel: try the 16-bit window codes (stream_variant 6 forces the attempt)
  int window_codes_min_knnz = 1000;  // ... from this many thousand nnz
  int pattern_codes = 1;           // staged-x plans: 1 = row-pattern records instead of per-nonzero codes when >= 90 % of the tiles have one
                                   // (27-pt 300^3 1.349 -> 1.267 ms, 7-pt 400^3 1.025 -> 0.940 ms), 2 = whenever any tile has one, 0 = never
  int pattern_codes_min_knnz = 10000;  // ... from this many thousand nnz (5-pt 1000^2, 5e6 nnz: 17.5 -> 18.3 us with the records)
};
static SpmvTuning g_spmv_default;

}  // namespace kk

struct kkamd_spmv_plan {
  int64_t num_rows = 0, num_cols = 0, nnz = 0;
  const void* row_map = nullptr;
  int offset_type = 0, algorithm = 0;
  kk::SpmvTuning tune;
  int tile = 0;             // nnz per workgroup of the analysed tiling (0 = no stream analysis)
  int64_t nblocks = 0;
  int num_cus = 256;
  int32_t* d_blk_row = nullptr;  // [nblocks+1] first row starting at or after b*tile
  void* d_carry = nullptr;       // [2*nblocks] 8-byte slots: head partials, then tail partials
  void* d_xpack = nullptr;       // rank-2: row-major packed copy of a column-major X (grown on demand)
  size_t xpack_bytes = 0;
  const void* entries = nullptr; // the matrix's column array (identity check + tile-local analysis)
  // tile-local column structure ("TLC", stream_variant 4): per tile the sorted list of DISTINCT columns and, per
  // nnz, a 16-bit index into that list.  Built once per matrix; values are still read from the caller's array.
  int64_t* d_uoff = nullptr;     // [nblocks+1] offsets into d_ucols
  int32_t* d_ucols = nullptr;    // distinct columns of every tile, ascending within a tile
  uint16_t* d_lidx = nullptr;    // [nnz] position of each nnz's column in its tile's list
  int64_t ucols_total = 0;
  // window codes (stream_variant 6): per tile up to 16 column windows of 4096 and, per nnz, a 16-bit code
  // (window << 12 | column - window base), stored in the order the kernel's work-items consume them
  uint16_t* d_wcode = nullptr;   // [nblocks * tile]
  int32_t* d_wbase = nullptr;    // [nblocks * 64] window meta: bases, LDS slots, x chunk columns
  bool win_stage = false;        // every tile's used column ranges fit its LDS x window
  int32_t* d_pmeta = nullptr;    // [nblocks * kPatW] row-pattern records (see pat_build_kernel); nseg = 0: the tile keeps its codes
  int64_t pat_tiles = 0;         // tiles with a record
  bool use_pat = false;
  // modes T/H: explicit transpose cached on first use (structure, permutation into A's values, refreshed values, sub-plan)
  void* d_t_rm = nullptr; int32_t* d_t_ent = nullptr; void* d_t_perm = nullptr; void* d_t_val = nullptr;
  kkamd_spmv_plan* t_plan = nullptr;
  bool t_ready = false, t_failed = false, t_values_valid = false;
  bool win_failed = false;       // some tile of this matrix needs more than 16 windows: plain entries
};

namespace kk {

// ------------------------------------------------------------------------------------------------
template <class YT> __global__ void scale_kernel(YT* __restrict__ y, int64_t n, int64_t s0, int64_t ncol, int64_t s1, YT beta) {
  // y(i,j) at i*s0 + j*s1; beta == 0 writes exact zeros (KokkosBlas::scal semantics)
  const int64_t total = n * ncol;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = g / ncol, j = g % ncol;
    YT* p = y + i * s0 + j * s1;
    *p = (beta == YT(0)) ? YT(0) : beta * (*p);
  }
}

template <class YT> static int launch_scale(YT* y, int64_t n, int64_t s0, int64_t ncol, int64_t s1, YT beta, hipStream_t st) {
  if (n * ncol == 0 || beta == YT(1)) return KKAMD_OK;
  const int64_t nb = ceil_div(n * ncol, kBlock);
  KK_LAUNCH((scale_kernel<YT>), (unsigned)(nb < 8192 ? nb : 8192), kBlock, 0, st, y, n, s0, ncol, s1, beta);
  KK_LAUNCH_CHECK();
  return KKAMD_OK;
}

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

// native 2-element vectors (accepted by __builtin_nontemporal_load; same syntax under clang and gcc)
typedef double kk_f64x2 __attribute__((vector_size(16)));
typedef float  kk_f32x2 __attribute__((vector_size(8)));
typedef int    kk_i32x2 __attribute__((vector_size(8)));
typedef unsigned kk_u32x2 __attribute__((vector_size(8)));
typedef unsigned kk_u32x4 __attribute__((vector_size(16)));
template <class T> struct vec2;
template <> struct vec2<double> { using type = kk_f64x2; };
template <> struct vec2<float>  { using type = kk_f32x2; };

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
                                                           int* __restrict__ fail) {
  // fail[0]: tiles that need more than 16 windows; fail[1]: tiles whose used column ranges exceed the LDS x window
  constexpr int TILE = kBlock * NPT, STEPS = NPT / 2, SPAN = kBlock * 2;
  __shared__ int s_base[kWinCount];
  __shared__ int s_len[kWinCount];
  __shared__ int s_off[kWinCount + 1];
  __shared__ unsigned s_bits[128];                           // used columns of the window being formed
  __shared__ int s_zero;
  __shared__ int s_min;
  const int t = threadIdx.x;
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
  bool uncovered = false;
  for (int pass = 0; pass < 2; ++pass) {                       // workgroup-uniform control flow throughout
    long long bound = 0;                                       // columns < bound are covered
    for (int w = 0; w < kWinCount; ++w) {
      if (t == 0) { s_min = INT_MAX; s_zero = 64; }
      if (t < 128) s_bits[t] = 0u;
      __syncthreads();
      int m = INT_MAX;
      KK_UNROLL
      for (int k = 0; k < NPT; ++k) if (c[k] >= 0 && (long long)c[k] >= bound && c[k] < m) m = c[k];
      if (m != INT_MAX) atomicMin(&s_min, m);
      __syncthreads();
      const int base = s_min;
      if (base == INT_MAX) {                                   // everything is covered: the unused windows repeat the last base
        if (t == 0) for (int q = w; q < kWinCount; ++q) s_base[q] = q ? s_base[q - 1] : 0;
        __syncthreads();
        break;
      }
      long long span = (1 << kWinBits);
      if (pass == 0) {
        KK_UNROLL
        for (int k = 0; k < NPT; ++k) {
          const long long d = (long long)c[k] - base;
          if (c[k] >= 0 && d >= 0 && d < (1 << kWinBits)) atomicOr(&s_bits[d >> 5], 1u << (d & 31));
        }
        __syncthreads();
        if (t < 64 && s_bits[2 * t] == 0u && s_bits[2 * t + 1] == 0u) atomicMin(&s_zero, t);
        __syncthreads();
        span = 64ll * s_zero;
      }
      if (t == 0) s_base[w] = base;
      bound = (long long)base + span;
      __syncthreads();
    }
    uncovered = false;
    KK_UNROLL
    for (int k = 0; k < NPT; ++k) uncovered |= (c[k] >= 0 && (long long)c[k] >= bound);
    if (t == 0) s_min = 0;
    __syncthreads();
    if (uncovered) atomicOr(&s_min, 1);
    __syncthreads();
    const int any = s_min;
    __syncthreads();
    if (!any) break;
  }
  if (uncovered) atomicAdd(fail, 1);
  if (t < kWinCount) s_len[t] = 0;
  __syncthreads();
  uint16_t* out = wcode + s + (int64_t)t * NPT;
  KK_UNROLL
  for (int k = 0; k < NPT; ++k) {
    int w = 0;
    for (int q = 1; q < kWinCount; ++q) if (s_base[q] <= c[k] && s_base[q] > s_base[q - 1]) w = q;
    const int d = c[k] - s_base[w];
    const bool ok = (c[k] >= 0 && d >= 0 && d < (1 << kWinBits));
    out[k] = ok ? (uint16_t)((w << kWinBits) | d) : (uint16_t)0;
    if (ok) atomicMax(&s_len[w], d + 1);
  }
  __syncthreads();
  // LDS slots of the used part of every window, padded to whole 64-slot chunks
  if (t == 0) {
    int off = 0;
    for (int w = 0; w < kWinCount; ++w) { s_off[w] = off; off += (s_len[w] + 63) & ~63; }
    s_off[kWinCount] = off;
    constexpr int CAP = TILE < kWinChunks * 64 ? TILE : kWinChunks * 64;
    if (off > CAP) atomicAdd(fail + 1, 1);
  }
  __syncthreads();
  int32_t* meta = wmeta + b * kWinMeta;
  if (t < kWinCount) { meta[t] = s_base[t]; meta[kWinCount + t] = s_off[t]; }
  if (t < kWinChunks) {
    const int slot = t * 64;
    int col = -1;
    for (int w = 0; w < kWinCount; ++w) if (slot >= s_off[w] && slot < s_off[w] + ((s_len[w] + 63) & ~63)) col = s_base[w] + (slot - s_off[w]);
    meta[2 * kWinCount + t] = col;
  }
}

// Row-pattern codes (on top of the staged x window).  On locally Toeplitz matrices -- stencils: column = row + a constant
// per diagonal -- the LDS slot of entry k of a row is the slot of entry k of the row before, plus one.  A tile then
// decomposes into a few SEGMENTS of consecutive rows of equal length L that share one slot table T[0..L): nonzero i of the
// tile belongs to segment g (its start sb_g <= i), j = i - (start of the segment's first row), row = j / L, k = j % L and
// its x entry sits in LDS slot T_g[k] + row.  Per tile that is kPatW ints instead of 2 bytes per nonzero.  Tiles that do not
// decompose into <= kPatSeg segments of rows with 1..kPatLen entries keep their 16-bit codes (nseg = 0).
// Tile record: [0] nseg, [1..kPatSeg-1] starts of segments 1.. (INT_MAX when unused; segment 0 starts at 0), [8 + 4g ..] {start, -first row start, L,
// float 1 / L}, [8 + 4 kPatSeg + 32 g + k] T_g[k].
constexpr int kPatSeg = 8, kPatLen = 32, kPatRec = 8, kPatTab = kPatRec + 4 * kPatSeg, kPatW = kPatTab + kPatSeg * kPatLen;

template <class OffT, int NPT>
__global__ __launch_bounds__(kBlock) void pat_build_kernel(int64_t nnz, const OffT* __restrict__ row_map,
                                                           const int32_t* __restrict__ blk_info,
                                                           const uint16_t* __restrict__ wcode, const int32_t* __restrict__ wmeta,
                                                           int32_t* __restrict__ pmeta, int* __restrict__ count) {
  constexpr int TILE = kBlock * NPT, STEPS = NPT / 2, SPAN = kBlock * 2;
  __shared__ unsigned short s_slot[TILE];
  __shared__ unsigned char s_head[TILE + 2];
  __shared__ int s_segq[kPatSeg];
  __shared__ int s_nseg, s_bad, s_chunks;
  const int t = threadIdx.x;
  const int64_t b = blockIdx.x, s = b * TILE;
  int32_t* out = pmeta + b * kPatW;
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
  if (s_bad || nseg > kPatSeg || nseg < 1) { if (t == 0) out[0] = 0; return; }      // workgroup-uniform
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
  if (t == 0) { out[0] = nseg; atomicAdd(count, 1); }
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
    out[kPatTab + t] = val;
  }
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
__device__ __forceinline__ void load_tile_codes(const uint16_t* __restrict__ wcode, int64_t ts, int t, unsigned (&w)[STEPS]) {
  constexpr int NPT = 2 * STEPS;
  const unsigned* cw = reinterpret_cast<const unsigned*>(wcode + ts + (int64_t)t * NPT);   // 4*STEPS bytes, aligned
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
__device__ __forceinline__ void load_tile_win(const AT* __restrict__ values, const uint16_t* __restrict__ wcode,
                                              const int32_t* __restrict__ wmeta, int64_t b, int64_t ts, int64_t te, int t,
                                              AT (&v0)[STEPS], AT (&v1)[STEPS], int (&c0)[STEPS], int (&c1)[STEPS]) {
  constexpr int SPAN = kBlock * 2;
  unsigned w[STEPS];
  load_tile_codes<STEPS>(wcode, ts, t, w);
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
template <class AT, class YT, int STEPS, bool FULL, bool NT, bool PAT = false>
__device__ __forceinline__ void stage_products_win(const AT* __restrict__ values, const uint16_t* __restrict__ wcode,
                                                   const int32_t* __restrict__ wmeta, const YT* __restrict__ x, int64_t ncols,
                                                   YT* prod, int64_t b, int64_t ts, int64_t te, int t,
                                                   const int32_t* __restrict__ pmeta = nullptr) {
  constexpr int SPAN = kBlock * 2, NPT = 2 * STEPS, TILE = kBlock * NPT;
  constexpr int CAPC = (TILE < kWinChunks * 64 ? TILE : kWinChunks * 64) / 64;     // chunks the LDS window can hold
  constexpr int CPW  = (CAPC + kBlock / 64 - 1) / (kBlock / 64);                     // chunks per wave
  AT v0[STEPS], v1[STEPS];
  unsigned w[STEPS];
  const int lane = t & 63, wave = t >> 6;
  const int meta = wmeta[b * kWinMeta + lane];
  // PAT: a tile with a row-pattern record (nseg > 0) needs no per-nonzero codes at all
  const int32_t* pm = PAT ? pmeta + b * kPatW : nullptr;
  const int nseg    = PAT ? pm[0] : 0;                         // workgroup-uniform
  int prec0 = 0, prec1 = 0;                                    // kPatW <= 2 * kBlock
  if (PAT) { prec0 = pm[t]; if (t + kBlock < kPatW) prec1 = pm[t + kBlock]; }
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
  if (!PAT || nseg == 0) load_tile_codes<STEPS, NT>(wcode, ts, t, w);
  KK_UNROLL
  for (int i = 0; i < CPW; ++i) {
    const int c = wave + i * (kBlock / 64);
    if (used[i]) prod[c * 64 + lane] = xv[i];
  }
  int* sseg = reinterpret_cast<int*>(prod + TILE) - kPatW;     // the record sits behind the x window (the analysis leaves room)
  if (PAT && nseg > 0) { sseg[t] = prec0; if (t + kBlock < kPatW) sseg[t + kBlock] = prec1; }
  __syncthreads();
  YT x0[STEPS], x1[STEPS];
  if (PAT && nseg > 0) {
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
          const int slot     = sseg[kPatTab + (int)(j - KK_UMUL24(row, L))] + (int)row;
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
          const int slot     = sseg[kPatTab + g * kPatLen + (int)(j - KK_UMUL24(row, L))] + (int)row;
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

// stream kernel: workgroup b owns nnz [b*TILE, (b+1)*TILE).
template <class OffT, class AT, class YT, int NPT, bool NT>
__global__ __launch_bounds__(kBlock) void spmv_stream_kernel(int64_t nnz, const OffT* __restrict__ row_map,
                                                             const int32_t* __restrict__ entries,
                                                             const AT* __restrict__ values, const YT* __restrict__ x,
                                                             YT* __restrict__ y, YT alpha, YT beta,
                                                             const int32_t* __restrict__ blk_row,
                                                             YT* __restrict__ carry_head, YT* __restrict__ carry_tail,
                                                             int remap, int ablate) {
  // ablate (bench-only diagnosis, 0 in production): bit 0 = no x gather, bit 1 = no LDS staging / row reduction
  constexpr int TILE  = kBlock * NPT;
  constexpr int STEPS = NPT / 2;      // two consecutive nnz per lane per step: 16 B of fp64 values + 8 B of columns
  __shared__ YT prod[TILE];

  const int t     = threadIdx.x;
  const int64_t b = remap ? xcd_remap(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
  const int64_t s = b * TILE;
  const int64_t e = (s + TILE < nnz) ? s + TILE : nnz;

  // 1. issue every streaming load of the tile up front (all independent, all aligned)
  AT v0[STEPS], v1[STEPS];
  int c0[STEPS], c1[STEPS];
  const bool full = (s + TILE <= nnz);
  if (full) load_tile<AT, STEPS, NT, true>(values, entries, s, e, t, v0, v1, c0, c1);
  else      load_tile<AT, STEPS, NT, false>(values, entries, s, e, t, v0, v1, c0, c1);
  // 2. gather x (L2 / Infinity-Cache resident window), multiply, stage in LDS
  if (ablate) {
    YT acc = YT(0);
    KK_UNROLL
    for (int k = 0; k < STEPS; ++k) {
      const YT x0 = (ablate & 1) ? (YT)(c0[k] & 3) : x[c0[k] < 0 ? 0 : c0[k]];
      const YT x1 = (ablate & 1) ? (YT)(c1[k] & 3) : x[c1[k] < 0 ? 0 : c1[k]];
      if (ablate & 2) { acc += (YT)v0[k] * x0 + (YT)v1[k] * x1; }
      else { prod[k * kBlock * 2 + t * 2] = (YT)v0[k] * x0; prod[k * kBlock * 2 + t * 2 + 1] = (YT)v1[k] * x1; }
    }
    if (ablate & 2) { if (acc == (YT)1.2345e30) y[0] = acc; return; }
  } else {
    if (full) stage_products<AT, YT, STEPS, true, false>(x, prod, t, v0, v1, c0, c1);
    else      stage_products<AT, YT, STEPS, false, false>(x, prod, t, v0, v1, c0, c1);
  }
  __syncthreads();

  // 3. per-row reduction out of LDS.  "Virtual rows" of this tile: an optional head (the row that
  //    started in an earlier tile) followed by the rows that start here; only the last may be cut.
  const int64_t ra         = blk_row[b] & 0x7fffffff;
  const int64_t rb         = blk_row[b + 1] & 0x7fffffff;
  const int64_t first_start = (int64_t)row_map[ra];
  const bool has_head      = first_start > s;
  const int64_t nv         = (rb - ra) + (has_head ? 1 : 0);
  int G = 1;
  while (G < kWave && nv * (G * 2) <= kBlock) G *= 2;
  const int lane = t & (G - 1), grp = t / G, ngrp = kBlock / G;
  for (int64_t base = 0; base < nv; base += ngrp) {
    const int64_t j  = base + grp;
    const bool valid = j < nv;
    YT sum           = YT(0);
    int64_t r        = -1;
    bool is_head = false, complete = false;
    if (valid) {
      const int64_t jj = has_head ? j - 1 : j;
      int64_t seg_s, seg_e;
      if (jj < 0) {
        is_head = true; seg_s = s; seg_e = first_start < e ? first_start : e;
      } else {
        r                = ra + jj;
        seg_s            = (int64_t)row_map[r];
        const int64_t re = (int64_t)row_map[r + 1];
        complete         = re <= e;
        seg_e            = complete ? re : e;
      }
      for (int i = (int)(seg_s - s) + lane; i < (int)(seg_e - s); i += G) sum += prod[i];
    }
    sum = group_sum(sum, G);
    if (valid && lane == 0) {
      if (is_head) carry_head[b] = sum;
      else if (!complete) carry_tail[b] = sum;
      else { sum *= alpha; y[r] = (beta == YT(0)) ? sum : beta * y[r] + sum; }
    }
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

// Latency-lean stream kernel (the default).  Same tiling and carry protocol as spmv_stream_kernel; what
// changes is the DEPENDENCY CHAIN each tile goes through, which is what bounds a kernel that needs ~100 KB
// in flight per CU: the tile descriptor (first row + "starts inside a row" flag, one 8-byte scalar load) is
// requested first, the streaming loads do not depend on it, and the per-lane row bounds
// row_map[r], row_map[r+1] are requested together with the x gathers -- so a tile sees two memory
// latencies (stream, then gather+bounds) instead of five (stream, gather, blk_row, row_map[ra], bounds).
// After the barrier the row reduction touches only LDS and registers.
template <class OffT, class AT, class YT, int NPT, bool NT, bool QP, int WIN = 0>
__global__ __launch_bounds__(kBlock) void spmv_stream3_kernel(int64_t nnz, const OffT* __restrict__ row_map,
                                                              const int32_t* __restrict__ entries,
                                                              const AT* __restrict__ values, const YT* __restrict__ x,
                                                              YT* __restrict__ y, YT alpha, YT beta,
                                                              const int32_t* __restrict__ blk_info,
                                                              YT* __restrict__ carry_head, YT* __restrict__ carry_tail,
                                                              int remap, int ablate, const uint16_t* __restrict__ wcode = nullptr,
                                                              const int32_t* __restrict__ wmeta = nullptr, int64_t ncols = 0,
                                                              const int32_t* __restrict__ pmeta = nullptr) {
  // WIN 1: the columns come from the plan's 16-bit window codes (wcode, wmeta) instead of entries; WIN 2: x is staged
  // in LDS from the tile's contiguous column ranges as well (stage_products_win); WIN 3: as 2, and tiles with a
  // row-pattern record (pmeta) read no per-nonzero codes at all
  // ablate (diagnosis knob, 0 in production; results in DESIGN.md 4.1): 4 = no y stores, 8 = no LDS reduction
  // loop, 16 = synthetic row bounds (no row_map loads), 32 = no barrier, 64 / 128 = y-store experiments (see the store)
  constexpr int TILE  = kBlock * NPT;
  constexpr int STEPS = NPT / 2;
  __shared__ YT prod[TILE];
  const int t     = threadIdx.x;
  const int64_t b = xcd_order(blockIdx.x, gridDim.x, remap);
  const int64_t s = b * TILE;
  const bool full = (s + TILE <= nnz);
  const int64_t e = full ? s + TILE : nnz;

  // tile descriptor: bit 31 = the tile starts inside a row (a "head" segment exists)
  const int32_t info0 = blk_info[b], info1 = blk_info[b + 1];

  AT v0[STEPS], v1[STEPS];
  int c0[STEPS], c1[STEPS];
  if (WIN >= 2) {
    // nothing to load here: stage_products_win requests values, codes and x chunks together
  } else if (WIN == 1) {
    if (full) load_tile_win<AT, STEPS, true>(values, wcode, wmeta, b, s, e, t, v0, v1, c0, c1);
    else      load_tile_win<AT, STEPS, false>(values, wcode, wmeta, b, s, e, t, v0, v1, c0, c1);
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
    if (ablate & 16) { rs = s + (int64_t)grp * 27; re = rs + 27; }
    else { rs = (int64_t)row_map[r]; re = (int64_t)row_map[r + 1]; }
  }
  // beta != 0: the old y of the first pass' rows is requested now, with everything else, instead of right before the store
  YT yold = YT(0);
  if (beta != YT(0) && valid && lane == 0 && r >= 0) yold = y[r];

  if (WIN >= 2) {
    if (full) stage_products_win<AT, YT, STEPS, true, NT, WIN == 3>(values, wcode, wmeta, x, ncols, prod, b, s, e, t, pmeta);
    else      stage_products_win<AT, YT, STEPS, false, NT, WIN == 3>(values, wcode, wmeta, x, ncols, prod, b, s, e, t, pmeta);
  } else {
    if (full) stage_products<AT, YT, STEPS, true, QP>(x, prod, t, v0, v1, c0, c1);
    else      stage_products<AT, YT, STEPS, false, false>(x, prod, t, v0, v1, c0, c1);
  }
  if (!(ablate & 32)) __syncthreads();

  for (int64_t base = 0; base < nv; base += ngrp) {
    if (base > 0) {
      valid = (base + grp) < nv;
      r     = ra + base + grp - has_head;
      if (valid) { rs = (int64_t)row_map[r]; re = (int64_t)row_map[r + 1]; }
    }
    const bool is_head  = r < ra;
    const bool complete = !is_head && re <= e;
    const int i0 = (int)((rs > s ? rs : s) - s), i1 = (int)((re < e ? re : e) - s);
    YT sum = valid ? ((ablate & 8) ? prod[i0 < TILE ? i0 : 0] : strided_lds_sum<YT>(prod, i0, i1, lane, G)) : YT(0);
    sum = group_sum(sum, G);
    if (valid && lane == 0) {
      if (is_head) carry_head[b] = sum;
      else if (!complete) carry_tail[b] = sum;
      else if (!(ablate & 4) && !((ablate & 64) && (b & 7))) {       // 64: only one tile in eight stores its rows
        sum *= alpha;
        const YT out = (beta == YT(0)) ? sum : beta * (base == 0 ? yold : y[r]) + sum;
        y[(ablate & 128) ? (r & 255) : r] = out;                         // 128: every store lands in the same 2 KB
      }
    }
  }
}

// Late-gather variant (stream_variant 5): the tile's (value, column) pairs are staged in LDS as they are, and x is
// gathered in the ROW phase -- the lanes that sum a row read its entries back from LDS and fetch x there.  With one
// lane per row, the lanes of a quad then gather the k-th entries of four CONSECUTIVE rows; on stencil-like matrices
// those are adjacent columns, i.e. one cache line and one L1 tag look-up per quad instead of the ~2 the quad-dealt
// layout of the default kernel needs (the tag look-up rate is what bounds the default kernel: 5.8e8 look-ups per
// launch on the 27-pt 300^3 matrix, ~1 per clock per CU).
template <class AT, class YT>
__device__ __forceinline__ YT strided_lds_dot(const AT* s_val, const int* s_col, const YT* __restrict__ x, int i0, int i1, int lane, int G) {
  YT s0 = YT(0), s1 = YT(0), s2 = YT(0), s3 = YT(0);
  int i = i0 + lane;
  const int G2 = 2 * G, G3 = 3 * G, G4 = 4 * G;
  for (; i + G3 < i1; i += G4) {
    const int ca = s_col[i], cb = s_col[i + G], cc = s_col[i + G2], cd = s_col[i + G3];
    const YT xa = x[ca], xb = x[cb], xc = x[cc], xd = x[cd];
    s0 += (YT)s_val[i] * xa; s1 += (YT)s_val[i + G] * xb; s2 += (YT)s_val[i + G2] * xc; s3 += (YT)s_val[i + G3] * xd;
  }
  for (; i < i1; i += G) s0 += (YT)s_val[i] * x[s_col[i]];
  return (s0 + s1) + (s2 + s3);
}
template <class OffT, class AT, class YT, int NPT, bool NT>
__global__ __launch_bounds__(kBlock) void spmv_stream6_kernel(int64_t nnz, const OffT* __restrict__ row_map,
                                                              const int32_t* __restrict__ entries,
                                                              const AT* __restrict__ values, const YT* __restrict__ x,
                                                              YT* __restrict__ y, YT alpha, YT beta,
                                                              const int32_t* __restrict__ blk_info,
                                                              YT* __restrict__ carry_head, YT* __restrict__ carry_tail, int remap) {
  constexpr int TILE  = kBlock * NPT;
  constexpr int STEPS = NPT / 2;
  constexpr int SPAN  = kBlock * 2;
  __shared__ AT s_val[TILE];
  __shared__ int s_col[TILE];
  const int t     = threadIdx.x;
  const int64_t b = remap ? xcd_remap(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
  const int64_t s = b * TILE;
  const bool full = (s + TILE <= nnz);
  const int64_t e = full ? s + TILE : nnz;
  const int32_t info0 = blk_info[b], info1 = blk_info[b + 1];
  AT v0[STEPS], v1[STEPS];
  int c0[STEPS], c1[STEPS];
  if (full) load_tile<AT, STEPS, NT, true>(values, entries, s, e, t, v0, v1, c0, c1);
  else      load_tile<AT, STEPS, NT, false>(values, entries, s, e, t, v0, v1, c0, c1);
  const int64_t ra    = info0 & 0x7fffffff;
  const int64_t rb    = info1 & 0x7fffffff;
  const int has_head  = (info0 >> 31) & 1;
  const int64_t nv    = (rb - ra) + has_head;
  int G = 1;
  while (G < kWave && nv * (G * 2) <= kBlock) G *= 2;
  const int lane = t & (G - 1), grp = t / G, ngrp = kBlock / G;
  bool valid = grp < nv;
  int64_t r  = ra + grp - has_head;
  int64_t rs = 0, re = 0;
  if (valid) { rs = (int64_t)row_map[r]; re = (int64_t)row_map[r + 1]; }
  KK_UNROLL
  for (int k = 0; k < STEPS; ++k) {
    const int li = k * SPAN + t * 2;
    s_val[li] = v0[k]; s_val[li + 1] = v1[k];
    s_col[li] = c0[k] < 0 ? 0 : c0[k]; s_col[li + 1] = c1[k] < 0 ? 0 : c1[k];     // padding entries carry value 0
  }
  __syncthreads();
  for (int64_t base = 0; base < nv; base += ngrp) {
    if (base > 0) {
      valid = (base + grp) < nv;
      r     = ra + base + grp - has_head;
      if (valid) { rs = (int64_t)row_map[r]; re = (int64_t)row_map[r + 1]; }
    }
    const bool is_head  = r < ra;
    const bool complete = !is_head && re <= e;
    const int i0 = (int)((rs > s ? rs : s) - s), i1 = (int)((re < e ? re : e) - s);
    YT sum = valid ? strided_lds_dot<AT, YT>(s_val, s_col, x, i0, i1, lane, G) : YT(0);
    sum = group_sum(sum, G);
    if (valid && lane == 0) {
      if (is_head) carry_head[b] = sum;
      else if (!complete) carry_tail[b] = sum;
      else { sum *= alpha; y[r] = (beta == YT(0)) ? sum : beta * y[r] + sum; }
    }
  }
}

// Wave-private variant: every 64-lane wave owns its own tile of 64*NPT nnz and its own slice of LDS, so the
// kernel contains no workgroup barrier at all -- a wave's load -> gather -> LDS -> reduce chain never waits
// for its three siblings, and a CU interleaves 32 independent chains instead of 8.  Same descriptor / carry
// protocol with tile = 64*NPT.
template <class OffT, class AT, class YT, int NPT, bool NT>
__global__ __launch_bounds__(kBlock) void spmv_wave_kernel(int64_t nnz, int64_t ntiles, const OffT* __restrict__ row_map,
                                                           const int32_t* __restrict__ entries,
                                                           const AT* __restrict__ values, const YT* __restrict__ x,
                                                           YT* __restrict__ y, YT alpha, YT beta,
                                                           const int32_t* __restrict__ blk_info,
                                                           YT* __restrict__ carry_head, YT* __restrict__ carry_tail) {
  constexpr int WT    = kWave * NPT;
  constexpr int STEPS = NPT / 2;
  constexpr int SPAN  = kWave * 2;
  using AV = typename vec2<AT>::type;
  __shared__ YT prod_all[kBlock / kWave][WT];
  const int lane64 = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b  = (int64_t)blockIdx.x * (kBlock / kWave) + w;
  YT* prod         = prod_all[w];
  const bool active = b < ntiles;          // wave-uniform
  const int64_t s   = b * WT;
  const bool full   = active && (s + WT <= nnz);
  const int64_t e   = full ? s + WT : nnz;

  AT v0[STEPS], v1[STEPS];
  int c0[STEPS], c1[STEPS];
  int32_t info0 = 0, info1 = 0;
  if (active) { info0 = blk_info[b]; info1 = blk_info[b + 1]; }
  KK_UNROLL
  for (int k = 0; k < STEPS; ++k) {
    const int64_t idx = s + (int64_t)k * SPAN + lane64 * 2;
    if (full) {
      const AV* vp       = reinterpret_cast<const AV*>(values + idx);
      const kk_i32x2* cp = reinterpret_cast<const kk_i32x2*>(entries + idx);
      const AV vv        = NT ? KK_NT_LOAD(vp) : *vp;
      const kk_i32x2 cc  = NT ? KK_NT_LOAD(cp) : *cp;
      v0[k] = vv[0]; v1[k] = vv[1]; c0[k] = cc[0]; c1[k] = cc[1];
    } else {
      v0[k] = (active && idx < e) ? values[idx] : AT(0);         c0[k] = (active && idx < e) ? entries[idx] : 0;
      v1[k] = (active && idx + 1 < e) ? values[idx + 1] : AT(0); c1[k] = (active && idx + 1 < e) ? entries[idx + 1] : 0;
    }
  }
  const int64_t ra   = info0 & 0x7fffffff;
  const int64_t rb   = info1 & 0x7fffffff;
  const int has_head = (info0 >> 31) & 1;
  const int64_t nv   = active ? (rb - ra) + has_head : 0;
  int G = 1;
  while (G < kWave && nv * (G * 2) <= kWave) G *= 2;
  const int lane = lane64 & (G - 1), grp = lane64 / G, ngrp = kWave / G;
  bool valid = grp < nv;
  int64_t r  = ra + grp - has_head;
  int64_t rs = 0, re = 0;
  if (valid) { rs = (int64_t)row_map[r]; re = (int64_t)row_map[r + 1]; }

  YT x0[STEPS], x1[STEPS];
  KK_UNROLL
  for (int k = 0; k < STEPS; ++k) { x0[k] = x[c0[k]]; x1[k] = x[c1[k]]; }     // masked lanes read x[0] times 0
  KK_UNROLL
  for (int k = 0; k < STEPS; ++k) {
    const int li = k * SPAN + lane64 * 2;
    prod[li]     = (YT)v0[k] * x0[k];
    prod[li + 1] = (YT)v1[k] * x1[k];
  }
  KK_WAVE_SYNC();   // the wave's own LDS writes precede its reads (LDS is in-order per wave); no s_barrier

  for (int64_t base = 0; base < nv; base += ngrp) {
    if (base > 0) {
      valid = (base + grp) < nv;
      r     = ra + base + grp - has_head;
      if (valid) { rs = (int64_t)row_map[r]; re = (int64_t)row_map[r + 1]; }
    }
    const bool is_head  = r < ra;
    const bool complete = !is_head && re <= e;
    const int i0 = (int)((rs > s ? rs : s) - s), i1 = (int)((re < e ? re : e) - s);
    YT sum = valid ? strided_lds_sum<YT>(prod, i0, i1, lane, G) : YT(0);
    sum = group_sum(sum, G);
    if (valid && lane == 0) {
      if (is_head) carry_head[b] = sum;
      else if (!complete) carry_tail[b] = sum;
      else { sum *= alpha; y[r] = (beta == YT(0)) ? sum : beta * y[r] + sum; }
    }
  }
}

// ---- tile-local column structure (TLC) ----------------------------------------------------------------
// Sorted distinct columns of tile b in LDS (uniq[0..nd)); returns nd.  keys/uniq: T ints each.
template <int T>
__device__ __forceinline__ int tile_sort_unique(const int32_t* __restrict__ entries, int64_t s, int64_t e, int* keys, int* uniq,
                                                int* s_wave) {
  const int t = threadIdx.x;
  constexpr int PER = T / kBlock;
  for (int i = t; i < T; i += kBlock) keys[i] = (s + i < e) ? entries[s + i] : INT_MAX;
  __syncthreads();
  for (int k = 2; k <= T; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < T; i += kBlock) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const int a = keys[i], b = keys[ixj];
          if ((a > b) == ((i & k) == 0)) { keys[i] = b; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  // each thread owns PER consecutive sorted keys; a key is kept when it differs from its predecessor
  int cnt = 0;
  for (int q = 0; q < PER; ++q) {
    const int i = t * PER + q;
    const int kv = keys[i];
    if (kv != INT_MAX && (i == 0 || keys[i - 1] != kv)) ++cnt;
  }
  int total;
  int pos = block_exclusive_scan<int>(cnt, &total, s_wave);
  for (int q = 0; q < PER; ++q) {
    const int i = t * PER + q;
    const int kv = keys[i];
    if (kv != INT_MAX && (i == 0 || keys[i - 1] != kv)) uniq[pos++] = kv;
  }
  __syncthreads();
  return total;
}

template <int T>
__global__ __launch_bounds__(kBlock) void tlc_count_kernel(int64_t nnz, const int32_t* __restrict__ entries, int64_t* __restrict__ uoff) {
  __shared__ int keys[T];
  __shared__ int uniq[T];
  __shared__ int s_wave[kBlock / 64];
  const int64_t b = blockIdx.x, s = b * T, e = (s + T < nnz) ? s + T : nnz;
  const int nd = tile_sort_unique<T>(entries, s, e, keys, uniq, s_wave);
  if (threadIdx.x == 0) { uoff[b] = nd; if (b == (int64_t)gridDim.x - 1) uoff[b + 1] = 0; }
}

template <int T>
__global__ __launch_bounds__(kBlock) void tlc_build_kernel(int64_t nnz, const int32_t* __restrict__ entries,
                                                           const int64_t* __restrict__ uoff, int32_t* __restrict__ ucols,
                                                           uint16_t* __restrict__ lidx) {
  __shared__ int keys[T];
  __shared__ int uniq[T];
  __shared__ int s_wave[kBlock / 64];
  const int64_t b = blockIdx.x, s = b * T, e = (s + T < nnz) ? s + T : nnz;
  const int nd = tile_sort_unique<T>(entries, s, e, keys, uniq, s_wave);
  const int64_t u0 = uoff[b];
  for (int j = threadIdx.x; j < nd; j += kBlock) ucols[u0 + j] = uniq[j];
  for (int64_t i = s + threadIdx.x; i < e; i += kBlock) {
    const int c = entries[i];
    int lo = 0, hi = nd;                       // lower bound in uniq (c is present)
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (uniq[mid] < c) lo = mid + 1; else hi = mid; }
    lidx[i] = (uint16_t)lo;
  }
}

// Stream kernel over the tile-local column structure: the tile's distinct x entries are fetched ONCE, through
// an ascending (hence well-coalesced) index list, into an LDS window; the per-nnz gather then reads LDS with a
// 16-bit local index.  Per nnz this moves 8 B (value) + 2 B (local index) + 4 B per DISTINCT column instead of
// 8 + 4, and cuts the texture-path lookups ~2-3x on matrices whose neighbouring rows share columns.  A tile with
// more than XCAP distinct columns (no reuse to exploit) falls back to direct gathers through `entries`.
template <class OffT, class AT, class YT, int NPT, int XCAP>
__global__ __launch_bounds__(kBlock) void spmv_stream5_kernel(int64_t nnz, const OffT* __restrict__ row_map,
                                                              const int32_t* __restrict__ entries,
                                                              const AT* __restrict__ values, const YT* __restrict__ x,
                                                              YT* __restrict__ y, YT alpha, YT beta,
                                                              const int32_t* __restrict__ blk_info,
                                                              const int64_t* __restrict__ uoff, const int32_t* __restrict__ ucols,
                                                              const uint16_t* __restrict__ lidx,
                                                              YT* __restrict__ carry_head, YT* __restrict__ carry_tail) {
  constexpr int TILE  = kBlock * NPT;
  constexpr int STEPS = NPT / 2;
  constexpr int SPAN  = kBlock * 2;
  using AV = typename vec2<AT>::type;
  __shared__ YT prod[TILE];
  __shared__ YT xw[XCAP];
  const int t     = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int64_t s = b * TILE;
  const bool full = (s + TILE <= nnz);
  const int64_t e = full ? s + TILE : nnz;
  const int32_t info0 = blk_info[b], info1 = blk_info[b + 1];
  const int64_t u0 = uoff[b];
  const int nd     = (int)(uoff[b + 1] - u0);
  const bool windowed = full && nd <= XCAP;                // workgroup-uniform

  AT v0[STEPS], v1[STEPS];
  int c0[STEPS], c1[STEPS];                                // local indices (windowed) or global columns (fallback)
  if (windowed) {
    KK_UNROLL
    for (int k = 0; k < STEPS; ++k) {
      const int64_t idx = s + (int64_t)k * SPAN + t * 2;
      const AV vv = *reinterpret_cast<const AV*>(values + idx);
      const unsigned pr = *reinterpret_cast<const unsigned*>(lidx + idx);     // two 16-bit local indices
      v0[k] = vv[0]; v1[k] = vv[1]; c0[k] = (int)(pr & 0xffffu); c1[k] = (int)(pr >> 16);
    }
    for (int j = t; j < nd; j += kBlock) xw[j] = x[ucols[u0 + j]];
  } else if (full) {
    load_tile<AT, STEPS, false, true>(values, entries, s, e, t, v0, v1, c0, c1);
  } else {
    load_tile<AT, STEPS, false, false>(values, entries, s, e, t, v0, v1, c0, c1);
  }
  const int64_t ra   = info0 & 0x7fffffff;
  const int64_t rb   = info1 & 0x7fffffff;
  const int has_head = (info0 >> 31) & 1;
  const int64_t nv   = (rb - ra) + has_head;
  int G = 1;
  while (G < kWave && nv * (G * 2) <= kBlock) G *= 2;
  const int lane = t & (G - 1), grp = t / G, ngrp = kBlock / G;
  bool valid = grp < nv;
  int64_t r  = ra + grp - has_head;
  int64_t rs = 0, re = 0;
  if (valid) { rs = (int64_t)row_map[r]; re = (int64_t)row_map[r + 1]; }

  if (windowed) {
    __syncthreads();                                       // x window complete
    KK_UNROLL
    for (int k = 0; k < STEPS; ++k) {
      const int li = k * SPAN + t * 2;
      prod[li]     = (YT)v0[k] * xw[c0[k]];
      prod[li + 1] = (YT)v1[k] * xw[c1[k]];
    }
  } else if (full) {
    stage_products<AT, YT, STEPS, true, true>(x, prod, t, v0, v1, c0, c1);
  } else {
    stage_products<AT, YT, STEPS, false, false>(x, prod, t, v0, v1, c0, c1);
  }
  __syncthreads();

  for (int64_t base = 0; base < nv; base += ngrp) {
    if (base > 0) {
      valid = (base + grp) < nv;
      r     = ra + base + grp - has_head;
      if (valid) { rs = (int64_t)row_map[r]; re = (int64_t)row_map[r + 1]; }
    }
    const bool is_head  = r < ra;
    const bool complete = !is_head && re <= e;
    const int i0 = (int)((rs > s ? rs : s) - s), i1 = (int)((re < e ? re : e) - s);
    YT sum = valid ? strided_lds_sum<YT>(prod, i0, i1, lane, G) : YT(0);
    sum = group_sum(sum, G);
    if (valid && lane == 0) {
      if (is_head) carry_head[b] = sum;
      else if (!complete) carry_tail[b] = sum;
      else { sum *= alpha; y[r] = (beta == YT(0)) ? sum : beta * y[r] + sum; }
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
template <class OffT, class AT, class YT, int LPRW, int RPL, int CHW>
__global__ __launch_bounds__(kBlock) void spmv_mv2_kernel(int64_t nrows, int64_t nnz, const OffT* __restrict__ row_map,
                                                          const int32_t* __restrict__ entries,
                                                          const AT* __restrict__ values, const YT* __restrict__ X,
                                                          int64_t xs0, YT* __restrict__ Y, int64_t ys0, int64_t ys1,
                                                          int64_t nvec, YT alpha, YT beta, int y_vec_ok, int remap) {
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
  const int64_t wg   = xcd_order(blockIdx.x, gridDim.x, (remap & 1) ? 1 : (remap & ~3));   // 1 contiguous, 4 / 8 / 16 grouped
  const int64_t row0 = (wg * (kBlock / kWave) + w) * RW;
  if (row0 >= nrows) return;                                   // whole wave leaves together
  const int64_t rowN = (row0 + RW < nrows) ? row0 + RW : nrows;
  const int grp = lane64 / LPRW, l = lane64 % LPRW;
  const int64_t row = row0 + grp;
  const int64_t lo  = (int64_t)row_map[row0] & ~(int64_t)3;    // 4-aligned staging windows
  const int64_t hi  = (int64_t)row_map[rowN];
  int64_t rs = 0, re = 0;
  if (row < rowN) { rs = (int64_t)row_map[row]; re = (int64_t)row_map[row + 1]; }
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
      KK_WAVE_SYNC();
      if (c + CHW <= nnz) {   // whole window inside the arrays (wave-uniform): three unguarded 16-byte loads per lane and 256 nnz
        KK_UNROLL
        for (int sub = 0; sub < CHW; sub += 256) {
          const kk_i32x4 cc = *reinterpret_cast<const kk_i32x4*>(entries + c + sub + lane64 * 4);
          const AV va = *reinterpret_cast<const AV*>(values + c + sub + lane64 * 2);
          const AV vb = *reinterpret_cast<const AV*>(values + c + sub + 128 + lane64 * 2);
          const int cm = (remap & 2) ? 255 : -1;       // bench-only ablation: every X access hits 256 rows that stay in L1
          s_col[sub + lane64 * 4] = cc[0] & cm; s_col[sub + lane64 * 4 + 1] = cc[1] & cm; s_col[sub + lane64 * 4 + 2] = cc[2] & cm; s_col[sub + lane64 * 4 + 3] = cc[3] & cm;
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

template <class OffT, class AT, class YT, int NPT, bool NT>
static int launch_stream(const kkamd_spmv_plan* p, const kkamd_crs_t* A, const YT* x, YT* y, YT alpha, YT beta,
                         hipStream_t st) {
  YT* ch = reinterpret_cast<YT*>(p->d_carry);
  YT* ct = reinterpret_cast<YT*>(reinterpret_cast<char*>(p->d_carry) + 8 * p->nblocks);
  const int variant = p->tune.stream_variant;
  if (variant == 0) {
    KK_LAUNCH((spmv_stream_kernel<OffT, AT, YT, NPT, NT>), (unsigned)p->nblocks, kBlock, 0, st, A->nnz,
              (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, x, y, alpha, beta,
              (const int32_t*)p->d_blk_row, ch, ct, p->tune.xcd_remap == 1, p->tune.ablate);
  } else if (variant == 2) {
    KK_LAUNCH((spmv_wave_kernel<OffT, AT, YT, NPT, NT>), (unsigned)ceil_div(p->nblocks, kBlock / kWave), kBlock, 0, st, A->nnz,
              p->nblocks, (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, x, y, alpha, beta,
              (const int32_t*)p->d_blk_row, ch, ct);
  } else if (variant == 4 && p->d_lidx && (NPT == 8 || NPT == 4)) {
    KK_LAUNCH((spmv_stream5_kernel<OffT, AT, YT, NPT, (NPT == 8 ? 1024 : 768)>), (unsigned)p->nblocks, kBlock, 0, st, A->nnz,
              (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, x, y, alpha, beta,
              (const int32_t*)p->d_blk_row, (const int64_t*)p->d_uoff, (const int32_t*)p->d_ucols, (const uint16_t*)p->d_lidx,
              ch, ct);
  } else if (variant == 5) {
    KK_LAUNCH((spmv_stream6_kernel<OffT, AT, YT, NPT, NT>), (unsigned)p->nblocks, kBlock, 0, st, A->nnz,
              (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, x, y, alpha, beta,
              (const int32_t*)p->d_blk_row, ch, ct, p->tune.xcd_remap == 1);
  } else if ((variant == 6 || variant == 1) && p->d_wcode) {
    if (p->win_stage && p->tune.window_codes != 2 && p->use_pat && (NPT == 16 || NPT == 8)) {
      KK_LAUNCH((spmv_stream3_kernel<OffT, AT, YT, (NPT == 4 ? 8 : NPT), NT, true, 3>), (unsigned)p->nblocks, kBlock, (size_t)p->tune.lds_pad_kb * 1024, st, A->nnz,
                (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, x, y, alpha, beta,
                (const int32_t*)p->d_blk_row, ch, ct, p->tune.xcd_remap, p->tune.ablate, (const uint16_t*)p->d_wcode,
                (const int32_t*)p->d_wbase, A->num_cols, (const int32_t*)p->d_pmeta);
    } else if (p->win_stage && p->tune.window_codes != 2) {
      KK_LAUNCH((spmv_stream3_kernel<OffT, AT, YT, NPT, NT, true, 2>), (unsigned)p->nblocks, kBlock, (size_t)p->tune.lds_pad_kb * 1024, st, A->nnz,
                (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, x, y, alpha, beta,
                (const int32_t*)p->d_blk_row, ch, ct, p->tune.xcd_remap, p->tune.ablate, (const uint16_t*)p->d_wcode,
                (const int32_t*)p->d_wbase, A->num_cols);
    } else {
      KK_LAUNCH((spmv_stream3_kernel<OffT, AT, YT, NPT, false, true, 1>), (unsigned)p->nblocks, kBlock, 0, st, A->nnz,
                (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, x, y, alpha, beta,
                (const int32_t*)p->d_blk_row, ch, ct, p->tune.xcd_remap, p->tune.ablate, (const uint16_t*)p->d_wcode,
                (const int32_t*)p->d_wbase, A->num_cols);
    }
  } else if (variant == 3) {
    KK_LAUNCH((spmv_stream3_kernel<OffT, AT, YT, NPT, NT, false>), (unsigned)p->nblocks, kBlock, 0, st, A->nnz,
              (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, x, y, alpha, beta,
              (const int32_t*)p->d_blk_row, ch, ct, p->tune.xcd_remap, p->tune.ablate);
  } else {
    KK_LAUNCH((spmv_stream3_kernel<OffT, AT, YT, NPT, NT, true>), (unsigned)p->nblocks, kBlock, (size_t)p->tune.lds_pad_kb * 1024, st, A->nnz,
              (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, x, y, alpha, beta,
              (const int32_t*)p->d_blk_row, ch, ct, p->tune.xcd_remap, p->tune.ablate);
  }
  KK_LAUNCH_CHECK();
  KK_LAUNCH((spmv_stream_fixup_kernel<OffT, YT>), (unsigned)ceil_div(p->nblocks, kBlock), kBlock, 0, st, p->nblocks,
            A->nnz, (int64_t)p->tile, (const OffT*)A->d_row_map, (const int32_t*)p->d_blk_row, (const YT*)ch,
            (const YT*)ct, y, alpha, beta);
  KK_LAUNCH_CHECK();
  return KKAMD_OK;
}

template <class OffT, class AT, class YT> struct StreamDispatch {
  // sweep variants (tile size, non-temporal loads) exist for the fp64 headline type; others use 8 / NT
  static int run(const kkamd_spmv_plan* p, const kkamd_crs_t* A, const YT* x, YT* y, YT alpha, YT beta, hipStream_t st) {
    const int npt = p->tile / (p->tune.stream_variant == 2 ? kWave : kBlock);
    if (npt == 16) return launch_stream<OffT, AT, YT, 16, false>(p, A, x, y, alpha, beta, st);
    return launch_stream<OffT, AT, YT, 8, true>(p, A, x, y, alpha, beta, st);
  }
};
template <class OffT> struct StreamDispatch<OffT, double, double> {
  static int run(const kkamd_spmv_plan* p, const kkamd_crs_t* A, const double* x, double* y, double alpha, double beta,
                 hipStream_t st) {
    const int npt = p->tile / (p->tune.stream_variant == 2 ? kWave : kBlock);
    const bool nt = p->tune.nontemporal != 0;
    if (npt == 4)  return nt ? launch_stream<OffT, double, double, 4, true>(p, A, x, y, alpha, beta, st)
                             : launch_stream<OffT, double, double, 4, false>(p, A, x, y, alpha, beta, st);
    if (npt == 16) return nt ? launch_stream<OffT, double, double, 16, true>(p, A, x, y, alpha, beta, st)
                             : launch_stream<OffT, double, double, 16, false>(p, A, x, y, alpha, beta, st);
    return nt ? launch_stream<OffT, double, double, 8, true>(p, A, x, y, alpha, beta, st)
              : launch_stream<OffT, double, double, 8, false>(p, A, x, y, alpha, beta, st);
  }
};

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
// Modes T / H: the reference scatters with atomics (spmv_impl.hpp:305-420) and so does the default here (11.1 ms on
// 27-pt 300^3).  Opt-in (knob explicit_transpose): the plan caches the TRANSPOSE once -- structure plus, for
// every entry of A^T, the position of its value in A -- and runs the planned N kernel on A^T: deterministic, no atomics.
//   explicit_transpose = 1: the transposed values are refreshed with one gather per call (the matrix values may change
//     between calls, the structure may not): 8.0 ms (the gather pulls a 64 B sector per 8 B value);
//   explicit_transpose = 2: the caller promises constant values, no refresh: 1.7 ms.
// Costs nnz * (4 + sizeof(offset) + sizeof(value)) bytes of plan memory -- which is why it is not the default -- and
// 0.3 s once for the transpose; falls back to the atomic kernel if the memory cannot be had.
__global__ void iota_f64_kernel(double* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
template <class OffT> __global__ void perm_from_f64_kernel(const double* __restrict__ src, OffT* __restrict__ perm, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) perm[i] = (OffT)src[i];
}
template <class OffT, class AT>
__global__ void gather_values_kernel(const AT* __restrict__ val, const OffT* __restrict__ perm, AT* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = val[perm[i]];
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
    if (p->d_t_val) (void)hipFree(p->d_t_val);
    p->d_t_rm = nullptr; p->d_t_ent = nullptr; p->d_t_perm = nullptr; p->d_t_val = nullptr;
    p->t_failed = true;
    return KKAMD_ERR_ALLOC;
  };
  if (hipMalloc(&p->d_t_rm, sizeof(OffT) * (size_t)(A->num_cols + 1)) != hipSuccess || hipMalloc((void**)&p->d_t_ent, sizeof(int32_t) * nnz) != hipSuccess ||
      hipMalloc(&p->d_t_perm, sizeof(OffT) * nnz) != hipSuccess || hipMalloc(&p->d_t_val, sizeof(AT) * nnz) != hipSuccess ||
      hipMalloc((void**)&d_iota, sizeof(double) * nnz) != hipSuccess || hipMalloc((void**)&d_tmp, sizeof(double) * nnz) != hipSuccess)
    return fail_clean();
  const unsigned grid = (unsigned)(ceil_div(A->nnz, kBlock) < 65536 ? ceil_div(A->nnz, kBlock) : 65536);
  KK_LAUNCH(iota_f64_kernel, grid, kBlock, 0, st, d_iota, A->nnz);
  // positions are exact in fp64 (nnz < 2^53), so the transpose of the "values" 0..nnz-1 IS the permutation
  int rc = kkamd_transpose(A->num_rows, A->num_cols, A->nnz, A->d_row_map, (const int32_t*)A->d_entries, d_iota, A->offset_type, KKAMD_F64,
                           p->d_t_rm, p->d_t_ent, d_tmp, reinterpret_cast<kkamd_stream_t>(st));
  if (rc != KKAMD_OK) { fail_clean(); return rc; }
  KK_LAUNCH((perm_from_f64_kernel<OffT>), grid, kBlock, 0, st, (const double*)d_tmp, (OffT*)p->d_t_perm, A->nnz);
  if (hipStreamSynchronize(st) != hipSuccess) return fail_clean();
  (void)hipFree(d_iota); (void)hipFree(d_tmp); d_iota = d_tmp = nullptr;
  kkamd_crs_t At{A->num_cols, A->num_rows, A->nnz, p->d_t_rm, p->d_t_ent, p->d_t_val, A->offset_type, A->value_type};
  rc = kkamd_spmv_plan_create(&p->t_plan, &At, p->algorithm, reinterpret_cast<kkamd_stream_t>(st));
  if (rc != KKAMD_OK) { fail_clean(); return rc; }
  p->t_ready = true;
  return KKAMD_OK;
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
template <class OffT>
static kkamd_spmv_plan* transient_plan(const kkamd_crs_t* A, const SpmvTuning& tn, int elem_size, hipStream_t st) {
  static thread_local TransientPlan tp;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  kkamd_spmv_plan& p = tp.plan;
  if (tp.used && (tp.device != dev || tp.last_stream != st)) {
    (void)hipStreamSynchronize(tp.last_stream);
    if (tp.device != dev) { p.d_blk_row = nullptr; p.d_carry = nullptr; tp.cap_blk = tp.cap_carry = 0; }   // other device's buffers are abandoned
  }
  p.num_rows = A->num_rows; p.num_cols = A->num_cols; p.nnz = A->nnz; p.row_map = A->d_row_map; p.entries = A->d_entries;
  p.offset_type = A->offset_type; p.algorithm = KKAMD_SPMV_FAST_SETUP; p.tune = tn;
  p.tune.stream_variant = 1;
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

template <class OffT, class AT, class YT>
static int spmv_typed(kkamd_spmv_plan* plan, const kkamd_crs_t* A, bool trans, double alpha_d, const void* dx,
                      double beta_d, void* dy, hipStream_t st) {
  const YT alpha = (YT)alpha_d, beta = (YT)beta_d;
  const YT* x    = (const YT*)dx;
  YT* y          = (YT*)dy;
  if (trans) {
    if (plan && plan->tile != 0 && plan->tune.explicit_transpose && A->nnz >= (int64_t)plan->tune.explicit_transpose_min_knnz * 1000 &&
        ensure_transpose<OffT, AT>(plan, A, st) == KKAMD_OK) {
      if (plan->tune.explicit_transpose != 2 || !plan->t_values_valid) {
        const unsigned grid = (unsigned)(ceil_div(A->nnz, kBlock) < 65536 ? ceil_div(A->nnz, kBlock) : 65536);
        KK_LAUNCH((gather_values_kernel<OffT, AT>), grid, kBlock, 0, st, (const AT*)A->d_values, (const OffT*)plan->d_t_perm, (AT*)plan->d_t_val, A->nnz);
        plan->t_values_valid = true;
      }
      kkamd_crs_t At{A->num_cols, A->num_rows, A->nnz, plan->d_t_rm, plan->d_t_ent, plan->d_t_val, A->offset_type, A->value_type};
      return spmv_typed<OffT, AT, YT>(plan->t_plan, &At, false, alpha_d, dx, beta_d, dy, st);
    }
    return run_transpose<OffT, AT, YT>(A, x, y, alpha, beta, st);
  }
  if (stream_usable(plan, A, (int)sizeof(AT))) return StreamDispatch<OffT, AT, YT>::run(plan, A, x, y, alpha, beta, st);
  // No analysed plan (handle-less overloads, SPMV_FAST_SETUP): a large matrix is still worth the nnz-split kernel --
  // its "analysis" is one tiny kernel (a binary search per 4096-nnz tile) into a per-thread scratch that is reused
  // from call to call, so nothing is allocated or kept per matrix and the call stays asynchronous.
  const SpmvTuning& tn = plan ? plan->tune : g_spmv_default;
  if (tn.kernel != 1 && tn.transient_min_knnz > 0 && A->nnz >= (int64_t)tn.transient_min_knnz * 1000 && A->num_rows > 0) {
    kkamd_spmv_plan* tp = transient_plan<OffT>(A, tn, (int)sizeof(AT), st);
    if (tp && stream_usable(tp, A, (int)sizeof(AT))) return StreamDispatch<OffT, AT, YT>::run(tp, A, x, y, alpha, beta, st);
  }
  return run_vector<OffT, AT, YT>(A, x, y, alpha, beta, tn, st);
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

template <class OffT, class AT, class YT>
static int spmv_mv_typed(kkamd_spmv_plan* plan, const kkamd_crs_t* A, bool trans, double alpha_d, const void* dX,
                         int64_t xs0, int64_t xs1, double beta_d, void* dY, int64_t ys0, int64_t ys1, int64_t nvec,
                         hipStream_t st) {
  const YT alpha = (YT)alpha_d, beta = (YT)beta_d;
  const YT* X    = (const YT*)dX;
  YT* Y          = (YT*)dY;
  const int remap = (plan ? plan->tune.xcd_remap : g_spmv_default.xcd_remap) == 1;
  if (trans) {
    int rc = launch_scale<YT>(Y, A->num_cols, ys0, nvec, ys1, beta, st);
    if (rc) return rc;
    KK_LAUNCH((spmv_mv_transpose_kernel<OffT, AT, YT>), (unsigned)ceil_div(A->num_rows, kBlock / 16), kBlock, 0, st,
              A->num_rows, (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, X, xs0, xs1,
              Y, ys0, ys1, nvec, alpha);
    KK_LAUNCH_CHECK();
    return KKAMD_OK;
  }
  const int mvk = plan ? plan->tune.mv_kernel : g_spmv_default.mv_kernel;      // 0 auto, 1 force generic, 2 force packed/row-major
  const bool a_aligned = ((uintptr_t)A->d_values % 16 == 0) && ((uintptr_t)A->d_entries % 16 == 0);
  if (mvk != 1 && a_aligned) {
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
    if (Xr) {
      const int yv = (ys1 == 1 && (ys0 % 2 == 0) && ((uintptr_t)Y % 16 == 0)) ? 1 : 0;
      const int mv_remap = plan ? plan->tune.mv_remap : g_spmv_default.mv_remap;
#define KK_MV2C(L, R, C)                                                                                                 \
      do {                                                                                                               \
        KK_LAUNCH((spmv_mv2_kernel<OffT, AT, YT, L, R, C>), (unsigned)ceil_div(A->num_rows, (kBlock / kWave) * (kWave / L)), kBlock, \
                  0, st, A->num_rows, A->nnz, (const OffT*)A->d_row_map, (const int32_t*)A->d_entries, (const AT*)A->d_values, \
                  Xr, ldx, Y, ys0, ys1, nvec, alpha, beta, yv, mv_remap);                                                        \
        KK_LAUNCH_CHECK();                                                                                               \
        return KKAMD_OK;                                                                                                 \
      } while (0)
      // staging window: the nnz of the wave's kWave/L rows (+15 % and the 4-alignment slack), rounded up to 256 / 512 / 1024
#define KK_MV2(L, R)                                                                                                     \
      do {                                                                                                               \
        const int64_t need = (int64_t)(1.15 * (double)(kWave / L) * (double)A->nnz / (double)A->num_rows) + 4;            \
        if (need <= 256) KK_MV2C(L, R, 256);                                                                              \
        if (need <= 512) KK_MV2C(L, R, 512);                                                                              \
        KK_MV2C(L, R, 1024);                                                                                              \
      } while (0)
      if (mvk == 3) { if (nvec >= 12) KK_MV2(8, 2); }                      // A/B: 8 lanes x 2 RHS
      if (mvk == 4) { if (nvec >= 12) KK_MV2(2, 4); }                      // A/B: 2 lanes x 4 RHS (strips of 8)
      if (nvec >= 12) KK_MV2(4, 4);
      if (nvec >= 6) KK_MV2(2, 4);
      if (nvec >= 3) KK_MV2(2, 2);
      KK_MV2(1, 2);
#undef KK_MV2C
#undef KK_MV2
    }
  }
  if (nvec >= 12) return launch_mv<OffT, AT, YT, 16>(A, X, xs0, xs1, Y, ys0, ys1, nvec, alpha, beta, remap, st);
  if (nvec >= 6)  return launch_mv<OffT, AT, YT, 8>(A, X, xs0, xs1, Y, ys0, ys1, nvec, alpha, beta, remap, st);
  if (nvec >= 3)  return launch_mv<OffT, AT, YT, 4>(A, X, xs0, xs1, Y, ys0, ys1, nvec, alpha, beta, remap, st);
  return launch_mv<OffT, AT, YT, 2>(A, X, xs0, xs1, Y, ys0, ys1, nvec, alpha, beta, remap, st);
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

static int check_plan(const kkamd_spmv_plan* p, const kkamd_crs_t* A) {
  if (!p) return KKAMD_OK;
  if (p->num_rows != A->num_rows || p->num_cols != A->num_cols || p->nnz != A->nnz || p->row_map != A->d_row_map ||
      p->offset_type != A->offset_type || ((p->d_wcode || p->d_lidx) && p->entries != A->d_entries))
    return fail(KKAMD_ERR_STATE, "kkamd_spmv: plan was created for a different matrix (a handle is bound to one matrix)");
  return KKAMD_OK;
}

int parse_mode(char mode, bool* trans) {
  switch (mode) {
    case 'N': case 'n': case 'C': case 'c': *trans = false; return KKAMD_OK;
    case 'T': case 't': case 'H': case 'h': *trans = true; return KKAMD_OK;
    default: return fail(KKAMD_ERR_INVALID_ARG, "Invalid transpose mode %c for KokkosSparse::spmv()", mode);
  }
}

#define KK_DISPATCH_TYPES(FN, ...)                                                                        \
  do {                                                                                                    \
    const bool o64 = A->offset_type == KKAMD_I64;                                                         \
    if (A->value_type == KKAMD_F64 && vector_type == KKAMD_F64)                                           \
      return o64 ? FN<int64_t, double, double>(__VA_ARGS__) : FN<int32_t, double, double>(__VA_ARGS__);   \
    if (A->value_type == KKAMD_F32 && vector_type == KKAMD_F32)                                           \
      return o64 ? FN<int64_t, float, float>(__VA_ARGS__) : FN<int32_t, float, float>(__VA_ARGS__);       \
    if (A->value_type == KKAMD_F32 && vector_type == KKAMD_F64)                                           \
      return o64 ? FN<int64_t, float, double>(__VA_ARGS__) : FN<int32_t, float, double>(__VA_ARGS__);     \
    return kk::fail(KKAMD_ERR_UNSUPPORTED, "kkamd_spmv: unsupported (value,vector) type pair (%d,%d)",        \
                A->value_type, vector_type);                                                              \
  } while (0)

static int set_tuning(SpmvTuning& t, const char* key, int value) {
  if (!key) return fail(KKAMD_ERR_INVALID_ARG, "null key");
  const std::string k(key);
  if (k == "kernel") t.kernel = value;
  else if (k == "lanes_per_row") t.lanes_per_row = value;
  else if (k == "nnz_per_thread") t.nnz_per_thread = value;
  else if (k == "xcd_remap") t.xcd_remap = value;
  else if (k == "nontemporal") t.nontemporal = value;
  else if (k == "mv_kernel") t.mv_kernel = value;
  else if (k == "stream_variant") t.stream_variant = value;
  else if (k == "wg_per_cu") t.wg_per_cu = value;
  else if (k == "ablate") t.ablate = value;
  else if (k == "mv_remap") t.mv_remap = value;
  else if (k == "window_codes") t.window_codes = value;
  else if (k == "window_codes_min_knnz") t.window_codes_min_knnz = value;
  else if (k == "pattern_codes") t.pattern_codes = value;
  else if (k == "pattern_codes_min_knnz") t.pattern_codes_min_knnz = value;
  else if (k == "transient_min_knnz") t.transient_min_knnz = value;
  else if (k == "explicit_transpose") t.explicit_transpose = value;
  else if (k == "explicit_transpose_min_knnz") t.explicit_transpose_min_knnz = value;
  else if (k == "lds_pad_kb") t.lds_pad_kb = value;
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

static int build_analysis(kkamd_spmv_plan* p, const kkamd_crs_t* A, hipStream_t st, int force_npt = 0) {
  if (p->d_blk_row) { KK_HIP(hipStreamSynchronize(st)); KK_HIP(hipFree(p->d_blk_row)); p->d_blk_row = nullptr; }
  if (p->d_carry) { KK_HIP(hipFree(p->d_carry)); p->d_carry = nullptr; }
  if (p->d_uoff) { KK_HIP(hipFree(p->d_uoff)); p->d_uoff = nullptr; }
  if (p->d_ucols) { KK_HIP(hipFree(p->d_ucols)); p->d_ucols = nullptr; }
  if (p->d_lidx) { KK_HIP(hipFree(p->d_lidx)); p->d_lidx = nullptr; }
  if (p->d_wcode) { KK_HIP(hipFree(p->d_wcode)); p->d_wcode = nullptr; }
  if (p->d_wbase) { KK_HIP(hipFree(p->d_wbase)); p->d_wbase = nullptr; }
  if (p->d_pmeta) { KK_HIP(hipFree(p->d_pmeta)); p->d_pmeta = nullptr; }
  p->pat_tiles = 0; p->use_pat = false;
  p->tile = 0; p->nblocks = 0;
  if (p->algorithm == KKAMD_SPMV_FAST_SETUP || p->tune.kernel == 1 || A->nnz == 0 || A->num_rows == 0) return KKAMD_OK;
  int npt = p->tune.nnz_per_thread;
  // auto: 4096-nnz tiles once there are plenty of them (measured best from ~2e8 nnz up), 2048-nnz tiles below that
  // (5-pt 1000^2: 19.2 vs 22.2 us)
  const bool want_win = !p->win_failed && p->entries &&
                        (p->tune.stream_variant == 6 || (p->tune.stream_variant == 1 && p->tune.window_codes &&
                                                         A->nnz >= (int64_t)p->tune.window_codes_min_knnz * 1000));
  const bool auto_npt = (npt != 4 && npt != 8 && npt != 16);
  // With window codes: 4096-nnz tiles when the matrix is large and their x windows fit LDS (27-pt 300^3: 1.38 ms against
  // 1.44 with 2048-nnz tiles), else 2048-nnz tiles (codes without staged x: 1.51 vs 1.58 ms; 7-pt 400^3 is only
  // stageable at 2048) -- the 4096 attempt is redone at 2048 below when it does not stage.
  if (force_npt) npt = force_npt;
  else if (auto_npt) {
    if (p->tune.stream_variant == 4) npt = 8;
    else if (want_win) npt = (A->nnz >= 50000000) ? 16 : 8;
    else npt = (A->nnz < 200000000) ? 8 : 16;
  }
  if (p->tune.stream_variant == 4 && npt == 16) npt = 8;     // the tile-local structure uses 2048- or 1024-nnz tiles
  if (!(A->value_type == KKAMD_F64) && p->tune.nnz_per_thread != 16) npt = 8;   // fp32 values: 2048-nnz tiles unless asked
  p->tile    = (p->tune.stream_variant == 2 ? kWave : kBlock) * npt;
  p->nblocks = ceil_div(A->nnz, p->tile);
  KK_HIP(hipMalloc((void**)&p->d_blk_row, sizeof(int32_t) * (size_t)(p->nblocks + 1)));
  KK_HIP(hipMalloc(&p->d_carry, (size_t)16 * (size_t)p->nblocks));
  int rc = A->offset_type == KKAMD_I64 ? analyse<int64_t>(p, A, st) : analyse<int32_t>(p, A, st);
  if (rc) return rc;
  if (p->tune.stream_variant == 4 && p->entries && (npt == 8 || npt == 4) && ((uintptr_t)p->entries % 8 == 0)) {
    // tile-local column structure: count distinct columns per tile, scan, then emit lists + 16-bit local indices
    KK_HIP(hipMalloc((void**)&p->d_uoff, sizeof(int64_t) * (size_t)(p->nblocks + 1)));
    if (npt == 8) { KK_LAUNCH((tlc_count_kernel<2048>), (unsigned)p->nblocks, kBlock, 0, st, A->nnz, (const int32_t*)p->entries, p->d_uoff); }
    else          { KK_LAUNCH((tlc_count_kernel<1024>), (unsigned)p->nblocks, kBlock, 0, st, A->nnz, (const int32_t*)p->entries, p->d_uoff); }
    KK_LAUNCH_CHECK();
    if ((rc = exclusive_scan_inplace<int64_t>(p->d_uoff, p->nblocks + 1, st))) return rc;
    KK_HIP(hipMemcpyAsync(&p->ucols_total, p->d_uoff + p->nblocks, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
    KK_HIP(hipMalloc((void**)&p->d_ucols, sizeof(int32_t) * (size_t)(p->ucols_total > 0 ? p->ucols_total : 1)));
    KK_HIP(hipMalloc((void**)&p->d_lidx, sizeof(uint16_t) * (size_t)(A->nnz + 8)));
    if (npt == 8) { KK_LAUNCH((tlc_build_kernel<2048>), (unsigned)p->nblocks, kBlock, 0, st, A->nnz, (const int32_t*)p->entries, (const int64_t*)p->d_uoff, p->d_ucols, p->d_lidx); }
    else          { KK_LAUNCH((tlc_build_kernel<1024>), (unsigned)p->nblocks, kBlock, 0, st, A->nnz, (const int32_t*)p->entries, (const int64_t*)p->d_uoff, p->d_ucols, p->d_lidx); }
    KK_LAUNCH_CHECK();
  }
  if (want_win) {
    // window codes: every tile must be coverable by 16 windows, otherwise the plan keeps reading entries
    int* d_fail = nullptr;
    int h_fails[2] = {0, 0};
    KK_HIP(hipMalloc((void**)&d_fail, 2 * sizeof(int)));
    KK_HIP(hipMemsetAsync(d_fail, 0, 2 * sizeof(int), st));
    // the codes are an optimisation: if HBM cannot hold them (2 bytes per nonzero) the plan simply keeps reading entries
    if (hipMalloc((void**)&p->d_wcode, sizeof(uint16_t) * (size_t)p->nblocks * (size_t)p->tile) != hipSuccess ||
        hipMalloc((void**)&p->d_wbase, sizeof(int32_t) * (size_t)p->nblocks * kWinMeta) != hipSuccess) {
      (void)hipGetLastError();
      if (p->d_wcode) { (void)hipFree(p->d_wcode); p->d_wcode = nullptr; }
      p->d_wbase = nullptr;
      (void)hipFree(d_fail);
      p->win_failed = true;
      if (auto_npt && !force_npt) return build_analysis(p, A, st);
      KK_HIP(hipStreamSynchronize(st));
      return KKAMD_OK;
    }
    if (npt == 16)     { KK_LAUNCH((win_build_kernel<16>), (unsigned)p->nblocks, kBlock, 0, st, A->nnz, (const int32_t*)p->entries, p->d_wcode, p->d_wbase, d_fail); }
    else if (npt == 8) { KK_LAUNCH((win_build_kernel<8>), (unsigned)p->nblocks, kBlock, 0, st, A->nnz, (const int32_t*)p->entries, p->d_wcode, p->d_wbase, d_fail); }
    else               { KK_LAUNCH((win_build_kernel<4>), (unsigned)p->nblocks, kBlock, 0, st, A->nnz, (const int32_t*)p->entries, p->d_wcode, p->d_wbase, d_fail); }
    KK_LAUNCH_CHECK();
    KK_HIP(hipMemcpyAsync(h_fails, d_fail, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
    KK_HIP(hipFree(d_fail));
    const int h_fail = h_fails[0];
    p->win_stage = (h_fails[0] == 0 && h_fails[1] == 0);
    if (h_fail) {
      KK_HIP(hipFree(p->d_wcode)); p->d_wcode = nullptr;
      KK_HIP(hipFree(p->d_wbase)); p->d_wbase = nullptr;
      p->win_failed = true;
      if (auto_npt && !force_npt) return build_analysis(p, A, st);   // the tile size was picked for the codes: redo
    } else if (!p->win_stage && auto_npt && !force_npt && npt == 16) {
      return build_analysis(p, A, st, 8);
    } else if (p->win_stage && p->tune.pattern_codes && (npt == 16 || npt == 8) &&
               (p->tune.pattern_codes >= 2 || A->nnz >= (int64_t)p->tune.pattern_codes_min_knnz * 1000) &&
               hipMalloc((void**)&p->d_pmeta, sizeof(int32_t) * (size_t)p->nblocks * kPatW) == hipSuccess) {
      // row-pattern records: which tiles decompose into a few segments of equal rows with a common slot table
      int* d_cnt = nullptr; int h_cnt = 0;
      KK_HIP(hipMalloc((void**)&d_cnt, sizeof(int)));
      KK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int), st));
      const bool o64 = A->offset_type == KKAMD_I64;
#define KK_PAT_BUILD(OT, N)                                                                                                  \
  KK_LAUNCH((pat_build_kernel<OT, N>), (unsigned)p->nblocks, kBlock, 0, st, A->nnz, (const OT*)A->d_row_map,                 \
            (const int32_t*)p->d_blk_row, (const uint16_t*)p->d_wcode, (const int32_t*)p->d_wbase, p->d_pmeta, d_cnt)
      if (npt == 16) { if (o64) { KK_PAT_BUILD(int64_t, 16); } else { KK_PAT_BUILD(int32_t, 16); } }
      else           { if (o64) { KK_PAT_BUILD(int64_t, 8); } else { KK_PAT_BUILD(int32_t, 8); } }
#undef KK_PAT_BUILD
      KK_LAUNCH_CHECK();
      KK_HIP(hipMemcpyAsync(&h_cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost, st));
      KK_HIP(hipStreamSynchronize(st));
      KK_HIP(hipFree(d_cnt));
      p->pat_tiles = h_cnt;
      p->use_pat   = p->tune.pattern_codes >= 2 ? h_cnt > 0 : (double)h_cnt >= 0.9 * (double)p->nblocks;
      if (!p->use_pat) { KK_HIP(hipFree(p->d_pmeta)); p->d_pmeta = nullptr; }
    } else {
      (void)hipGetLastError();
    }
  }
  KK_HIP(hipStreamSynchronize(st));   // setup is synchronous, like the vendor analysis it replaces
  return KKAMD_OK;
}

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

int kkamd_set_default(const char* key, int value) {
  if (key && std::strncmp(key, "spgemm_", 7) == 0) return kk::spgemm_set_default(key, value);
  if (key && std::strcmp(key, "struct_remap") == 0) { kk::g_struct_remap = value; return KKAMD_OK; }
  if (key && std::strcmp(key, "struct_group") == 0) { kk::g_struct_group = value; return KKAMD_OK; }
  if (key && std::strcmp(key, "struct_strip") == 0) { kk::g_struct_strip = value; return KKAMD_OK; }
  if (key && std::strcmp(key, "struct_lds_pad_kb") == 0) { kk::g_struct_lds_pad_kb = value; return KKAMD_OK; }
  return kk::set_tuning(kk::g_spmv_default, key, value);
}

int kkamd_spmv_plan_create(kkamd_spmv_plan_t** plan, const kkamd_crs_t* A, int algorithm, kkamd_stream_t stream) {
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
  p->offset_type = A->offset_type; p->algorithm = algorithm; p->tune = kk::g_spmv_default;
  {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      p->num_cus = prop.multiProcessorCount;
  }
  rc = kk::build_analysis(p, A, kk::to_hip(stream));
  if (rc) { kkamd_spmv_plan_destroy(p); return rc; }
  *plan = p;
  return KKAMD_OK;
}

int kkamd_spmv_plan_destroy(kkamd_spmv_plan_t* plan) {
  if (!plan) return KKAMD_OK;
  // hipFree synchronises the device, so kernels still using the buffers have finished
  // (the reference's rocSPARSE sub-handle relies on the same property, spmv_handle.hpp:148-152)
  if (plan->d_blk_row) (void)hipFree(plan->d_blk_row);
  if (plan->d_carry) (void)hipFree(plan->d_carry);
  if (plan->d_xpack) (void)hipFree(plan->d_xpack);
  if (plan->d_uoff) (void)hipFree(plan->d_uoff);
  if (plan->d_ucols) (void)hipFree(plan->d_ucols);
  if (plan->d_lidx) (void)hipFree(plan->d_lidx);
  if (plan->d_wcode) (void)hipFree(plan->d_wcode);
  if (plan->d_wbase) (void)hipFree(plan->d_wbase);
  if (plan->d_pmeta) (void)hipFree(plan->d_pmeta);
  if (plan->d_t_rm) (void)hipFree(plan->d_t_rm);
  if (plan->d_t_ent) (void)hipFree(plan->d_t_ent);
  if (plan->d_t_perm) (void)hipFree(plan->d_t_perm);
  if (plan->d_t_val) (void)hipFree(plan->d_t_val);
  if (plan->t_plan) kkamd_spmv_plan_destroy(plan->t_plan);
  delete plan;
  return KKAMD_OK;
}

int kkamd_spmv_plan_set(kkamd_spmv_plan_t* plan, const char* key, int value) {
  if (!plan) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_plan_set: null plan");
  const int old_npt = plan->tune.nnz_per_thread, old_kernel = plan->tune.kernel, old_var = plan->tune.stream_variant;
  const int old_win = plan->tune.window_codes, old_pat = plan->tune.pattern_codes;
  int rc = kk::set_tuning(plan->tune, key, value);
  if (rc) return rc;
  if (plan->tune.nnz_per_thread != old_npt || plan->tune.kernel != old_kernel ||
      (plan->tune.stream_variant == 2) != (old_var == 2) || (plan->tune.stream_variant == 4) != (old_var == 4) ||
      (plan->tune.stream_variant == 6) != (old_var == 6) || (plan->tune.stream_variant == 1) != (old_var == 1) ||
      plan->tune.window_codes != old_win || plan->tune.pattern_codes != old_pat) {
    // tiling changed: redo the analysis (needs the matrix again; rebuilt lazily from the stored row_map)
    plan->win_failed = false;
    kkamd_crs_t A{};
    A.num_rows = plan->num_rows; A.num_cols = plan->num_cols; A.nnz = plan->nnz; A.d_row_map = plan->row_map;
    A.offset_type = plan->offset_type; A.value_type = KKAMD_F64;
    return kk::build_analysis(plan, &A, nullptr);
  }
  return KKAMD_OK;
}

int kkamd_spmv_plan_query(const kkamd_spmv_plan_t* plan, const char* key, int64_t* value) {
  if (!plan || !key || !value) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_plan_query: null argument");
  const std::string k(key);
  if (k == "tile") *value = plan->tile;
  else if (k == "tiles") *value = plan->nblocks;
  else if (k == "window_codes") *value = plan->d_wcode ? 1 : 0;
  else if (k == "window_staged_x") *value = (plan->d_wcode && plan->win_stage && plan->tune.window_codes != 2) ? 1 : 0;
  else if (k == "pattern_tiles") *value = plan->use_pat ? plan->pat_tiles : 0;
  else if (k == "transpose_cached") *value = plan->t_ready ? 1 : 0;
  else return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spmv_plan_query: unknown key '%s'", key);
  return KKAMD_OK;
}

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
  KK_DISPATCH_TYPES(kk::spmv_typed, plan, A, trans, alpha, d_x, beta, d_y, st);
}

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
  // one contiguous column: the rank-1 path (sparse/src/KokkosSparse_spmv.hpp:203-217)
  if (nvec == 1 && x_stride0 == 1 && y_stride0 == 1) return kkamd_spmv(plan, A, mode, alpha, d_X, beta, d_Y, vector_type, stream);
  KK_DISPATCH_TYPES(kk::spmv_mv_typed, plan, A, trans, alpha, d_X, x_stride0, x_stride1, beta, d_Y, y_stride0,
                    y_stride1, nvec, st);
}

}  // extern "C"
