// kk_spmv_colslab.hip -- rank-1 y = beta*y + alpha*A*x for matrices whose x gather defeats the caches (uniform random columns:
// every nonzero pulls its own 128-byte line of x through the fabric; the CRS kernel then runs at 0.09 of its CRS-byte roofline).
//
// The plan keeps a second copy of the matrix in COLUMN-SLAB order: the column range is cut into slabs of 2^shift columns whose x
// segment (2 MB) fits an XCD's 4 MB L2 next to the streams; the entries are stably sorted by slab, so inside a slab they ascend by
// row (then by position in the row).  The kernel streams (row, column, value) triples in that order: all eight XCDs sweep the same
// slab at the same time, x comes out of L2, and every product goes to y through a global fp64/fp32 atomic add -- consecutive lanes
// hold consecutive rows, so a wave's atomics fall into a handful of lines; runs of one row inside a wave are folded first.
//   measured (tools/proto/colslab.py, one MI355X): uniform random 5e6 x 20: 1.76 -> 0.74 ms; R-MAT scale 22: 0.54 -> 0.43 ms.
// The matrix values may change between calls (the handle is bound to the matrix, not to its values; the reference's plug-ins read
// A.values through the pointer at every call, sparse/tpls/KokkosSparse_spmv_tpl_spec_decl.hpp:386-425), so a re-ordered copy of them must
// be kept current.  Knob "values_tracking" (it also governs the cached transpose of modes T / H, kk_spmv.hip):
//   0 (default) EXACT: the plan keeps a shadow copy of A.values in A's own order; every call streams both (2 x 8 bytes per nonzero),
//     compares them bit for bit tile by tile (4096 values) and moves the tiles that differ into the re-ordered copy (and the shadow);
//   1 NOTIFY: the caller says when the values changed (kkamd_spmv_plan_values_changed); the next call then copies all of them, calls in
//     between read nothing extra;
//   2 FINGERPRINTS: 128 bits per tile (two sums of products of the value's two 32-bit halves, each mixed with position-dependent
//     constants: not linear in the values), one stream of 8 bytes per nonzero, no shadow copy.  A changed tile whose fingerprint does
//     not change is missed -- for unstructured changes that takes a 2^-64-ish coincidence per sum, but it is a heuristic, not a proof;
//     callers who need certainty use 0 or 1.
// "colslab_const" = 1 promises constant values (no pass at all; = NOTIFY without notifications).
// Costs nnz * (8 + sizeof(value) + sizeof(offset)) bytes of plan memory.  The summation order differs from the CRS kernel's
// (slab by slab, atomics): results agree to rounding, not bit for bit, and vary from run to run in the last bits.
// No reference counterpart: KokkosSparse's native SpMV (sparse/impl/KokkosSparse_spmv_impl.hpp:104-160) and the rocSPARSE
// wrapper (sparse/tpls/KokkosSparse_spmv_tpl_spec_decl.hpp:383-470) run row-major CRS kernels on these matrices.
#include "kk_spmv_plan.h"

#ifdef KK_EMU
#define KK_CS_FADD(p, v) atomicAdd((p), (v))
#else
#define KK_CS_FADD(p, v) unsafeAtomicAdd((p), (v))   // global_atomic_add_f64 / _f32, no return
#endif

struct kkamd_cs_plan {
  int shift = 0, nslabs = 0, offset_type = 0, value_type = 0;
  int64_t nnz = 0, ntiles = 0;
  int32_t* d_row = nullptr;
  int32_t* d_col = nullptr;
  void* d_val = nullptr;                 // [nnz] values in slab order
  void* d_dst = nullptr;                 // [nnz] offset type: where entry i of A sits in the slab order
  unsigned long long* d_fp = nullptr;    // [2 * ntiles] fingerprints of A.values, tile by tile (values_tracking 2)
  void* d_shadow = nullptr;              // [nnz] A.values as the copy last saw them (values_tracking 0; allocated on first use)
  bool fp_valid = false, shadow_valid = false, shadow_failed = false, stale = false;
  size_t bytes = 0;
  // DETERMINISTIC form (round 5): no atomics.  Per-slab partial sums of every row, y_part[slab][row]; which slabs a row has entries in;
  // and, per chunk of the slab-order stream (a wave's kCsDetChunk entries), the pieces of the runs that a chunk boundary cuts.
  bool det = false;
  int64_t nrows = 0, nchunks = 0;
  unsigned long long* d_mask = nullptr;  // [nrows] bit s: the row has entries in slab s (nslabs <= 64)
  void* d_part = nullptr;                // [nslabs][nrows] vector type
  void* d_head = nullptr;                // [nchunks] vector type: the chunk's first run when it continues the previous chunk's last
  void* d_tail = nullptr;                // [nchunks] ... its last run when the next chunk continues it
  long long* d_tkey = nullptr;           // [nchunks] slab * nrows + row of that last run
  int* d_cflag = nullptr;                // [nchunks] bit 0: the head piece exists, bit 1: the tail piece exists, bit 2: the whole chunk is one run cut at both ends
  size_t part_bytes = 0;
};

namespace kk {

constexpr int kCsPer      = 16;                 // entries per work-item of a sort / fingerprint tile
constexpr int kCsTile     = kBlock * kCsPer;    // 4096
constexpr int kCsMaxSlabs = 256;
constexpr int kCsU        = 8;                  // entries per work-item of the SpMV kernel
constexpr int kDenseThreads = 1024;

// ------------------------------------------------------------------------------------------------
// stable counting sort by slab, pass 1: entries per (slab, tile), slab-major so that one prefix sum gives every (slab, tile) its start
__global__ __launch_bounds__(kBlock) void cs_hist_kernel(int64_t nnz, const int32_t* __restrict__ ent, int shift, int nslabs, int64_t ntiles,
                                                         int64_t* __restrict__ H) {
  __shared__ int s_h[kCsMaxSlabs];
  for (int k = threadIdx.x; k < nslabs; k += kBlock) s_h[k] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kCsTile;
  for (int u = 0; u < kCsPer; ++u) {
    const int64_t i = base + (int64_t)u * kBlock + threadIdx.x;
    if (i < nnz) atomicAdd(&s_h[ent[i] >> shift], 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < nslabs; k += kBlock) H[(int64_t)k * ntiles + blockIdx.x] = s_h[k];
}

// largest r in [lo, hi) with rm[r] <= i  (rm[lo] <= i < rm[hi] on entry)
template <class OffT> __device__ __forceinline__ int64_t cs_row_of(const OffT* __restrict__ rm, int64_t i, int64_t lo, int64_t hi) {
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)rm[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// pass 2: every wave takes a contiguous quarter of the tile in steps of 64 consecutive entries; inside a step the lanes of one slab
// are ranked by lane (match-any over the slab bits), the steps and waves by their per-slab cursors -- the order inside a slab is the
// CRS order, i.e. ascending rows
template <class OffT, class AT>
__global__ __launch_bounds__(kBlock) void cs_scatter_kernel(int64_t nrows, int64_t nnz, const OffT* __restrict__ rm, const int32_t* __restrict__ ent,
                                                            const AT* __restrict__ val, int shift, int nslabs, int nbits, int64_t ntiles,
                                                            const int64_t* __restrict__ H, int32_t* __restrict__ o_row, int32_t* __restrict__ o_col,
                                                            AT* __restrict__ o_val, OffT* __restrict__ dst) {
  __shared__ int s_cnt[kBlock / 64][kCsMaxSlabs];
  __shared__ int64_t s_start[kBlock / 64][kCsMaxSlabs];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int kQuarter = kCsTile / (kBlock / 64);
  const int64_t q0 = (int64_t)blockIdx.x * kCsTile + (int64_t)w * kQuarter;
  for (int k = threadIdx.x; k < (kBlock / 64) * kCsMaxSlabs; k += kBlock) (&s_cnt[0][0])[k] = 0;
  __syncthreads();
  for (int s = 0; s < kQuarter / 64; ++s) {
    const int64_t i = q0 + (int64_t)s * 64 + lane;
    if (i < nnz) atomicAdd(&s_cnt[w][ent[i] >> shift], 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < nslabs; k += kBlock) {
    int64_t b = H[(int64_t)k * ntiles + blockIdx.x];
    for (int ww = 0; ww < kBlock / 64; ++ww) { s_start[ww][k] = b; b += s_cnt[ww][k]; }
  }
  __syncthreads();
  int64_t r_prev = 0;                                           // rows only go up along the quarter
  for (int s = 0; s < kQuarter / 64; ++s) {
    const int64_t i_first = q0 + (int64_t)s * 64;
    if (i_first >= nnz) break;                                  // wave-uniform
    const int64_t i_last = i_first + 63 < nnz ? i_first + 63 : nnz - 1;
    const int64_t i = i_first + lane;
    const bool ok = i < nnz;
    const int32_t c = ok ? ent[i] : 0;
    const int key = c >> shift;
    unsigned long long mask = __ballot(ok);
    for (int b = 0; b < nbits; ++b) {
      const bool bit = (key >> b) & 1;
      const unsigned long long m = __ballot(ok && bit);
      mask &= bit ? m : ~m;
    }
    const int rank = __popcll(mask & ((1ull << lane) - 1ull)), total = __popcll(mask);
    int64_t pos = 0;
    if (ok) pos = s_start[w][key] + rank;
    KK_WAVE_SYNC();
    if (ok && rank == total - 1) s_start[w][key] += total;
    KK_WAVE_SYNC();
    const int64_t r_lo = cs_row_of<OffT>(rm, i_first, r_prev, nrows), r_hi = cs_row_of<OffT>(rm, i_last, r_lo, nrows);
    r_prev = r_hi;
    if (ok) {
      const int64_t row = cs_row_of<OffT>(rm, i, r_lo, r_hi + 1);
      o_row[pos] = (int32_t)row; o_col[pos] = c; o_val[pos] = val[i]; dst[i] = (OffT)pos;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fingerprints of A.values
__device__ __forceinline__ unsigned long long cs_mix(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned long long cs_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ unsigned long long cs_bits(float v) { return (unsigned long long)(unsigned)__float_as_int(v); }

// MODE 0: tiles whose fingerprint moved are copied into their places (o_val[dst[i]] = val[i]) again; 1: record the fingerprints only
// (the copy was just made from these values); 2: record them and copy every tile (the copy holds nothing yet); 3: EXACT -- tiles that
// differ from the shadow copy bit for bit are copied (into the shadow too); 4: copy every tile (and fill the shadow when there is one)
template <class OffT, class AT, int MODE>
__global__ __launch_bounds__(kBlock) void cs_check_kernel(int64_t nnz, const AT* __restrict__ val, const OffT* __restrict__ dst, AT* __restrict__ o_val,
                                                          unsigned long long* __restrict__ fp, AT* __restrict__ shadow) {
  __shared__ unsigned long long s_a[kBlock / 64], s_b[kBlock / 64];
  __shared__ int s_diff;
  const int64_t base = (int64_t)blockIdx.x * kCsTile;
  AT v[kCsPer];
  if (MODE >= 3) {
    if (MODE == 3) {
      if (threadIdx.x == 0) s_diff = 0;
      __syncthreads();
      bool diff = false;
      KK_UNROLL
      for (int u = 0; u < kCsPer; ++u) {
        const int64_t i = base + (int64_t)u * kBlock + threadIdx.x;
        v[u] = i < nnz ? val[i] : AT(0);
        const AT old = i < nnz ? shadow[i] : AT(0);
        diff |= cs_bits(v[u]) != cs_bits(old);
      }
      if (diff) s_diff = 1;
      __syncthreads();
      if (!s_diff) return;
    } else {
      KK_UNROLL
      for (int u = 0; u < kCsPer; ++u) { const int64_t i = base + (int64_t)u * kBlock + threadIdx.x; v[u] = i < nnz ? val[i] : AT(0); }
    }
    KK_UNROLL
    for (int u = 0; u < kCsPer; ++u) {
      const int64_t i = base + (int64_t)u * kBlock + threadIdx.x;
      if (i < nnz) { o_val[dst[i]] = v[u]; if (shadow) shadow[i] = v[u]; }
    }
    return;
  }
  unsigned long long a = 0, b = 0;
  KK_UNROLL
  for (int u = 0; u < kCsPer; ++u) {
    const int64_t i = base + (int64_t)u * kBlock + threadIdx.x;
    v[u] = i < nnz ? val[i] : AT(0);
    const unsigned long long bits = cs_bits(v[u]);
    // products of the value's halves, each first mixed with a constant of its position: not linear in the value bits (flipping the
    // sign of an even number of values, or scaling by two, moved the round-3 linear sums by 0 or by a multiple of a power of two)
    const unsigned lo = (unsigned)bits, hi = (unsigned)(bits >> 32);
    const unsigned k1 = ((unsigned)i * 0x9E3779B1u) ^ ((unsigned)((unsigned long long)i >> 32) * 0x85EBCA6Bu), k2 = (k1 << 16 | k1 >> 16) ^ 0xC2B2AE35u;
    a += (unsigned long long)(lo ^ k1) * (unsigned long long)((hi + k2) | 1u);
    b += (unsigned long long)(((lo << 13) | (lo >> 19)) + k2) * (unsigned long long)(((hi ^ (lo >> 7)) ^ k1) | 1u) + (bits ^ ((unsigned long long)k2 << 32 | k1));
  }
  a = group_sum(a, 64); b = group_sum(b, 64);
  if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long ta = 0, tb = 0;
    for (int w = 0; w < kBlock / 64; ++w) { ta += s_a[w]; tb += s_b[w]; }
    const bool diff = MODE != 0 || fp[2 * (int64_t)blockIdx.x] != ta || fp[2 * (int64_t)blockIdx.x + 1] != tb;
    if (diff) { fp[2 * (int64_t)blockIdx.x] = ta; fp[2 * (int64_t)blockIdx.x + 1] = tb; }
    s_diff = (MODE != 1 && diff) ? 1 : 0;
  }
  __syncthreads();
  if (s_diff) {
    KK_UNROLL
    for (int u = 0; u < kCsPer; ++u) {
      const int64_t i = base + (int64_t)u * kBlock + threadIdx.x;
      if (i < nnz) o_val[dst[i]] = v[u];
    }
  }
}

template <class OffT, class AT>
static int values_refresh_typed(int64_t nnz, const void* val, const void* dst, void* o_val, unsigned long long* fp, void* shadow, int mode, hipStream_t st) {
  const unsigned grid = (unsigned)ceil_div(nnz, (int64_t)kCsTile);
  const AT* v = (const AT*)val; const OffT* d = (const OffT*)dst; AT* o = (AT*)o_val; AT* sh = (AT*)shadow;
  if (mode == 0)      { KK_LAUNCH((cs_check_kernel<OffT, AT, 0>), grid, kBlock, 0, st, nnz, v, d, o, fp, sh); }
  else if (mode == 1) { KK_LAUNCH((cs_check_kernel<OffT, AT, 1>), grid, kBlock, 0, st, nnz, v, d, o, fp, sh); }
  else if (mode == 2) { KK_LAUNCH((cs_check_kernel<OffT, AT, 2>), grid, kBlock, 0, st, nnz, v, d, o, fp, sh); }
  else if (mode == 3) { KK_LAUNCH((cs_check_kernel<OffT, AT, 3>), grid, kBlock, 0, st, nnz, v, d, o, fp, sh); }
  else                { KK_LAUNCH((cs_check_kernel<OffT, AT, 4>), grid, kBlock, 0, st, nnz, v, d, o, fp, sh); }
  KK_LAUNCH_CHECK();
  return KKAMD_OK;
}
// A re-ordered copy of a matrix's values (o_val[dst[i]] = val[i]) kept current (see the head of this file); also what the cached
// transpose of modes T / H uses (kk_spmv.hip).  mode: the MODE of cs_check_kernel (0..2 want fp, 3 wants shadow, 4 takes it when given)
int64_t values_fp_tiles(int64_t nnz) { return ceil_div(nnz, (int64_t)kCsTile); }
int values_refresh(int offset_type, int value_type, int64_t nnz, const void* val, const void* dst, void* o_val, unsigned long long* fp, void* shadow, int mode, hipStream_t st) {
  if (nnz <= 0) return KKAMD_OK;
  const bool o64 = offset_type == KKAMD_I64;
  if (value_type == KKAMD_F64) return o64 ? values_refresh_typed<int64_t, double>(nnz, val, dst, o_val, fp, shadow, mode, st) : values_refresh_typed<int32_t, double>(nnz, val, dst, o_val, fp, shadow, mode, st);
  return o64 ? values_refresh_typed<int64_t, float>(nnz, val, dst, o_val, fp, shadow, mode, st) : values_refresh_typed<int32_t, float>(nnz, val, dst, o_val, fp, shadow, mode, st);
}
// One policy for both re-ordered copies.  tracking: the knob (0 exact, 1 notify, 2 fingerprints); promise: the caller's constant-values
// promise; the flags are the copy's state (the copy is current when this returns).
int values_track(int tracking, bool promise, int offset_type, int value_type, int64_t nnz, const void* val, const void* dst, void* o_val,
                 unsigned long long* fp, void** shadow, bool* fp_valid, bool* shadow_valid, bool* shadow_failed, bool* stale, hipStream_t st) {
  const size_t vb = value_type == KKAMD_F64 ? 8 : 4;
  int rc = KKAMD_OK;
  if (promise || tracking == 1) {
    if (*stale) { rc = values_refresh(offset_type, value_type, nnz, val, dst, o_val, fp, nullptr, 4, st); *fp_valid = false; *shadow_valid = false; }
  } else if (tracking == 2) {
    rc = values_refresh(offset_type, value_type, nnz, val, dst, o_val, fp, nullptr, (*fp_valid && !*stale) ? 0 : 2, st);
    *fp_valid = true; *shadow_valid = false;
  } else {
    if (!*shadow && !*shadow_failed && hipMalloc(shadow, vb * (size_t)nnz) != hipSuccess) { (void)hipGetLastError(); *shadow = nullptr; *shadow_failed = true; }
    if (*shadow) { rc = values_refresh(offset_type, value_type, nnz, val, dst, o_val, fp, *shadow, (*shadow_valid && !*stale) ? 3 : 4, st); *shadow_valid = true; }
    else rc = values_refresh(offset_type, value_type, nnz, val, dst, o_val, fp, nullptr, 4, st);      // no memory for the shadow: copy everything, every call
    *fp_valid = false;
  }
  *stale = false;
  return rc;
}

// ------------------------------------------------------------------------------------------------
// y += alpha * A * x over the slab order
template <class AT, class YT>
__global__ __launch_bounds__(kBlock) void cs_spmv_kernel(int64_t nnz, const int32_t* __restrict__ row, const int32_t* __restrict__ col,
                                                         const AT* __restrict__ val, int shift, const YT* __restrict__ x, YT* __restrict__ y, YT alpha) {
  const int64_t base = (int64_t)blockIdx.x * (kBlock * kCsU);
  const int lane = threadIdx.x & 63;
  int32_t r[kCsU], c[kCsU];
  AT v[kCsU];
  KK_UNROLL
  for (int u = 0; u < kCsU; ++u) {
    const int64_t i = base + (int64_t)u * kBlock + threadIdx.x;
    const bool ok = i < nnz;
    r[u] = ok ? row[i] : -1; c[u] = ok ? col[i] : 0; v[u] = ok ? val[i] : AT(0);
  }
  YT p[kCsU];
  KK_UNROLL
  for (int u = 0; u < kCsU; ++u) p[u] = (YT)v[u] * x[c[u]];
  KK_UNROLL
  for (int u = 0; u < kCsU; ++u) {
    // runs of one (slab, row) inside the wave fold into their first lane
    const int slab = r[u] < 0 ? -1 : (c[u] >> shift);
    const int rp = __shfl_up(r[u], 1, 64), sp = __shfl_up(slab, 1, 64);
    const bool head = lane == 0 || rp != r[u] || sp != slab;
    const unsigned long long heads = __ballot(head);
    YT s = p[u];
    if (heads != ~0ull) {                                       // wave-uniform
      const unsigned long long rest = lane == 63 ? 0ull : heads >> (lane + 1);
      const int end = rest ? lane + __ffsll(rest) : 64;         // first lane of the next run
      for (int o = 1; o < 64; o <<= 1) {
        const YT t = __shfl_down(s, o, 64);
        if (lane + o < end) s += t;
      }
    }
    if (head && r[u] >= 0) KK_CS_FADD(&y[r[u]], alpha * s);
  }
}

// ------------------------------------------------------------------------------------------------
// DETERMINISTIC slab SpMV (no atomics, no timing, the same bits on every run; round-4 review item 7).  The reference's native mode-N
// kernel is deterministic (sparse/impl/KokkosSparse_spmv_impl.hpp:134-165); the atomic form above is not, which is why it was opt-in.
//   pass 1 (cs_det_kernel): a wave takes kCsDetChunk consecutive entries of the slab-order stream, 64 at a time.  Inside the 64 the runs of
//     one (slab, row) are folded by a fixed shuffle tree into their first lane; the last run of every 64 is carried (wave-uniform
//     registers) into the next 64; a run that ends inside the chunk is STORED to y_part[slab][row] -- every (slab, row) has exactly one
//     writer.  The chunk's first run when it continues the previous chunk's, and its last run when the next chunk continues it, go to
//     the chunk's head / tail slot;
//   pass 2 (cs_det_fix_kernel): the chunk in which a cut run BEGINS adds its tail piece and the head pieces of the following chunks, in
//     chunk order, and stores the run;
//   pass 3 (cs_det_reduce_kernel): y[r] = beta y[r] + alpha * (sum over the slabs the row has entries in, ascending): the slab mask of
//     the row says which partial sums exist (nothing is zero-filled per call).
constexpr int kCsDetU = 8;                          // rounds of 64 entries per wave
constexpr int kCsDetChunk = 64 * kCsDetU;
__global__ __launch_bounds__(kBlock) void cs_mask_kernel(int64_t nnz, const int32_t* __restrict__ row, const int32_t* __restrict__ col, int shift,
                                                         unsigned long long* __restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nnz) return;
  const int slab = col[i] >> shift;
  if (i == 0 || row[i - 1] != row[i] || (col[i - 1] >> shift) != slab) atomicOr(&mask[row[i]], 1ull << slab);       // the first entry of a run
}
template <class AT, class YT>
__global__ __launch_bounds__(kBlock) void cs_det_kernel(int64_t nnz, int64_t nrows, const int32_t* __restrict__ row, const int32_t* __restrict__ col,
                                                        const AT* __restrict__ val, int shift, const YT* __restrict__ x, YT* __restrict__ part,
                                                        YT* __restrict__ head, YT* __restrict__ tail, long long* __restrict__ tkey, int* __restrict__ cflag) {
  const int lane = threadIdx.x & 63;
  const int64_t chunk = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t i0 = chunk * kCsDetChunk;
  if (i0 >= nnz) return;                                          // the whole wave leaves
  const int64_t i1 = i0 + kCsDetChunk < nnz ? i0 + kCsDetChunk : nnz;
  auto key_at = [&](int64_t i) { return (long long)(col[i] >> shift) * nrows + row[i]; };
  const long long prev_key = i0 > 0 ? key_at(i0 - 1) : -1, next_key = i1 < nnz ? key_at(i1) : -1;
  // the OPEN run: the last run of the rounds so far, which the next round may continue (wave-uniform registers)
  bool open = false, open_fp = false;                             // open_fp: it began at the chunk's first entry and continues the previous chunk's last run
  long long open_key = -1;
  YT open_sum = YT(0);
  int flags = 0;
  // all rounds' triples, then all their x gathers, are requested before the first round is folded (unconditional loads at clamped
  // indices: a round at a time was a chain of two memory round trips per round)
  int32_t cu[kCsDetU], ru[kCsDetU];
  AT vu[kCsDetU];
  YT xu[kCsDetU];
  KK_UNROLL
  for (int u = 0; u < kCsDetU; ++u) {
    int64_t ic = i0 + (int64_t)u * 64 + lane;
    ic = ic < i1 ? ic : i1 - 1;
    cu[u] = col[ic]; ru[u] = row[ic]; vu[u] = val[ic];
  }
  KK_UNROLL
  for (int u = 0; u < kCsDetU; ++u) xu[u] = x[cu[u]];
  KK_UNROLL
  for (int u = 0; u < kCsDetU; ++u) {
    const int64_t ib = i0 + (int64_t)u * 64;
    if (ib >= i1) break;                                          // uniform
    const int64_t i = ib + lane;
    const bool ok = i < i1;
    const int32_t c = cu[u], r = ru[u];
    const long long key = ok ? (long long)(c >> shift) * nrows + r : -3;
    YT s = ok ? (YT)vu[u] * xu[u] : YT(0);
    // runs of one key inside the 64 fold into their first lane (a fixed tree: the same order on every run)
    const long long kp = __shfl_up(key, 1, 64);
    const bool hd = lane == 0 || kp != key;
    const unsigned long long heads = __ballot(hd);
    const unsigned long long rest = lane == 63 ? 0ull : heads >> (lane + 1);
    const int end = rest ? lane + __ffsll(rest) : 64;             // first lane of the next run
    for (int o = 1; o < 64; o <<= 1) {
      const YT t = __shfl_down(s, o, 64);
      if (lane + o < end) s += t;
    }
    const int n_ok = (int)(i1 - ib < 64 ? i1 - ib : 64);
    const int last_head = 63 - __clzll((long long)(heads & (n_ok == 64 ? ~0ull : ((1ull << n_ok) - 1ull))));      // head lane of the round's last run
    const long long first_key = __shfl(key, 0, 64);
    bool r1_fp;                                                   // the round's first run continues the previous chunk
    if (open && first_key == open_key) { if (lane == 0) s = open_sum + s; r1_fp = open_fp; }          // the open run goes on (the earlier piece first)
    else {
      if (open) {                                                 // it ended with the previous round
        if (lane == 0) { if (open_fp) head[chunk] = open_sum; else part[open_key] = open_sum; }
        if (open_fp) flags |= 1;
      }
      r1_fp = u == 0 && first_key == prev_key;
    }
    // every run but the round's last ends inside the round: its head lane stores it -- the one writer of that (slab, row)
    if (hd && ok && lane != last_head) { if (lane == 0 && r1_fp) head[chunk] = s; else part[key] = s; }
    if (last_head != 0 && r1_fp) flags |= 1;
    open = true;
    open_key = __shfl(key, last_head, 64); open_sum = __shfl(s, last_head, 64);
    open_fp = last_head == 0 ? r1_fp : false;
  }
  if (open_key == next_key) {                                     // the next chunk continues the last run
    if (open_fp) { if (lane == 0) head[chunk] = open_sum; flags |= 1 | 4; }                     // ... and the previous chunk began it: a whole-chunk piece
    else { if (lane == 0) { tail[chunk] = open_sum; tkey[chunk] = open_key; } flags |= 2; }     // the run BEGINS in this chunk: pass 2 sums it here
  } else {
    if (lane == 0) { if (open_fp) head[chunk] = open_sum; else part[open_key] = open_sum; }
    if (open_fp) flags |= 1;
  }
  if (lane == 0) cflag[chunk] = flags;
}
template <class YT>
__global__ __launch_bounds__(kBlock) void cs_det_fix_kernel(int64_t nchunks, const YT* __restrict__ head, const YT* __restrict__ tail, const long long* __restrict__ tkey,
                                                            const int* __restrict__ cflag, YT* __restrict__ part) {
  const int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (w >= nchunks || !(cflag[w] & 2)) return;                    // only a chunk in which a cut run BEGINS does the sum
  YT sum = tail[w];
  for (int64_t c = w + 1; c < nchunks; ++c) {                     // its pieces: the head slots of the following chunks, while they are whole
    sum += head[c];
    if (!(cflag[c] & 4)) break;
  }
  part[tkey[w]] = sum;
}
template <class YT>
__global__ __launch_bounds__(kBlock) void cs_det_reduce_kernel(int64_t nrows, int nslabs, const unsigned long long* __restrict__ mask, const YT* __restrict__ part,
                                                               YT* __restrict__ y, YT alpha, YT beta) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= nrows) return;
  const unsigned long long m = mask[r];
  YT sum = YT(0);
  for (int s = 0; s < nslabs; ++s) if ((m >> s) & 1ull) sum += part[(int64_t)s * nrows + r];
  y[r] = (beta == YT(0)) ? alpha * sum : beta * y[r] + alpha * sum;
}

// ------------------------------------------------------------------------------------------------
// Is the x gather of the CRS kernel cache-defeating?  A RULE, not a timing (the deterministic form is chosen at the first call without
// running anything twice): eight windows of 32768 consecutive nonzeros spread over the matrix -- about what one XCD's 4 MB L2 holds in
// 128-byte lines of x while it streams its share of the matrix -- and in each the number of DISTINCT lines of x its columns name (a
// 512 K-bit set in LDS; line ids beyond that are hashed, which can only undercount).  distinct / nonzeros near 1 means every nonzero
// pulls its own line through the fabric (uniform random columns: 0.95); hub columns, bands and stencils repeat lines (R-MAT scale 22: 0.6).
constexpr int kCsWin = 32768, kCsWins = 8, kCsSetWords = 16384;      // 16384 x 32 bits
__global__ __launch_bounds__(kDenseThreads) void cs_distinct_lines_kernel(int64_t nnz, const int32_t* __restrict__ ent, int line_shift, unsigned long long* __restrict__ out /*[2]*/) {
  __shared__ unsigned s_set[kCsSetWords];
  __shared__ unsigned s_new;
  for (int i = threadIdx.x; i < kCsSetWords; i += kDenseThreads) s_set[i] = 0u;
  if (threadIdx.x == 0) s_new = 0u;
  __syncthreads();
  const int64_t nwin = gridDim.x;
  const int64_t span = nnz > (int64_t)kCsWin ? nnz - kCsWin : 0;
  const int64_t b0 = nwin > 1 ? (span / (nwin - 1)) * blockIdx.x : 0;
  const int64_t b1 = b0 + kCsWin < nnz ? b0 + kCsWin : nnz;
  unsigned mine = 0;
  for (int64_t i = b0 + threadIdx.x; i < b1; i += kDenseThreads) {
    const unsigned line = (unsigned)ent[i] >> line_shift;
    const unsigned h = line < (unsigned)(kCsSetWords * 32) ? line : (line * 2654435761u) >> 13;          // (hashed beyond the set's size)
    const unsigned bit = 1u << (h & 31u);
    const unsigned old = atomicOr(&s_set[(h >> 5) & (kCsSetWords - 1)], bit);
    mine += (old & bit) ? 0u : 1u;
  }
  atomicAdd(&s_new, mine);
  __syncthreads();
  if (threadIdx.x == 0) { atomicAdd(&out[0], (unsigned long long)s_new); atomicAdd(&out[1], (unsigned long long)(b1 - b0)); }
}
// *ratio = distinct lines of x per nonzero over the sampled windows (0 when nothing could be measured)
int cs_gather_ratio(const kkamd_crs_t* A, int x_elem, hipStream_t st, double* ratio) {
  *ratio = 0.0;
  if (A->nnz <= 0) return KKAMD_OK;
  DevBuf cnt;
  if (cnt.alloc(2 * sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); return KKAMD_OK; }
  unsigned long long* d_c = cnt.as<unsigned long long>();
  KK_HIP(hipMemsetAsync(d_c, 0, 2 * sizeof(unsigned long long), st));
  const int nwin = A->nnz >= (int64_t)kCsWin * kCsWins ? kCsWins : 1;
  KK_LAUNCH(cs_distinct_lines_kernel, (unsigned)nwin, kDenseThreads, 0, st, A->nnz, (const int32_t*)A->d_entries, x_elem == 8 ? 4 : 5, d_c);
  KK_LAUNCH_CHECK();
  unsigned long long h[2] = {0, 0};
  KK_HIP(hipMemcpyAsync(h, d_c, sizeof h, hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  if (h[1]) *ratio = (double)h[0] / (double)h[1];
  return KKAMD_OK;
}

// ------------------------------------------------------------------------------------------------
void cs_plan_destroy(kkamd_cs_plan* cs) {
  if (!cs) return;
  void* bufs[] = {cs->d_row, cs->d_col, cs->d_val, cs->d_dst, cs->d_fp, cs->d_shadow, cs->d_mask, cs->d_part, cs->d_head, cs->d_tail, cs->d_tkey, cs->d_cflag};
  for (void* b : bufs) if (b) (void)hipFree(b);
  delete cs;
}

void cs_mark_stale(kkamd_cs_plan* cs) { if (cs) cs->stale = true; }
void cs_reset_tracking(kkamd_cs_plan* cs) { if (cs) { cs->stale = true; cs->fp_valid = false; cs->shadow_valid = false; } }

int64_t cs_plan_query(const kkamd_cs_plan* cs, int what) {
  if (!cs) return 0;
  switch (what) {
    case 0: return cs->nslabs;
    case 1: return cs->shift;
    case 2: return (int64_t)cs->bytes;
    case 3: return cs->det ? 1 : 0;
    default: return 0;
  }
}

template <class OffT, class AT>
static int cs_build_typed(kkamd_cs_plan** out, const kkamd_crs_t* A, int shift, int x_elem, bool det, hipStream_t st) {
  *out = nullptr;
  const int64_t nnz = A->nnz;
  kkamd_cs_plan* cs = new (std::nothrow) kkamd_cs_plan();
  if (!cs) return KKAMD_OK;
  cs->shift = shift; cs->nslabs = (int)ceil_div(A->num_cols, (int64_t)1 << shift);
  cs->nnz = nnz; cs->ntiles = ceil_div(nnz, (int64_t)kCsTile);
  cs->offset_type = A->offset_type; cs->value_type = A->value_type;
  auto give_up = [&]() { (void)hipGetLastError(); cs_plan_destroy(cs); return KKAMD_OK; };      // the copy is an optimisation: no memory, no copy
  const size_t hn = (size_t)cs->nslabs * (size_t)cs->ntiles + 1;
  DevBuf hist;
  if (hipMalloc((void**)&cs->d_row, sizeof(int32_t) * (size_t)nnz) != hipSuccess || hipMalloc((void**)&cs->d_col, sizeof(int32_t) * (size_t)nnz) != hipSuccess ||
      hipMalloc(&cs->d_val, sizeof(AT) * (size_t)nnz) != hipSuccess || hipMalloc(&cs->d_dst, sizeof(OffT) * (size_t)nnz) != hipSuccess ||
      hipMalloc((void**)&cs->d_fp, 16 * (size_t)values_fp_tiles(nnz)) != hipSuccess || hist.alloc(sizeof(int64_t) * hn) != hipSuccess)
    return give_up();
  cs->bytes = (size_t)nnz * (8 + sizeof(AT) + sizeof(OffT)) + 16 * (size_t)cs->ntiles;
  int64_t* H = hist.as<int64_t>();
  int nbits = 0;
  while ((1 << nbits) < cs->nslabs) ++nbits;
  if (hipMemsetAsync(H + (hn - 1), 0, sizeof(int64_t), st) != hipSuccess) return give_up();
  KK_LAUNCH(cs_hist_kernel, (unsigned)cs->ntiles, kBlock, 0, st, nnz, (const int32_t*)A->d_entries, shift, cs->nslabs, cs->ntiles, H);
  int rc = kkamd_exclusive_scan(H, (int64_t)hn, KKAMD_I64, reinterpret_cast<kkamd_stream_t>(st));
  if (rc) { cs_plan_destroy(cs); return rc; }
  int32_t* o_row = cs->d_row; int32_t* o_col = cs->d_col; AT* o_val = (AT*)cs->d_val; OffT* dst = (OffT*)cs->d_dst;
  unsigned long long* fp = cs->d_fp;
  const int nslabs = cs->nslabs; const int64_t ntiles = cs->ntiles;
  KK_LAUNCH((cs_scatter_kernel<OffT, AT>), (unsigned)ntiles, kBlock, 0, st, A->num_rows, nnz, (const OffT*)A->d_row_map, (const int32_t*)A->d_entries,
            (const AT*)A->d_values, shift, nslabs, nbits, ntiles, (const int64_t*)H, o_row, o_col, o_val, dst);
  // the scatter kernel copied the values: the copy is current, nothing is recorded yet (the first tracked call fills shadow / fingerprints)
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return give_up();
  if (det) {
    // the deterministic form's state: slab mask per row, per-slab partial sums, the cut runs' pieces per chunk
    cs->det = true; cs->nrows = A->num_rows; cs->nchunks = ceil_div(nnz, (int64_t)kCsDetChunk);
    cs->part_bytes = (size_t)cs->nslabs * (size_t)A->num_rows * (size_t)x_elem;
    if (cs->nslabs > 64 || hipMalloc((void**)&cs->d_mask, sizeof(unsigned long long) * (size_t)A->num_rows) != hipSuccess ||
        hipMalloc(&cs->d_part, cs->part_bytes) != hipSuccess || hipMalloc(&cs->d_head, (size_t)x_elem * (size_t)cs->nchunks) != hipSuccess ||
        hipMalloc(&cs->d_tail, (size_t)x_elem * (size_t)cs->nchunks) != hipSuccess || hipMalloc((void**)&cs->d_tkey, sizeof(long long) * (size_t)cs->nchunks) != hipSuccess ||
        hipMalloc((void**)&cs->d_cflag, sizeof(int) * (size_t)cs->nchunks) != hipSuccess || hipMemsetAsync(cs->d_mask, 0, sizeof(unsigned long long) * (size_t)A->num_rows, st) != hipSuccess)
      return give_up();
    unsigned long long* d_mask = cs->d_mask; const int32_t* c_row = cs->d_row; const int32_t* c_col = cs->d_col;
    KK_LAUNCH(cs_mask_kernel, (unsigned)ceil_div(nnz, (int64_t)kBlock), kBlock, 0, st, nnz, c_row, c_col, shift, d_mask);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return give_up();
    cs->bytes += sizeof(unsigned long long) * (size_t)A->num_rows + cs->part_bytes + (size_t)cs->nchunks * (2 * (size_t)x_elem + 12);
  }
  *out = cs;
  return KKAMD_OK;
}

// log2 of the columns per slab the copy is built with (also what the selection rule of kk_spmv.hip prices: one place for both)
int cs_pick_shift(const kkamd_crs_t* A, int x_elem, int shift_knob, bool det) {
  int shift = shift_knob;
  if (shift <= 0) {
    // 2 MB of x per slab; up to 8 MB when the rows are so short that a slab would hold less than half an entry per row (the
    // atomics of a wave then scatter over y: 2e7 rows x 8 at 2 / 4 / 8 MB slabs 4.97 / 3.72 / 2.85 ms, 1e7 x 12: 2.14 / 1.48 / 1.76)
    shift = x_elem == 8 ? 18 : 19;
    const double per_row = (double)A->nnz / (double)A->num_rows;
    for (int widen = 0; widen < 2 && per_row * (double)((int64_t)1 << shift) < 0.5 * (double)A->num_cols; ++widen) ++shift;
  }
  while (ceil_div(A->num_cols, (int64_t)1 << shift) > (det ? 64 : kCsMaxSlabs)) ++shift;      // (the deterministic form keeps a 64-bit slab mask per row)
  return shift;
}
// Builds the slab-order copy (nullptr in *out when HBM cannot hold it).  x_elem: bytes per x element (sizes the slabs).
int cs_build(kkamd_cs_plan** out, const kkamd_crs_t* A, int x_elem, int shift_knob, hipStream_t st, bool det) {
  *out = nullptr;
  if (A->nnz <= 0 || A->num_rows <= 0 || A->num_cols <= 0) return KKAMD_OK;
  const int shift = cs_pick_shift(A, x_elem, shift_knob, det);
  const size_t off_b = A->offset_type == KKAMD_I64 ? 8 : 4, val_b = A->value_type == KKAMD_F64 ? 8 : 4;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return KKAMD_OK; }
  // (+ val_b: the shadow copy of A.values that exact value tracking, the default, compares against at every call)
  const double need = (double)A->nnz * (8.0 + 2.0 * val_b + off_b) + 8.0 * (double)ceil_div(A->num_cols, (int64_t)1 << shift) * (double)ceil_div(A->nnz, (int64_t)kCsTile);
  const double need_det = det ? (double)ceil_div(A->num_cols, (int64_t)1 << shift) * (double)A->num_rows * x_elem + 8.0 * (double)A->num_rows : 0.0;
  if (need + need_det > (double)free_b / 4.0) return KKAMD_OK;
  const bool o64 = A->offset_type == KKAMD_I64;
  if (A->value_type == KKAMD_F64) return o64 ? cs_build_typed<int64_t, double>(out, A, shift, x_elem, det, st) : cs_build_typed<int32_t, double>(out, A, shift, x_elem, det, st);
  return o64 ? cs_build_typed<int64_t, float>(out, A, shift, x_elem, det, st) : cs_build_typed<int32_t, float>(out, A, shift, x_elem, det, st);
}

template <class OffT, class AT, class YT>
static int cs_apply_typed(kkamd_cs_plan* cs, const kkamd_crs_t* A, const YT* x, YT* y, YT alpha, YT beta, int tracking, hipStream_t st) {
  const int64_t nnz = cs->nnz;
  AT* o_val = (AT*)cs->d_val;
  // tracking: the "values_tracking" knob, or -1 = the caller promised constant values
  int rc = values_track(tracking < 0 ? 1 : tracking, tracking < 0, A->offset_type, A->value_type, nnz, A->d_values, cs->d_dst, o_val, cs->d_fp, &cs->d_shadow,
                        &cs->fp_valid, &cs->shadow_valid, &cs->shadow_failed, &cs->stale, st);
  if (rc) return rc;
  const int32_t* row = cs->d_row; const int32_t* col = cs->d_col; const int shift = cs->shift;
  if (cs->det) {
    if (cs->part_bytes != (size_t)cs->nslabs * (size_t)A->num_rows * sizeof(YT)) {
      // the handle's first call had another vector type (the CRS kernels take either on one handle, so must this form): the per-slab
      // partial sums and the chunk heads / tails are sized again for this one
      KK_HIP(hipStreamSynchronize(st));
      if (cs->d_part) (void)hipFree(cs->d_part);
      if (cs->d_head) (void)hipFree(cs->d_head);
      if (cs->d_tail) (void)hipFree(cs->d_tail);
      cs->d_part = cs->d_head = cs->d_tail = nullptr;
      const size_t want = (size_t)cs->nslabs * (size_t)A->num_rows * sizeof(YT);
      if (hipMalloc(&cs->d_part, want) != hipSuccess || hipMalloc(&cs->d_head, sizeof(YT) * (size_t)cs->nchunks) != hipSuccess ||
          hipMalloc(&cs->d_tail, sizeof(YT) * (size_t)cs->nchunks) != hipSuccess) {
        (void)hipGetLastError(); cs->part_bytes = 0;
        return fail(KKAMD_ERR_ALLOC, "kkamd_spmv: no memory for the column-slab form's partial sums in this vector type");
      }
      cs->bytes += want; cs->bytes -= cs->part_bytes;
      cs->part_bytes = want;
    }
    YT* part = (YT*)cs->d_part; YT* head = (YT*)cs->d_head; YT* tail = (YT*)cs->d_tail; long long* tkey = cs->d_tkey; int* cflag = cs->d_cflag;
    const int64_t nchunks = cs->nchunks, nrows = A->num_rows; const int nslabs = cs->nslabs; const unsigned long long* mask = cs->d_mask;
    KK_LAUNCH((cs_det_kernel<AT, YT>), (unsigned)ceil_div(nchunks, (int64_t)(kBlock / 64)), kBlock, 0, st, nnz, nrows, row, col, (const AT*)o_val, shift, x, part, head, tail, tkey, cflag);
    KK_LAUNCH_CHECK();
    KK_LAUNCH((cs_det_fix_kernel<YT>), (unsigned)ceil_div(nchunks, (int64_t)kBlock), kBlock, 0, st, nchunks, (const YT*)head, (const YT*)tail, (const long long*)tkey, (const int*)cflag, part);
    KK_LAUNCH_CHECK();
    KK_LAUNCH((cs_det_reduce_kernel<YT>), (unsigned)ceil_div(nrows, (int64_t)kBlock), kBlock, 0, st, nrows, nslabs, mask, (const YT*)part, y, alpha, beta);
    KK_LAUNCH_CHECK();
    return KKAMD_OK;
  }
  rc = launch_scale<YT>(y, A->num_rows, 1, 1, 0, beta, st);
  if (rc) return rc;
  KK_LAUNCH((cs_spmv_kernel<AT, YT>), (unsigned)ceil_div(nnz, (int64_t)kBlock * kCsU), kBlock, 0, st, nnz, row, col, (const AT*)o_val, shift, x, y, alpha);
  KK_LAUNCH_CHECK();
  return KKAMD_OK;
}

int cs_apply(kkamd_cs_plan* cs, const kkamd_crs_t* A, int vector_type, const void* x, void* y, double alpha, double beta, int tracking, hipStream_t st) {
  if (cs->nnz != A->nnz || cs->offset_type != A->offset_type || cs->value_type != A->value_type)
    return fail(KKAMD_ERR_STATE, "kkamd_spmv: the column-slab copy belongs to another matrix");
  const bool o64 = A->offset_type == KKAMD_I64;
  if (A->value_type == KKAMD_F64 && vector_type == KKAMD_F64)
    return o64 ? cs_apply_typed<int64_t, double, double>(cs, A, (const double*)x, (double*)y, alpha, beta, tracking, st)
               : cs_apply_typed<int32_t, double, double>(cs, A, (const double*)x, (double*)y, alpha, beta, tracking, st);
  if (A->value_type == KKAMD_F32 && vector_type == KKAMD_F32)
    return o64 ? cs_apply_typed<int64_t, float, float>(cs, A, (const float*)x, (float*)y, (float)alpha, (float)beta, tracking, st)
               : cs_apply_typed<int32_t, float, float>(cs, A, (const float*)x, (float*)y, (float)alpha, (float)beta, tracking, st);
  if (A->value_type == KKAMD_F32 && vector_type == KKAMD_F64)
    return o64 ? cs_apply_typed<int64_t, float, double>(cs, A, (const double*)x, (double*)y, alpha, beta, tracking, st)
               : cs_apply_typed<int32_t, float, double>(cs, A, (const double*)x, (double*)y, alpha, beta, tracking, st);
  return fail(KKAMD_ERR_UNSUPPORTED, "kkamd_spmv: unsupported (value,vector) type pair (%d,%d)", A->value_type, vector_type);
}

}  // namespace kk
