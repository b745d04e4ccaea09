// kk_spmv_plan.h -- the SpMV plan (the analogue of SPMVHandleImpl::tpl_rank1 / tpl_rank2,
// sparse/src/KokkosSparse_spmv_handle.hpp:241-242) and its tuning knobs, shared by the rank-1 (kk_spmv.hip),
// rank-2 (kk_spmv_mv.hip) and multi-GPU (kk_dist.hip) translation units.
#pragma once
#include "kk_common.h"
#include <string>
#include <utility>
#include <vector>

// Measurement build (-DKK_ABLATE, tools/ only): parts of the kernel can be switched off through an extra kernel argument.
// In the product build the argument does not exist and every KK_ABL(bit) folds to false.
#ifdef KK_ABLATE
#define KK_ABL_PARAM , int ablate
#define KK_ABL_ARG(p) , (p)->tune.ablate
#define KK_ABL(bit) ((ablate & (bit)) != 0)
#define KK_LDS_PAD(p) ((size_t)(p)->tune.lds_pad_kb * 1024)
#else
#define KK_ABL_PARAM
#define KK_ABL_ARG(p)
#define KK_ABL(bit) false
#define KK_LDS_PAD(p) ((size_t)0)
#endif

namespace kk {

struct SpmvTuning {
  int kernel         = 0;  // 0 auto, 1 vector (no analysis), 2 stream
  int lanes_per_row  = 0;  // vector kernel, 0 = auto
  int nnz_per_thread = 0;  // stream kernel: 4 (fp64 only), 8 or 16; 0 = by size
  int xcd_remap      = 16; // tile order of the nnz-split kernel: 0 dispatch order (tile b on XCD b % 8), 1 XCD-contiguous (3-8 % slower than 0),
                           // G = 2^k >= 2 grouped (G consecutive tiles per XCD inside blocks of 8G tiles; 16: 1.37 -> 1.29 ms on C2, 7-pt 400^3 -7 %)
  int nontemporal    = 0;  // measured: no consistent gain from nt loads on the value/column streams
  int mv_kernel      = 0;  // rank-2: 0 auto, 1 generic strided kernel, 2 wave-private row-major kernel, 3 LDS-staged X tiles (needs an analysed handle)
  int stream_variant = 1;  // 1 default; 6 = attempt the window codes whatever the matrix size (tests)
  int mv_remap       = 16; // rank-2 wave-private kernel: 0 dispatch order, 1 XCD-contiguous, 2^k grouped (16: 5.60 -> 4.76 ms on C3)
  int mv_order       = 2;  // rank-2 LDS-staged kernel, tile order: 0 dispatch, 1 XCD-contiguous, 2 strips from the detected grid strides (falls back to 1)
  int mv_strip_min_kb = 3000;  // ... strips engage when three periods' worth of X rows exceed this (an XCD's L2 holds 4 MB)
  int mv_strip_l2_kb  = 2500;  // ... and are sized so that three periods' worth of a strip's X rows stay below this
  int mv_nt          = 0;  // rank-2 gather kernel: nontemporal loads of the matrix streams (entries, values)
  int mv_glds        = 1;  // rank-2 LDS-staged kernel: X window through global_load_lds (1) or through registers (0)
  int mv_long_T      = 0;  // rank-2 gather kernel: rows above this many entries get a workgroup each (0 = automatic: 4 x the average row, at least 64)
  int mv4_min_nvec   = 4;  // narrowest multivector the plane-marching kernel takes (a block of fewer than 16 columns runs its partial-block form)
  int mv4_xcol       = 1;  // rank-2 plane-marching kernel, column-major X: 1 = pieces dealt out column-wise + swizzled slab rows (whole cache lines per load), 0 = as for general strides
  int mv4_2d         = 1;  // rank-2 plane-marching kernel on 2-D lattices (lines grouped into planes): 1 on, 0 off
  int mv4_wg_per_cu  = 8;  // rank-2 plane-marching kernel: workgroups per CU the k-chunking aims for (one is resident at a time)
  int mv5            = 1;  // rank-2 matrix-core kernel (kk_spmv_mvblk.hip: 16-row tiles whose rows share columns, v_mfma_f64_16x16x4): 1 = when the
                           // analysis finds the tiles dense enough, 2 = whenever a tile can be described (tests), 0 = never
  int mv5_min_fill_pct = 25;  // ... a tile takes the matrix core when at least this share of its 16 x 4 operand slots holds an entry
  int mv5_max_other_pct = 50; // ... and the kernel is used when at most this share of the rows is left to the gather rows
  int mv6            = 1;  // rank-2 nonzero-split kernel (kk_spmv_mvnnz.hip: 128-entry chunks per 16-lane group, row index per nonzero in the plan, cut
                           // rows finished from carries): 1 = on matrices with long rows ("mv6_min_long_pct"), 2 = whenever the gather kernel would
                           // run, 0 = never
  int mv6_min_long_pct = 10;  // ... at least this share of the nonzeros sits in rows above the long-row threshold (4 x the average, at least 64)
  int march          = 0;  // rank 1 on the plane-marching analysis (lattice stencils, fp64 vectors): 0 off, 1 on
  int march_planes   = 20; // ... planes a workgroup marches (its k-chunk)
  int explicit_transpose = 1;   // modes T/H with an analysed handle: 1 = cache A^T in the plan (when it fits an eighth of free HBM), move the
                                // values that changed since the last call into it and run the N kernel on it; 2 = same, the caller promises
                                // constant values (no comparison); 0 = the reference's atomic scatter
  int explicit_transpose_min_knnz = 1000;   // ... from this many thousand nnz
  int transient_min_knnz = 10000;  // handle-less / FAST_SETUP calls analyse on the fly from this many thousand nnz (0 = never)
  int window_codes = 1;            // analysed handles: 16-bit window codes + LDS-staged x, tile by tile (2 = codes without staged x, 0 = never)
  int window_codes_min_knnz = 1000;  // ... from this many thousand nnz
  int window_codes_min_pct = 25;   // ... when at least this share of the tiles can use them (the others read entries, per tile)
  int pattern_codes = 1;           // staged-x tiles: 1 = row-pattern records instead of per-nonzero codes when >= 90 % of the tiles have one
                                   // (27-pt 300^3 1.349 -> 1.267 ms, 7-pt 400^3 1.025 -> 0.940 ms), 2 = whenever any tile has one, 0 = never
  int pattern_direct = 1;          // ... 1 = the records are first sought directly in the matrix (pat_direct_kernel: rows compared, then the window cover over
                                   // <= 256 column intervals per tile) and the window codes are built only when more than one tile in a hundred has none:
                                   // plan of 27-pt 300^3 6.2 -> 2.5 ms (first call 7.5 -> 3.8); 0 = always by way of the codes
  int pattern_codes_min_knnz = 10000;  // ... from this many thousand nnz (5-pt 1000^2, 5e6 nnz: 17.5 -> 18.3 us with the records)
  int colslab = 3;                 // rank 1, mode N, matrices whose x gather defeats the caches (kk_spmv_colslab.hip): a column-slab copy of the matrix.
                                   // 3 (default) = the DETERMINISTIC form (per-slab partial sums of every row, added in slab order: no atomics, the
                                   // same bits on every run), chosen by a RULE at the first call -- the analysis says "no column structure", x is
                                   // several L2s large and sampled windows of the matrix name nearly one line of x per nonzero (colslab_min_pct) --
                                   // nothing is timed; 4 = always the deterministic form (tests); 1 = the atomic form, chosen by timing both kernels
                                   // inside the first call (results vary in the last bits from run to run); 2 = always the atomic form (tests); 0 = never
  int colslab_min_pct = 85;        // ... the rule of 3: distinct lines of x per nonzero, in percent, from which the gather counts as cache-defeating
  int colslab_min_knnz = 20000;    // ... from this many thousand nnz
  int colslab_shift = 0;           // ... log2 of the columns per slab (0 = 2 MB of x)
  int colslab_rate_pct = 61;       // ... the selection rule's price ratio: bytes per second of the slab form / of the CRS kernel's x lines, in percent (4.24 / 6.95 TB/s on MI355X)
  int defer_rank1 = 0;             // 1 = the rank-1 analysis (tiles, window codes, pattern records: 5 ms on 27-pt 300^3) waits for the first rank-1
                                   // call; set by a caller whose first call is rank 2 (the Python SPMVHandle does: the reference's handle is set up
                                   // by its first spmv call too, for that call's rank).  Queries of the rank-1 plan return 0 until then.
  int check_entries = 0;           // debug aid: 1 = every call hashes the matrix's column array and compares it with the hash the analysis saw (one
                                   // extra pass over entries and a stream synchronisation per call): a structure edited in place under a live handle
                                   // is reported (KKAMD_ERR_STATE) instead of silently multiplied with the old analysis
  int colslab_const = 0;           // ... 1 = the caller promises constant matrix values (no tracking pass per call)
  int values_tracking = 0;         // how re-ordered copies of A.values (cached transpose, column-slab copy) follow value changes: 0 exact (bitwise
                                   // comparison against a shadow copy, every call), 1 the caller notifies (kkamd_spmv_plan_values_changed), 2
                                   // per-tile fingerprints (one stream, no shadow; a heuristic)
#ifdef KK_ABLATE                   // measurement build only (tools/, libkkamd_ablate.so): never part of libkkamd.so
  int ablate         = 0;          // switches parts of the kernels off (see the kernels)
  int lds_pad_kb     = 0;          // extra dynamic LDS per workgroup (caps workgroups per CU)
#endif
};
extern SpmvTuning g_spmv_default;

// 0, 1 or a power of two >= 2: the tile orders of xcd_order() (kk_common.h) are only bijective for those
inline bool valid_order_knob(int v) { return v == 0 || v == 1 || (v >= 2 && v <= (1 << 20) && (v & (v - 1)) == 0); }

}  // namespace kk

// Tile modes of the planned rank-1 kernel (low two bits of tinfo[b]; the upper bits index the tile's codes / record):
//   0 plain (entries + gather), 1 window codes + gather, 2 window codes + LDS-staged x, 3 row-pattern record + staged x
constexpr int kTilePlain = 0, kTileCodes = 1, kTileStaged = 2, kTilePattern = 3;

struct kkamd_mv_plan;   // kk_spmv_mv.hip
struct kkamd_mv4_plan;  // kk_spmv_mv.hip
struct kkamd_cs_plan;   // kk_spmv_colslab.hip
struct kkamd_mv5_plan;  // kk_spmv_mvblk.hip
struct kkamd_mv6_plan;  // kk_spmv_mvnnz.hip

struct kkamd_spmv_plan {
  // knobs the caller set on this handle, in order: replayed on the plan of the cached transpose when that is created (and forwarded to it
  // afterwards), so that a kernel choice forced on a handle also holds for its modes T / H
  std::vector<std::pair<std::string, int>> set_log;
  int64_t num_rows = 0, num_cols = 0, nnz = 0;
  const void* row_map = nullptr;
  int offset_type = 0, value_type = 1, algorithm = 0;
  kk::SpmvTuning tune;
  int tile = 0;             // nnz per workgroup of the analysed tiling (0 = no stream analysis)
  int64_t nblocks = 0;
  int num_cus = 256;
  int32_t* d_blk_row = nullptr;  // [nblocks+1] first row starting at or after b*tile
  void* d_carry = nullptr;       // [2*nblocks] 8-byte slots: head partials, then tail partials
  void* d_xpack = nullptr;       // rank-2: row-major packed copy of a column-major X (grown on demand)
  size_t xpack_bytes = 0;
  void* d_ypack = nullptr;       // rank-2 nonzero-split kernel: row-major Y scratch for a column-major Y
  size_t ypack_bytes = 0;
  const void* entries = nullptr; // the matrix's column array (identity check + analysis)
  // Window codes: per tile up to 16 column windows of 4096 and, per nonzero, a 16-bit code (window << 12 | column - window
  // base), stored in the order the kernel's work-items consume them -- only for the tiles that use them (tinfo)
  int32_t* d_tinfo = nullptr;    // [nblocks] mode | index << 2 (index: tile's position in d_wcode resp. d_pmeta)
  uint16_t* d_wcode = nullptr;   // [code_tiles * tile]
  int32_t* d_wbase = nullptr;    // [nblocks * 64] window meta: bases, LDS slots, x chunk columns
  int32_t* d_pmeta = nullptr;    // [nblocks * kPatW] row-pattern records (see pat_build_kernel), allocated when records are in use
  int32_t* d_list[4] = {nullptr, nullptr, nullptr, nullptr};   // per tile mode: ascending list of its tiles (null: none, or every tile)
  int64_t n_mode[4]  = {0, 0, 0, 0};                           // tiles per mode (the launch sizes)
  int64_t code_tiles = 0;        // tiles that read per-nonzero codes (modes 1, 2)
  int64_t staged_tiles = 0;      // tiles whose x window is staged in LDS (modes 2, 3)
  int64_t pat_tiles = 0;         // tiles decoded from a row-pattern record (mode 3)
  bool pat_direct = false;       // the records came straight from the matrix (pat_direct_kernel), no window codes were built
  int64_t plain_tiles = 0;       // tiles that read entries (mode 0)
  size_t plan_bytes = 0;         // HBM the analysis keeps
  // modes T/H: explicit transpose cached on first use (structure, permutation into A's values, refreshed values, sub-plan)
  void* d_t_rm = nullptr; int32_t* d_t_ent = nullptr; void* d_t_perm = nullptr; void* d_t_val = nullptr; unsigned long long* d_t_fp = nullptr;
  kkamd_spmv_plan* t_plan = nullptr;
  void* d_t_shadow = nullptr;    // A.values as the transpose last saw them (values_tracking 0)
  bool t_ready = false, t_failed = false, t_fp_valid = false, t_shadow_valid = false, t_shadow_failed = false, t_stale = true;
  bool rank1_deferred = false;   // knob defer_rank1: `tile` is set (the handle counts as analysed) but nothing of the rank-1 plan is built yet
  bool win_failed = false;       // the codes are not worth it on this matrix (or HBM cannot hold them): plain entries
  // rank-2 analysis (LDS-staged X tiles), built by the first rank-2 call that can use it
  kkamd_mv_plan* mv = nullptr;
  bool mv_failed = false;
  // rank-2 analysis of the plane-marching kernel (lattice strides, per-row conformity), built by the first call that asks for it
  kkamd_mv4_plan* mv4 = nullptr;
  bool mv4_tried = false;
  // rank-2 matrix-core kernel: per 16-row tile the union of its columns in blocks of four and a 64-bit occupancy mask per block
  kkamd_mv5_plan* mv5 = nullptr;
  bool mv5_tried = false;
  // rank-2 nonzero-split kernel: row index per nonzero, empty rows, carry slots
  kkamd_mv6_plan* mv6 = nullptr;
  bool mv6_tried = false;
  // rank-2 wave-private kernel: its row blocks in strip order (see mv_build_strip_order)
  int32_t* d_mv2_order = nullptr;
  // rank 2, gather kernel: rows longer than mv_long_T entries are left out of the wave-per-16-rows walk (one row group of a wave
  // would chew through them alone) and done by a workgroup each afterwards; found once, at the first rank-2 call
  int32_t* d_mv_long = nullptr; int64_t n_mv_long = 0, mv_long_T = 0, mv_long_nnz = 0; bool mv_long_known = false;
  // rank 1, column-slab copy (kk_spmv_colslab.hip): decided at the first mode-N call that can use it
  kkamd_cs_plan* cs = nullptr;
  bool cs_tried = false;
  double cs_crs_us = 0.0, cs_us = 0.0;     // what the selection measured, kept when the copy lost and was freed
  double cs_ratio = 0.0;                   // distinct lines of x per nonzero the selection rule of the deterministic form measured
  int mv2_rb = 0;
  bool mv2_tried = false, mv_period_known = false;
  int64_t mv_period = 0;
  // stream the plan's scratch (carry, packs) was last used on: a change of stream fences the old one first
  // (TPL_SpMV_Data::set_exec_space, sparse/src/KokkosSparse_spmv_handle.hpp:95-104)
  hipStream_t last_stream = nullptr;
  bool used = false;
  unsigned long long entries_hash = 0; bool entries_hash_known = false;     // knob check_entries
};

namespace kk {
// the plan's scratch (carry, packs) is stream-ordered: when the stream changes, the old one is fenced first
int  bind_stream(kkamd_spmv_plan* p, hipStream_t st);
int  check_plan(const kkamd_spmv_plan* p, const kkamd_crs_t* A);
int  check_entries_content(kkamd_spmv_plan* p, const kkamd_crs_t* A, hipStream_t st);   // knob check_entries; no-op otherwise
void mv_plan_destroy(kkamd_mv_plan* mv);
int64_t mv_plan_query(const kkamd_mv_plan* mv, int what);   // 0 tiles, 1 pattern tiles, 2 order in use, 3 bytes
void mv4_plan_destroy(kkamd_mv4_plan* p);
int64_t mv4_plan_query(const kkamd_mv4_plan* p, int what);  // 0 workgroups, 1 rows outside the stencil, 2 stencil entries, 3 bytes, 4 near stride
void mv5_plan_destroy(kkamd_mv5_plan* p);
int64_t mv5_plan_query(const kkamd_mv5_plan* p, int what);  // 0 tiles on the matrix core, 1 rows left to the gather rows, 2 column blocks, 3 bytes, 4 fill in 1/1000
// builds the analysis on first use (plan->mv5 stays null when the matrix does not qualify); then one pass per 16 right-hand sides
int  mv5_plan_build(kkamd_spmv_plan* plan, const kkamd_crs_t* A, hipStream_t st);
int  mv5_spmv(const kkamd_spmv_plan* plan, const kkamd_crs_t* A, const double* X, int64_t xs0, int64_t xs1, double* Y, int64_t ys0, int64_t ys1,
              int64_t nvec, double alpha, double beta, hipStream_t st);
void mv6_plan_destroy(kkamd_mv6_plan* p);
int64_t mv6_plan_query(const kkamd_mv6_plan* p, int what);  // 0 chunks, 1 empty rows, 2 bytes
int  mv6_plan_build(kkamd_spmv_plan* plan, const kkamd_crs_t* A, hipStream_t st);
int  mv6_spmv(kkamd_spmv_plan* plan, const kkamd_crs_t* A, const double* X, int64_t ldx, double* Y, int64_t ys0, int64_t ys1,
              int64_t nvec, double alpha, double beta, hipStream_t st);   // X row-major, ldx even, 16-byte aligned
void cs_plan_destroy(kkamd_cs_plan* cs);
int64_t cs_plan_query(const kkamd_cs_plan* cs, int what);   // 0 slabs, 1 log2 columns per slab, 2 bytes, 3 deterministic form
int  cs_pick_shift(const kkamd_crs_t* A, int x_elem, int shift_knob, bool det);   // log2 of the columns per slab cs_build uses
int  cs_build(kkamd_cs_plan** out, const kkamd_crs_t* A, int x_elem, int shift_knob, hipStream_t st, bool det = false);   // det: the deterministic form (per-slab partial sums, no atomics)
int  cs_apply(kkamd_cs_plan* cs, const kkamd_crs_t* A, int vector_type, const void* x, void* y, double alpha, double beta, int tracking, hipStream_t st);
int64_t values_fp_tiles(int64_t nnz);
// keeps a re-ordered copy of A.values (o_val[dst[i]] = val[i]) current under the "values_tracking" policy (kk_spmv_colslab.hip)
int  values_track(int tracking, bool promise, int offset_type, int value_type, int64_t nnz, const void* val, const void* dst, void* o_val,
                  unsigned long long* fp, void** shadow, bool* fp_valid, bool* shadow_valid, bool* shadow_failed, bool* stale, hipStream_t st);
int  cs_gather_ratio(const kkamd_crs_t* A, int x_elem, hipStream_t st, double* ratio);   // distinct 128-byte lines of x per nonzero over sampled windows
void cs_mark_stale(kkamd_cs_plan* cs);
void cs_reset_tracking(kkamd_cs_plan* cs);
// modes T / H of an analysed handle: the cached transpose (built on first use, values brought up to date) as a matrix + its plan;
// returns KKAMD_OK with *tplan == nullptr when there is none (knob off, too small, no memory): the caller scatters with atomics
int  transpose_view(kkamd_spmv_plan* plan, const kkamd_crs_t* A, hipStream_t st, kkamd_crs_t* At, kkamd_spmv_plan** tplan);
int  release_transient();
int  release_bitmap_pool();      // kk_spgemm.hip: the pooled bitmap store of the SpGEMM symbolic -> numeric hand-over
// rank 1 on the rank-2 plane-marching analysis (kk_spmv_mv.hip): builds the analysis on first use; returns 1 when it ran
int  march_spmv(kkamd_spmv_plan* plan, const kkamd_crs_t* A, const double* x, double* y, double alpha, double beta, hipStream_t st, int* ran);

// native 2-element vectors (accepted by __builtin_nontemporal_load; same syntax under clang and gcc)
typedef double kk_f64x2 __attribute__((vector_size(16)));
typedef float  kk_f32x2 __attribute__((vector_size(8)));
typedef int    kk_i32x2 __attribute__((vector_size(8)));
typedef unsigned kk_u32x2 __attribute__((vector_size(8)));
typedef unsigned kk_u32x4 __attribute__((vector_size(16)));
template <class T> struct vec2;
template <> struct vec2<double> { using type = kk_f64x2; };
template <> struct vec2<float>  { using type = kk_f32x2; };


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

}  // namespace kk

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

