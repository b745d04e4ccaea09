/* kkamd.h -- C ABI of libkkamd.so: the MI355X (gfx950) native implementation of the
 * KokkosSparse::spmv / KokkosSparse::spgemm hot path.
 *
 * This is the drop-in boundary.  It sits exactly where kokkos-kernels plugs vendor libraries in
 * today -- the "TPL specialisation" layer -- and takes what those specialisations hand over:
 * raw device pointers unwrapped from Kokkos::Views, extents, and the execution-space instance's
 * HIP stream.  Each entry point cites the reference interface it replaces (paths relative to the
 * kokkos-kernels 4.7.00 tree).  INTEGRATION.md shows the *_tpl_spec_{avail,decl}.hpp siblings a
 * maintainer adds on the reference side to bind them.
 *
 * Conventions
 *   - plain C, no C++/torch/Kokkos types; every pointer named d_* is a DEVICE pointer, borrowed
 *     for the duration of the call (the reference's Unmanaged views);
 *   - every function returns a kkamd_status (0 = ok) and never throws; kkamd_last_error() gives the
 *     message for the calling thread.  The C++ shim turns non-zero into the exception type the
 *     reference would throw (std::runtime_error / std::invalid_argument);
 *   - kkamd_stream_t is hipStream_t (a pointer to the opaque ihipStream_t); SpMV entry points are
 *     asynchronous on that stream, SpGEMM phases synchronise it where they must return counts;
 *   - CSR: zero-based, row_map has num_rows+1 offsets (int32 or int64), entries are int32 ordinals
 *     (the reference requires a signed ordinal: sparse/src/KokkosSparse_CrsMatrix.hpp:320),
 *     values float or double.
 */
#ifndef KKAMD_H
#define KKAMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KKAMD_VERSION 100

typedef struct ihipStream_t* kkamd_stream_t;

typedef enum {
  KKAMD_OK               = 0,
  KKAMD_ERR_INVALID_ARG  = 1, /* dimension/type/pointer problems (reference: std::runtime_error)        */
  KKAMD_ERR_UNSUPPORTED  = 2, /* type tuple or mode not implemented here: caller diverts to native path  */
  KKAMD_ERR_HIP          = 3, /* a HIP runtime call failed (reference: *_SAFE_CALL -> std::runtime_error)*/
  KKAMD_ERR_ALLOC        = 4,
  KKAMD_ERR_STATE        = 5  /* handle misuse, e.g. numeric before symbolic (std::invalid_argument)    */
} kkamd_status;

typedef enum { KKAMD_F32 = 0, KKAMD_F64 = 1 } kkamd_scalar_type;
typedef enum { KKAMD_I32 = 0, KKAMD_I64 = 1 } kkamd_offset_type;

/* KokkosSparse::SPMVAlgorithm (sparse/src/KokkosSparse_spmv_handle.hpp:32-47), CRS subset. */
typedef enum {
  KKAMD_SPMV_DEFAULT           = 0,
  KKAMD_SPMV_FAST_SETUP        = 1,
  KKAMD_SPMV_NATIVE            = 2,
  KKAMD_SPMV_MERGE_PATH        = 3,
  KKAMD_SPMV_NATIVE_MERGE_PATH = 4
} kkamd_spmv_algorithm;

/* CrsMatrix as the TPL layer sees it: (numRows, numCols, nnz, graph.row_map.data(),
 * graph.entries.data(), values.data()) -- sparse/tpls/KokkosSparse_spmv_tpl_spec_decl.hpp:332-338. */
typedef struct {
  int64_t num_rows;
  int64_t num_cols;
  int64_t nnz;
  const void* d_row_map; /* offset_type[num_rows + 1] */
  const void* d_entries; /* int32_t[nnz]              */
  const void* d_values;  /* value_type[nnz]           */
  int offset_type;       /* kkamd_offset_type         */
  int value_type;        /* kkamd_scalar_type         */
} kkamd_crs_t;

const char* kkamd_last_error(void);
int kkamd_version(void);
/* Profiling ranges (roctx, shown by rocprofv3 --marker-trace): what Kokkos::Profiling::pushRegion / popRegion map to on the
 * host side of the boundary.  The library's own entry points open ranges labelled like the reference's
 * ("KokkosSparse::spmv[TPL_KKAMD,double]", sparse/tpls/KokkosSparse_spmv_tpl_spec_decl.hpp:411-414). */
int kkamd_trace_push(const char* label);
int kkamd_trace_pop(void);
/* name of the device the library is running on and whether it is gfx950; used by tests. */
int kkamd_device_info(char* name, int name_len, int* is_gfx950, int* num_cus);

/* ------------------------------------------------------------------------------------------------
 * SpMV.  Replaces Impl::SPMV<Kokkos::HIP,...>::spmv and Impl::SPMV_MV<...>::spmv_mv
 * (sparse/impl/KokkosSparse_spmv_spec.hpp:92-135; vendor precedent
 * sparse/tpls/KokkosSparse_spmv_tpl_spec_decl.hpp:278-425 and ..._spmv_mv_tpl_spec_decl.hpp:284-430).
 *
 * A plan is the analogue of SPMVHandleImpl::tpl_rank1 / tpl_rank2
 * (sparse/src/KokkosSparse_spmv_handle.hpp:241-242): created lazily on the first call with a handle,
 * bound to ONE matrix for life (:273-277), destroyed with the handle.  plan == NULL is the
 * handle-less / SPMV_FAST_SETUP route: no analysis, no workspace.
 * ------------------------------------------------------------------------------------------------ */
typedef struct kkamd_spmv_plan kkamd_spmv_plan_t;

int kkamd_spmv_plan_create(kkamd_spmv_plan_t** plan, const kkamd_crs_t* A, int algorithm, kkamd_stream_t stream);
/* same, with this plan's expert knobs (see kkamd_spmv_plan_set) applied BEFORE the analysis they shape; the library-wide
 * defaults are not touched. */
int kkamd_spmv_plan_create_knobs(kkamd_spmv_plan_t** plan, const kkamd_crs_t* A, int algorithm, const char* const* keys,
                                 const int* values, int nknobs, kkamd_stream_t stream);
int kkamd_spmv_plan_destroy(kkamd_spmv_plan_t* plan);
/* Frees the calling host thread's scratch of the handle-less route (tile descriptors + carry slots, grown on demand and
 * otherwise kept for the thread's life) and the two process-wide buffers of the SpGEMM symbolic phase: the store of the structure it
 * keeps for the first numeric call (bitmaps and entry lists of the dense class's units: at most 0.225 of the free HBM, or
 * "spgemm_store_cap_mb") and the phase's temporaries (the two indices and the units' products: 0.7 GB at R-MAT scale 20); also the
 * 512-byte counter blocks (device + pinned host) that destroyed SpGEMM handles leave for the next handle.  The two large buffers are
 * kept between uses under the policy of "spgemm_pool_keep" because allocating and freeing GBs per handle costs more than the phase
 * itself. */
int kkamd_release_scratch(void);

/* y := alpha*op(A)*x + beta*y.  mode 'N','C' (== 'N' for real scalars), 'T','H' (== 'T').
 * vector_type is the scalar type of x and y (and of alpha/beta, the reference's coefficient_type);
 * supported (value_type, vector_type) pairs: (F64,F64), (F32,F32), (F32,F64).
 * beta == 0 overwrites y (NaN/Inf in the old y are discarded: sparse/src/KokkosSparse_spmv.hpp:145-154,
 * sparse/impl/KokkosSparse_spmv_impl.hpp:127-131).  alpha == 0 / empty A reduce to the y scaling. */
int kkamd_spmv(kkamd_spmv_plan_t* plan, const kkamd_crs_t* A, char mode, double alpha, const void* d_x, double beta,
               void* d_y, int vector_type, kkamd_stream_t stream);

/* Rank-2: X is (num_cols x nvec), Y is (num_rows x nvec) for 'N'; element (i,j) lives at
 * i*stride0 + j*stride1 (in elements), which covers LayoutLeft (1, ld), LayoutRight (ld, 1) and the
 * mixed pairs the rocSPARSE plug-in accepts (sparse/tpls/KokkosSparse_spmv_mv_tpl_spec_avail.hpp:108-134). */
int kkamd_spmv_mv(kkamd_spmv_plan_t* plan, const kkamd_crs_t* A, char mode, double alpha, const void* d_X,
                  int64_t x_stride0, int64_t x_stride1, double beta, void* d_Y, int64_t y_stride0, int64_t y_stride1,
                  int64_t nvec, int vector_type, kkamd_stream_t stream);

/* KokkosSparse::Experimental::spmv_struct (sparse/src/KokkosSparse_spmv.hpp:478-559, impl
 * sparse/impl/KokkosSparse_spmv_struct_impl.hpp:640-705): y := alpha*op(A)*x + beta*y for a matrix that comes from a
 * 3-pt (1-D), 5-/9-pt (2-D, stencil_type 1 = FD / 2 = FE) or 7-/27-pt (3-D) stencil on a structure[0] x structure[1] x
 * structure[2] grid, rows numbered i fastest.  Interior rows never read entries(): their idx-th value multiplies
 * x(row + offset(idx)); exterior rows take the CRS row.  y is not read when beta == 0 (BLAS convention; the reference's
 * functor evaluates beta*y + alpha*sum also then) and modes 'T'/'H' ignore the structure.  structure is a HOST array of ndim extents. */
int kkamd_spmv_struct(const kkamd_crs_t* A, char mode, int stencil_type, int ndim, const int64_t* structure, double alpha,
                      const void* d_x, double beta, void* d_y, int vector_type, kkamd_stream_t stream);

/* Expert knobs, the analogue of SPMVHandleImpl's public tuning members
 * (sparse/src/KokkosSparse_spmv_handle.hpp:243-252); per plan (kkamd_spmv_plan_set / kkamd_spmv_plan_create_knobs), or as
 * defaults for plans created later (kkamd_set_default).  Values outside the listed ranges return KKAMD_ERR_INVALID_ARG.
 *   SpMV, rank 1
 *     "kernel"          0 auto, 1 no-analysis vector kernel, 2 nnz-split kernel
 *     "lanes_per_row"   vector kernel, 0 = from nnz / row (the reference's vector_length)
 *     "nnz_per_thread"  nnz-split kernel: 4 (fp64 values only) | 8 | 16 nonzeros per work-item = 1024 / 2048 / 4096 per tile, 0 = by size
 *     "xcd_remap"       tile order: 0 dispatch, 1 XCD-contiguous, 2^k grouped (default 16)
 *     "window_codes"    per-tile column analysis of an analysed handle: 1 (default) 16-bit window codes + LDS-staged x where a tile's
 *                       columns allow it, 2 codes without staged x, 0 never; from "window_codes_min_knnz" thousand nonzeros and when
 *                       at least "window_codes_min_pct" percent of the tiles can use them (the others read entries, tile by tile);
 *                       "stream_variant" 6 attempts the analysis whatever the size (tests), 1 is the default
 *     "pattern_codes"   staged-x tiles: row-pattern records instead of per-nonzero codes; 1 (default) when >= 90 % of the tiles
 *                       decompose, 2 whenever one does, 0 never; from "pattern_codes_min_knnz" thousand nonzeros.  "pattern_direct" 1
 *                       (default): the records are first sought straight in the matrix (rows compared, the window cover over the <= 256
 *                       column intervals of a tile) and kept when at most one tile in a hundred has none -- those read entries --, no
 *                       window codes are built (plan of 27-pt 300^3: 6.2 -> 2.5 ms); 0: always by way of the codes.  Query "pattern_direct"
 *     "transient_min_knnz"  handle-less / FAST_SETUP calls analyse on the fly from this many thousand nonzeros (0 never)
 *     "explicit_transpose"  modes T/H with an analysed handle (rank 1 and rank 2): 1 (default) through a transpose cached in the plan when
 *                       it fits an eighth of free HBM, its values follow "values_tracking"; 2 the caller promises constant
 *                       values (no tracking pass either); 0 the reference's atomic scatter.  From "explicit_transpose_min_knnz"
 *     "values_tracking" how a re-ordered copy of A.values (cached transpose; column-slab copy) follows value changes: 0 (default) EXACT --
 *                       a shadow copy of A.values in A's order, compared bit for bit at every call (two streams of 8 B per nonzero),
 *                       changed 4096-value tiles are moved; without memory for the shadow every call copies all values; 1 NOTIFY --
 *                       kkamd_spmv_plan_values_changed says when, nothing is read in between; 2 FINGERPRINTS -- 128 bits per tile, one
 *                       stream, no shadow: a heuristic (a changed tile with an unchanged fingerprint is missed)
 *     "check_entries"   debug aid, 0 (default) / 1: every call hashes the matrix's column array (one pass + a stream synchronisation) and
 *                       returns KKAMD_ERR_STATE when it differs from what the handle first saw: a structure edited in place under a live
 *                       handle (the pointer comparison of every call cannot see that; rocSPARSE's analysis has the same contract)
 *     "colslab"         mode N on matrices whose x gather defeats the caches (most tiles read plain entries, x is >= 16 MB, from
 *                       "colslab_min_knnz" thousand nonzeros): a second copy of the matrix in column-slab order (entries sorted by 2 MB
 *                       segments of x, then by row; nnz * (8 + value + offset) bytes) whose values follow "values_tracking".
 *                       3 (default): the DETERMINISTIC form -- per-slab partial sums of every row with one writer each, slabs added in
 *                       ascending order: no atomics, the same bits on every run --, chosen at the first call by a RULE, nothing is timed:
 *                       sampled windows of the matrix name at least "colslab_min_pct" (85) percent of a 128-byte line of x per nonzero,
 *                       and the bytes the slab form streams, priced at "colslab_rate_pct" (61: its rate over the rate at which the CRS
 *                       kernel's gathers pull lines, 4.24 / 6.95 TB/s measured on MI355X) of the lines the CRS kernel would pull, are
 *                       fewer by 10 %; 4 always the deterministic form (tests); 1 the ATOMIC form (products reach y through atomics:
 *                       results agree with the CRS kernel to rounding and vary in the last bits from run to run), chosen by timing both
 *                       kernels ON THE CALLER'S STREAM inside the first call (the call blocks), kept when 10 % faster; 2 always the
 *                       atomic form (tests); 0 never.  "colslab_shift" log2 of the columns per slab (0 = automatic),
 *                       "colslab_const" 1 = the caller promises constant matrix values (no tracking pass)
 *   SpMV, rank 2
 *     "mv6"             nonzero-split kernel (kk_spmv_mvnnz.hip; analysed plan, fp64 vectors): the nonzeros are cut into chunks of 128 per
 *                       16-lane group, the plan keeps the row index of every nonzero (4 B per nonzero), rows cut by a chunk boundary are
 *                       finished from carry slots (no atomics, deterministic).  1 (default) on matrices with at least
 *                       "mv6_min_long_pct" (10) percent of their nonzeros in rows above 4 x the average length (at least 64): power-law
 *                       graphs; 2 whenever the gather kernel would run; 0 never
 *     "mv_kernel"       0 auto (plane-marching kernel where it applies -- analysed plan, fp64 vectors, right-hand sides in
 *                       blocks of 16 (a remainder: one more pass over the last 16 columns when beta = 0, else the gather kernel), a matrix that verifies as a radius-1 lattice stencil --, else the wave-private gather
 *                       kernel), 1 generic strided kernel, 2 wave-private gather kernel, 3 LDS-staged X tiles, 4 = 0 without the matrix-core
 *                       kernel, 5 = 0 (the matrix-core kernel where its analysis accepts the matrix, see "mv5"), 6 = the nonzero-split or the gather kernel
 *     "mv5"             matrix-core kernel (v_mfma_f64_16x16x4f64; analysed plan, fp64 vectors, any width and strides; not on matrices
 *                       the plane-marching kernel takes): the rows are cut into tiles of 16, a tile is DESCRIBED by the union of its
 *                       columns in blocks of four plus a 64-bit occupancy mask per block (24 B per block instead of 4 B per entry;
 *                       the values stay where they are) when its rows ascend strictly, it holds 1..2048 entries and at least
 *                       "mv5_min_fill_pct" (default 25) percent of its 16 x 4 operand slots hold an entry; the other tiles' rows go to a
 *                       gather kernel.  1 (default) = use it when at most "mv5_max_other_pct" (default 50) percent of the rows are
 *                       left to the gather rows, 2 = whenever a tile can be described (no fill threshold: tests), 0 = never
 *     "mv_order"        row-block order: 2 (default) strips from the far stride found in the matrix (falls back to "mv_remap"),
 *                       1 XCD-contiguous (LDS-staged kernel), 0 "mv_remap": 0 dispatch, 1 XCD-contiguous, 2^k grouped (default 16)
 *     "mv_strip_min_kb" / "mv_strip_l2_kb"  when strips engage / how much of an XCD's L2 a strip's X rows may take
 *     "mv_glds"         LDS-staged kernel: X window through global_load_lds (1) or registers (0)
 *     "mv4_2d"          plane-marching kernel on 2-D lattices (lines grouped 32..128 at a time into the planes it marches through;
 *                       the first and last line of a group go to its gather rows): 1 (default) on, 0 off
 *     "mv4_xcol"        plane-marching kernel, column-major X (LayoutLeft): 1 (default) the X pieces of a plane are fetched column-wise
 *                       (whole cache lines per load) into slab rows swizzled in LDS; 0 the order used for general strides
 *     "mv4_wg_per_cu"   plane-marching kernel: workgroups per CU the split along the far stride aims for (default 8, 1..64)
 *     "march"           rank 1 on the plane-marching analysis (fp64 vectors, matrices that verify as lattice stencils): 0 (default)
 *                       off -- measured equal to the planned stream kernel on C2 --, 1 on; "march_planes" planes per workgroup (20)
 *   kkamd_spmv_struct (global only): "struct_remap", "struct_group", "struct_strip" workgroup orders, all off
 *   "verbose" (global): 1 = the library reports what it chose on stdout
 *   SpGEMM (kkamd_set_default only) "spgemm_win_bits" (columns per LDS bitmap pass), "spgemm_val_cap", "spgemm_val_shape",
 *          "spgemm_val_la", "spgemm_force_unsorted", "spgemm_emit_chunked" (test hooks for alternative code paths);
 *          "spgemm_emit_sort" (1: entries(C) of rows with more than 256 entries out of at most 2048 products are sorted in LDS, eight rows
 *          per CU; 0: they take the bitmap kernel like every other dense row), "spgemm_col_quads" (0 / 4: 16-byte loads of entries(B) per
 *          work-item and step in the row-by-row bitmap kernels), "spgemm_sym_units" (1, default: the symbolic phase counts its dense class
 *          by units = (row of C, window of columns), each a workgroup of its own around a 32 KB LDS bitmap, described by heads built from an
 *          index of B and of A's entries at window granularity; 0: one workgroup per row, as before round 6), "spgemm_unit_bits" (log2 of a
 *          unit's window, 6..18, default 18; small values are for tests), "spgemm_store_cap_mb" (upper limit, in MB, of the structure the
 *          symbolic phase keeps for the first numeric call; 0 = default: 0.225 of the free HBM; room goes to the heaviest rows first and
 *          rows past the limit walk their products again in the numeric phase), "spgemm_quad_rows" (wave-per-row kernels with four rows per wave, 16 lanes each: 1 = for the rows
 *          with at most 64 products (symbolic) / 32 entries (numeric) -- stencil and multigrid products --, 2 = for every row of the wave bin, 0 = never).
 * Knobs that switch parts of kernels OFF ("ablate", "lds_pad_kb", "struct_lds_pad_kb", "spgemm_debug") exist only in the
 * measurement build libkkamd_ablate.so (csrc: make ablate, -DKK_ABLATE); libkkamd.so answers KKAMD_ERR_INVALID_ARG. */
int kkamd_spmv_plan_set(kkamd_spmv_plan_t* plan, const char* key, int value);
int kkamd_set_default(const char* key, int value);
/* The caller has written to A.values since the plan's last SpMV.  The native CRS kernels read A.values at every call and need no
 * notice; a plan that keeps a RE-ORDERED COPY of the values (the cached transpose of modes T / H; the opt-in column-slab copy) copies
 * them again at its next call.  Under "values_tracking" 0 (default) and 2 the plan notices changes itself and this call is optional;
 * under 1 it is the only way the copies learn of a change.  No reference counterpart: the reference's handle holds no values
 * (sparse/src/KokkosSparse_spmv_handle.hpp:273-277), vendor analyses of this kind ask for the same call (rocsparse update-values). */
int kkamd_spmv_plan_values_changed(kkamd_spmv_plan_t* plan);
/* What the analysis of a plan produced: "tile" (nnz per workgroup, 0 = no tiling), "tiles", per-mode tile counts "plain_tiles" /
 * "code_tiles" / "staged_tiles" / "pattern_tiles", "window_codes" (1 if any tile uses the column analysis), "window_staged_x",
 * "plan_bytes" (HBM the analysis keeps), "transpose_cached", rank 2: "mv_tiles", "mv_staged_tiles", "mv_order" (order in use),
 * "mv_period" (far stride found), "mv_plan_bytes", plane-marching kernel: "mv4_workgroups" (0 = not in use), "mv4_other_rows"
 * (rows left to its gather kernel), "mv4_stencil" (entries of the stencil), "mv4_near_stride"; matrix-core kernel: "mv5_tiles" (described
 * 16-row tiles, 0 = not in use), "mv5_other_rows", "mv5_blocks" (column blocks = MFMA instructions per pass), "mv5_fill_permille"; nonzero-split kernel: "mv6_chunks" (0 = not in use),
 * "mv6_empty_rows"; "mv_long_rows" / "mv_long_nnz" (rows above the long-row threshold and their entries); "march_workgroups" (rank-1 marching kernel, 0 = not in use);
 * column-slab copy: "colslab" (1 = in use), "colslab_tried", "colslab_slabs", "colslab_shift", "colslab_bytes", and what the selection
 * measured, "colslab_crs_us" / "colslab_us" (microseconds per call of the CRS kernel / of the copy; 0 = not measured). */
int kkamd_spmv_plan_query(const kkamd_spmv_plan_t* plan, const char* key, int64_t* value);
/* Copies a per-tile array of the analysis to a HOST buffer of `count` int32: "tile_first_row" (tiles + 1 entries: the first row that
 * starts at or after nonzero b * tile, bit 31 set when the tile starts inside a row -- the nnz-split counterpart of the
 * reference's merge-path diagonal search, sparse/impl/KokkosSparse_merge_matrix.hpp:196-227) or "tile_mode" (tiles entries,
 * low two bits 0 plain / 1 codes / 2 codes + staged x / 3 row-pattern record). */
int kkamd_spmv_plan_export(const kkamd_spmv_plan_t* plan, const char* what, void* h_out, int64_t count);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU SpMV: the CrsMatrix is 1-D row-partitioned over the GPUs of one node, one process per GPU.  Rank r owns the row slab
 * [row_offsets[r], row_offsets[r+1]) of A (LOCAL row_map, GLOBAL column indices), the matching slab of y and shard of x; one
 * exchange of x entries per SpMV (RCCL over xGMI), no other communication.  The reference has no counterpart (upstream this is
 * Tpetra's job); the local SpMV is kkamd_spmv.
 *   exchange  0 auto (the column-range halo when it moves less than half of the all-gather, else the column-set halo when that
 *             does, else the all-gather), 1 halo by column RANGE (the x ranges the slab's columns span, contiguous pieces point to
 *             point straight into x), 2 all-gather (every shard to every rank through the transport's collective), 3 all-gather
 *             by peer-to-peer pulls (every rank maps the others' x buffers through hipIpc once and pulls world - 1 shards with
 *             concurrent copies between two 8-byte barrier collectives: all 7 xGMI links at once, no ring), 4 halo by column SET
 *             (the general importer: exactly the x entries the slab's off-slab columns name -- per-peer index lists agreed once,
 *             a pack kernel, point-to-point pieces, a scatter kernel), 5 all-gather in the form the operator picks BY TIMING at
 *             creation: every form the transport and the runtime offer (the collective; every shard to every peer point to point;
 *             peer-to-peer pulls) runs one warm-up and two timed exchanges with all ranks in step, and the form with the smallest
 *             maximum over the ranks stays.  Since round 6 the auto rule's all-gather is 5; 2 and 3 remain as forced forms;
 *   overlap   1: rows that reference only the rank's own x entries are computed while the halo is in flight.
 * Transport: by default RCCL, bound at run time from the librccl.so.1 already in the process; rank 0 obtains the 128-byte id
 * with kkamd_dist_unique_id and the host's launcher (MPI, torch.distributed, a file) hands it to every rank.  A host that
 * communicates otherwise passes its own kkamd_transport_t (both functions are stream-ordered; sizes in bytes).
 * ------------------------------------------------------------------------------------------------ */
typedef struct kkamd_dist_spmv kkamd_dist_spmv_t;
typedef struct {
  void* ctx;
  /* every rank contributes bytes_per_rank bytes; d_recv receives world * bytes_per_rank bytes in rank order */
  int (*all_gather)(void* ctx, const void* d_send, void* d_recv, int64_t bytes_per_rank, kkamd_stream_t stream);
  /* one batch of point-to-point transfers: nsend sends and nrecv receives, matched by peer */
  int (*exchange)(void* ctx, int nsend, const void* const* d_send, const int64_t* send_bytes, const int* send_peer, int nrecv,
                  void* const* d_recv, const int64_t* recv_bytes, const int* recv_peer, kkamd_stream_t stream);
} kkamd_transport_t;

int kkamd_dist_unique_id(void* id128);
/* Preflight of the built-in RCCL transport on the calling rank alone (no counterpart in the reference, which has no distributed
 * layer: cmake/fake_tribits.cmake:210 runs every test with NUM_MPI_PROCS 1): binds librccl, forms a ONE-rank communicator and runs
 * every entry point the N-rank exchanges use -- ncclAllGather in place and out of place, a group of ncclSend / ncclRecv to itself,
 * an empty group -- on `bytes` of device data that are compared afterwards.  A binding or ABI problem shows as an error code here
 * instead of a hang in the first exchange of an N-rank job. */
int kkamd_dist_transport_selftest(int64_t bytes, kkamd_stream_t stream);
/* world == 1 with id128 given and exchange != 0: the one-rank job still forms its RCCL communicator and runs the forced exchange
 * through it (the in-place ncclAllGather of one shard; groups with no peers) -- the N-rank code path on one GPU. */
int kkamd_dist_spmv_create(kkamd_dist_spmv_t** op, const kkamd_crs_t* A_local, const int64_t* row_offsets /* host, world + 1 */,
                           int world, int rank, const void* id128 /* NULL with a transport; optional when world == 1 */,
                           const kkamd_transport_t* transport /* NULL = RCCL */, int algorithm, int exchange, int overlap,
                           int vector_type, kkamd_stream_t stream);
int kkamd_dist_spmv_destroy(kkamd_dist_spmv_t* op);
/* x lives in a full-length buffer owned by the operator: *d_x_local is the rank's own window of it (a caller that keeps its x
 * shard there never copies it), *d_x_full the whole buffer.  Either pointer argument may be NULL. */
int kkamd_dist_spmv_x_local(kkamd_dist_spmv_t* op, void** d_x_local, void** d_x_full);
/* y_shard := alpha * A_local * exchanged(x) + beta * y_shard.  d_x_shard may be the pointer of kkamd_dist_spmv_x_local (no copy)
 * or any device buffer of the shard's length (copied in).  what: 0 the whole step, 1 the exchange only, 2 the local SpMV only
 * (measurement aids: 1 and 2 split a step into its two costs). */
int kkamd_dist_spmv_apply(kkamd_dist_spmv_t* op, double alpha, const void* d_x_shard, double beta, void* d_y_shard, int what,
                          kkamd_stream_t stream);
/* "exchange" (0 local, 1 halo by column range, 2 all-gather, 3 all-gather by peer-to-peer pulls, 4 halo by column set, 5 all-gather in
 * the timed form: "allgather_selected" 1, "allgather_form" 0 collective / 1 send-receive / 2 peer-to-peer pulls,
 * "allgather_us_collective" / "_sendrecv" / "_p2p" the maximum over the ranks measured at creation, -1 = form not offered), "exchange_bytes"
 * (received per SpMV), "interior_rows", "parts", "sends", "recvs", "part0_rows" (rows of the interior view, or of the slab when it is
 * not split) and "part0_<key>" = kkamd_spmv_plan_query(<key>) of that view's plan (e.g. "part0_pattern_tiles") */
int kkamd_dist_spmv_query(const kkamd_dist_spmv_t* op, const char* key, int64_t* value);

/* ------------------------------------------------------------------------------------------------
 * SpGEMM.  Replaces Impl::SPGEMM_SYMBOLIC<...>::spgemm_symbolic and
 * Impl::SPGEMM_NUMERIC<...>::spgemm_numeric (sparse/impl/KokkosSparse_spgemm_symbolic_spec.hpp:72-126,
 * ..._numeric_spec.hpp:91-145; vendor precedent sparse/tpls/KokkosSparse_spgemm_symbolic_tpl_spec_decl.hpp:367-,
 * ..._numeric_tpl_spec_decl.hpp:259-).  A is m x n, B is n x k, C is m x k.
 *
 * The handle is the analogue of SPGEMMHandle's cross-phase state (c_nnz, row flops, max nnz per row:
 * sparse/src/KokkosSparse_spgemm_handle.hpp:231-247).  Contract honoured: symbolic fills row_map C and
 * returns nnz(C); the caller allocates entries/values of that size; numeric fills both, with every row's
 * columns in ascending order (the reference sorts after numeric:
 * sparse/impl/KokkosSparse_spgemm_numeric_spec.hpp:138-140); numeric may be repeated with new values.
 * Inputs need not be sorted or merged.
 * ------------------------------------------------------------------------------------------------ */
typedef struct kkamd_spgemm_handle kkamd_spgemm_handle_t;

int kkamd_spgemm_create(kkamd_spgemm_handle_t** handle);
int kkamd_spgemm_destroy(kkamd_spgemm_handle_t* handle);

int kkamd_spgemm_symbolic(kkamd_spgemm_handle_t* handle, int64_t m, int64_t n, int64_t k, const void* d_row_mapA,
                          const int32_t* d_entriesA, const void* d_row_mapB, const int32_t* d_entriesB,
                          void* d_row_mapC, int offset_type, int64_t* c_nnz, kkamd_stream_t stream);

int kkamd_spgemm_numeric(kkamd_spgemm_handle_t* handle, int64_t m, int64_t n, int64_t k, const void* d_row_mapA,
                         const int32_t* d_entriesA, const void* d_valuesA, const void* d_row_mapB,
                         const int32_t* d_entriesB, const void* d_valuesB, const void* d_row_mapC,
                         int32_t* d_entriesC, void* d_valuesC, int offset_type, int value_type,
                         kkamd_stream_t stream);

/* Options: what SPGEMMHandle / KokkosKernelsHandle setters map to (sparse/src/KokkosSparse_spgemm_handle.hpp:427-501,562-616;
 * sparse/src/KokkosKernels_Handle.hpp:380-465).  Set before the phase they shape.
 *   "algorithm"            SPGEMMAlgorithm value: SPGEMM_KK (default) and its aliases KK_MEMORY / KK_SPEED / KK_MEMSPEED / KK_LP run the
 *                          LDS hash-accumulator numeric; SPGEMM_KK_DENSE runs the dense-accumulator numeric (impl_speed.hpp:28-150);
 *                          SPGEMM_DEBUG / SPGEMM_SERIAL (host-sequential in the reference; every algorithm's C leaves the public
 *                          spgemm_numeric sorted, numeric_spec.hpp:138-140, so their C is everybody's C) run the hash algorithm
 *   "accumulator"          SPGEMMAccumulator: 1 = dense (same as SPGEMM_KK_DENSE), 0 / 2 = hash
 *   "compression"          B compression for the symbolic phase (impl_compression.hpp): 0 off (default), 1 keep when it pays, 2 always
 *   "compression_cut_off"  keep the compressed B when it leaves at most this share of the symbolic work (default 0.85)
 *   "verbose"              1: chosen algorithm, row bins, kernels and compression decision on stdout (KOKKOSKERNELS_VERBOSE)
 *   "entries_computed"     the reference's are_entries_computed() as the caller knows it: 0 forces the next numeric call to write
 *                          entries(C) again (re-allocated / overwritten arrays); 1 leaves the decision to the handle, which keeps
 *                          entries(C) across numeric calls only when it wrote them into the same row_map / entries arrays itself
 *                          (numeric reuse, sparse/tpls/KokkosSparse_spgemm_numeric_tpl_spec_decl.hpp:288-329)
 * Hints -- accepted, recorded, reported under "verbose", without effect, exactly as the reference's rocSPARSE / cuSPARSE paths treat
 * them (the reference's driver and unit tests set them before every spgemm: perf_test/sparse/KokkosSparse_spgemm.cpp:311-317,378-399,
 * sparse/unit_test/Test_Sparse_spgemm.hpp:91-92): "team_work_size", "shmem_size", "suggested_team_size", "suggested_vector_size",
 * "dynamic_scheduling", "min_hash_size_scale", "first_level_hash_cut_off", "mkl_sort_option", "mkl_keep_output",
 * "mkl_convert_to_1base", "multi_color_scale", "read_write_cost_calc", "compression_steps", "max_col_dense_acc", "sort_option"
 * (rows of C always leave column-sorted).  Any other key: KKAMD_ERR_INVALID_ARG. */
int kkamd_spgemm_set(kkamd_spgemm_handle_t* handle, const char* key, double value);
/* what: 0 c_nnz, 1 total multiplications (the reference's original_overall_flops / 2),
 * 2 max row flops, 3 max nnz in a row of C, 4 symbolic called, 5 numeric called, 6 B was compressed for the symbolic phase,
 * 7 symbolic insertions after compression, 8 numeric algorithm in use (0 hash, 1 dense accumulator), 9 the SPGEMMAlgorithm value the
 * caller set (4 = SPGEMM_DEFAULT until then), 10 number of distinct hints recorded, 11 the last numeric call kept the entries(C)
 * of the dense rows from the previous call (numeric reuse), 12 rows whose entries(C) the last numeric call wrote from a bitmap kept by the
 * symbolic phase, 13 bitmaps the symbolic phase holds at this moment (they are freed by the numeric call that uses them), 14 rows whose entries(C) the
 * last numeric call copied from the entry lists the symbolic phase left (dense rows whose bitmap is not kept), 15 rows whose entries(C)
 * the last numeric call sorted in LDS (rows of more than 256 entries out of at most 2048 products), 16 - 18 rows / items of the column-block
 * value kernel, 19 units (row of C, window of 2^18 columns) the last symbolic phase counted its dense class by (0: none, or row by row),
 * 20 units whose bitmap it kept for the numeric phase, 21 rows of the class whose every unit kept its structure (bitmap or entry list),
 * 22 bytes the process-wide store of kept structure holds at this moment, 23 the most it has held (process-wide; both are independent of
 * the handle passed; kkamd_release_scratch returns the store to the device).
 * With units, 12 counts the rows whose entries(C) the last numeric call wrote from kept units, 13 the rows held, 14 those of 12 with at
 * least one unit kept as an entry list. */
int kkamd_spgemm_get(kkamd_spgemm_handle_t* handle, int what, int64_t* value);
/* the recorded value of a hint key (the reference's get_* of the same setter); KKAMD_ERR_INVALID_ARG when the key was never set */
int kkamd_spgemm_get_hint(kkamd_spgemm_handle_t* handle, const char* key, double* value);

/* Row-partitioned SpGEMM over the GPUs of one node: rank r owns the row slab [row_offsets[r], row_offsets[r+1]) of A (LOCAL row_map)
 * and of C = A*B; B is replicated on every GPU, so there is no data-path communication at all (this is what lets BASELINE config 4
 * as specified -- R-MAT scale 22, nnz(C) = 7.2e10 = 863 GB -- fit eight GPUs).  No reference counterpart (upstream: Tpetra).
 *   _partition  contiguous slabs of near-equal MULTIPLICATIONS (row flops of sparse/impl/KokkosSparse_spgemm_impl_symbolic.hpp:1108-1185,
 *               computed on the device from the full A and B structure): fills row_offsets[world + 1] (host) and, when given,
 *               the multiplications of every slab; every rank that calls it with the same A and B gets the same answer.
 *   _symbolic / _numeric   kkamd_spgemm_symbolic / _numeric on the rank's slab (checked against the partition), state in the operator;
 *   _handle     the operator's SpGEMM handle, for kkamd_spgemm_set / _get;
 *   _query      "row0", "rows_local", "rows_global", "c_nnz_local", "mults_local". */
typedef struct kkamd_dist_spgemm kkamd_dist_spgemm_t;
int kkamd_dist_spgemm_partition(int64_t m, const void* d_row_mapA, const int32_t* d_entriesA, const void* d_row_mapB, int offset_type, int world,
                                int64_t* row_offsets /* host, world + 1 */, int64_t* mults_per_rank /* host, world, or NULL */, kkamd_stream_t stream);
int kkamd_dist_spgemm_create(kkamd_dist_spgemm_t** op, int world, int rank, const int64_t* row_offsets /* host, world + 1 */);
int kkamd_dist_spgemm_destroy(kkamd_dist_spgemm_t* op);
kkamd_spgemm_handle_t* kkamd_dist_spgemm_handle(kkamd_dist_spgemm_t* op);
int kkamd_dist_spgemm_symbolic(kkamd_dist_spgemm_t* op, int64_t m_local, int64_t n, int64_t k, const void* d_row_mapA_local, const int32_t* d_entriesA_local,
                               const void* d_row_mapB, const int32_t* d_entriesB, void* d_row_mapC_local, int offset_type, int64_t* c_nnz_local,
                               kkamd_stream_t stream);
int kkamd_dist_spgemm_numeric(kkamd_dist_spgemm_t* op, int64_t m_local, int64_t n, int64_t k, const void* d_row_mapA_local, const int32_t* d_entriesA_local,
                              const void* d_valuesA_local, const void* d_row_mapB, const int32_t* d_entriesB, const void* d_valuesB,
                              const void* d_row_mapC_local, int32_t* d_entriesC_local, void* d_valuesC_local, int offset_type, int value_type,
                              kkamd_stream_t stream);
int kkamd_dist_spgemm_query(const kkamd_dist_spgemm_t* op, const char* key, int64_t* value);


/* ------------------------------------------------------------------------------------------------
 * Helpers either side of the path (KokkosSparse::sort_crs_matrix, sparse/src/KokkosSparse_SortCrs.hpp:43-120;
 * kk_exclusive_parallel_prefix_sum, common/src/KokkosKernels_SimpleUtils.hpp:86-135) and the synthetic
 * inputs of the benchmark configurations, generated in place in HBM
 * (test_common/KokkosKernels_Test_Structured_Matrix.hpp, BC = 1 on every face).
 * ------------------------------------------------------------------------------------------------ */
int kkamd_sort_crs(int64_t num_rows, const void* d_row_map, int32_t* d_entries, void* d_values, int offset_type,
                   int value_type, kkamd_stream_t stream);

/* KokkosSparse::sort_and_merge_matrix (sparse/src/KokkosSparse_SortCrs.hpp:304-363): sort every row by column and
 * collapse runs of equal columns into one entry whose value is the sum of the run (left to right).  Two calls:
 *   1. d_entries_out == NULL: sorts (d_entries, d_values) IN PLACE, writes the merged row_map to d_row_map_out
 *      (num_rows + 1 offsets) and the merged entry count to *nnz_out;
 *   2. d_entries_out / d_values_out sized *nnz_out: fills them (the inputs must be the ones sorted by call 1).
 * d_values may be NULL (graph only). */
int kkamd_sort_and_merge(int64_t num_rows, const void* d_row_map, int32_t* d_entries, void* d_values, int offset_type,
                         int value_type, void* d_row_map_out, int32_t* d_entries_out, void* d_values_out, int64_t* nnz_out,
                         kkamd_stream_t stream);

/* KokkosSparse::Impl::transpose_matrix (sparse/src/KokkosSparse_Utils.hpp:338-400): CRS of the transpose into
 * preallocated d_t_row_map (num_cols + 1), d_t_entries / d_t_values (nnz).  Rows of the result are column-sorted (the
 * reference leaves the order to its atomics).  d_values may be NULL (transpose_graph, :402-440). */
int kkamd_transpose(int64_t num_rows, int64_t num_cols, int64_t nnz, const void* d_row_map, const int32_t* d_entries,
                    const void* d_values, int offset_type, int value_type, void* d_t_row_map, int32_t* d_t_entries,
                    void* d_t_values, kkamd_stream_t stream);
int kkamd_exclusive_scan(void* d_data, int64_t n, int offset_type, kkamd_stream_t stream);

/* dim = 2 or 3; stencil 0 = FD (5/7-pt), 1 = FE (9/27-pt).  With d_entries == NULL only row_map is
 * filled (and *nnz returned), so the caller can size entries/values. */
int kkamd_gen_laplace(int dim, int stencil, int64_t nx, int64_t ny, int64_t nz, void* d_row_map, int32_t* d_entries,
                      void* d_values, int offset_type, int value_type, int64_t* nnz, kkamd_stream_t stream);
/* One contiguous row slab [row_begin, row_begin+row_count) of the same matrix: local row_map (starting
 * at 0), GLOBAL column indices -- the per-GPU piece of the 1-D row partition. */
int kkamd_gen_laplace_rows(int dim, int stencil, int64_t nx, int64_t ny, int64_t nz, int64_t row_begin,
                           int64_t row_count, void* d_row_map, int32_t* d_entries, void* d_values, int offset_type,
                           int value_type, int64_t* nnz, kkamd_stream_t stream);

/* Device streaming-read microbenchmark (bench tooling): reads `bytes` from d_data with `loads` independent
 * 16-byte loads in flight per lane; the measured HBM read ceiling quoted beside roofline fractions. */
int kkamd_bench_read(const void* d_data, int64_t bytes, int loads, int nontemporal, int persistent, void* d_out,
                     kkamd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KKAMD_H */
