// kk_spgemm.hip -- C = A*B over CSR for gfx950: symbolic (row_map of C) + numeric (entries, values).
//
// Reference structure being replaced (sparse/impl/KokkosSparse_spgemm_impl_def.hpp:25-136): row-flop
// estimate (K8) -> [optional bit-compression of B] -> hash symbolic with a 16 KB LDS first level and a
// pooled global second level, one Kokkos thread per row (K10) -> prefix sum (K11) -> hash numeric in three
// team shapes (K12-K14) -> a separate per-row sort pass over C (K17).
//
// gfx950-native structure here.  A CU has 160 KB of LDS, so every per-row working set (hash table, column bitmap,
// value window) lives in LDS and no second level / memory pool exists; HBM holds accumulators only for the few rows of
// A with more than 4096 entries (or for dense rows when B is not column-sorted):
//   1. spgemm_flops_kernel      upper bound per row, total multiplications, max; rows_sorted_kernel: is B sorted?
//   2. rows are BINNED by that bound (symbolic) / by their exact nnz (numeric); each bin gets the launch shape
//      that fits it:
//      symbolic   flops <= 1365    wave per row, 2048-slot key table (4 rows per workgroup)
//                 flops <= 2048    256-thread workgroup, 4096-slot table
//                 flops <= 8192    1024-thread workgroup, 16384-slot table (64 KB: two workgroups per CU)
//                 above            1024-thread workgroup, k-bit column BITMAP in LDS (2^20 columns per pass), popcount
//      numeric    nnz <= 256       wave per row, 512-slot key+value table, compaction + rank-by-counting -> sorted
//                 nnz <= 2048/5461 (only when B is unsorted) workgroup per row, 4096/8192-slot table, bitonic network
//                 above (dense)    a) the bitmap kernel again, now EMITTING entries(C) in ascending order;
//                                  b) values: the sorted entries are cut into windows of 2048, each hashed into an LDS
//                                     table; every A entry keeps a cursor into its (sorted) B row, so a window only
//                                     streams the part of each B row that falls inside it (dense_vals / hub_vals);
//                                  c) A rows > 4096 entries or unsorted B: many workgroups per row, L2 atomics into
//                                     a k-wide HBM accumulator, gathered in C order (hub_acc / hub_extract).
//      Block-per-row kernels walk the row's products FLAT (flat_products): a scan of the B row lengths lets work-item q
//      take product q, so all loads of a lane are independent -- sub-groups chasing "their" B row were bound by the
//      entries(A) -> row_map(B) -> entries(B) latency chain.
//      Small tables: open addressing, linear probing, hash (col*107) & mask, empty = -1 -- the same function as the
//      reference's linear-probe kernels (sparse/impl/KokkosSparse_spgemm_impl_kkmem.hpp:17,679).
//   3. exclusive scan of the counts -> row_map C, nnz(C) returned to the host.
//   4. every numeric kernel leaves its rows column-sorted, so the reference's extra sort pass over C
//      (numeric_spec.hpp:138-140) disappears.
// Symbolic results are exact (bit-identical row_map / entries to the SPGEMM_DEBUG oracle after its
// sort); numeric sums are order-dependent (atomics) and compared at 1e-6 relative.
#include "kk_common.h"
#include "kk_scan.h"
#include <climits>
#include <chrono>
#include <algorithm>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#define KK_VERBOSE(...) do { printf(__VA_ARGS__); fflush(stdout); } while (0)

// Measurement build (-DKK_ABLATE, tools/ only): ablation bits reach the dense-row kernels through an extra argument; in the
// product build the argument does not exist and every KK_DBG(bit) folds to false.
#ifdef KK_ABLATE
#define KK_DBG_PARAM , int debug
#define KK_DBG_ARG , g_spgemm.debug
#define KK_DBG(bit) ((debug & (bit)) != 0)
#else
#define KK_DBG_PARAM
#define KK_DBG_ARG
#define KK_DBG(bit) false
#endif

#ifdef KK_EMU
#define KK_ATOMIC_FADD(p, v) atomicAdd((p), (v))
#define KK_LOAD_L2(p) (*(p))
#else
#define KK_ATOMIC_FADD(p, v) unsafeAtomicAdd((p), (v))   // hardware global_atomic_add_f64 / ds_add_f64
// agent-scope relaxed load: global_load ... sc1, served by L2 (where the atomics were performed), never by a stale L1 line
#define KK_LOAD_L2(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

namespace kk {

constexpr int kNumBins   = 5;      // 0 empty, 1 wave, 2 block-small, 3 block-large, 4 dense
constexpr int kHashMul   = 107;
constexpr int kFlopsLong = 512;    // rows of A above this many entries get a workgroup in the row-flops pass
constexpr int kSymWaveTable = 2048;   // symbolic keys only: 8 KB per wave
constexpr int kWaveTable = 512;       // numeric keys + values
constexpr int kSymBlkS = 4096,  kSymBlkL = 16384;   // 64 KB of keys: two workgroups per CU (32768 slots: one; R-MAT s20 symbolic 101 -> 94 ms)
constexpr int kNumBlkS = 4096,  kNumBlkL = 8192;
constexpr int kDenseBlock = 1024;     // dense-row column kernel: 16 waves around one LDS bitmap
constexpr int kValBlock   = 512;      // dense-row value kernel
constexpr int kValTable   = 4096;     // value-window hash table (16 KB keys + 32 KB fp64 sums in LDS: 3 workgroups per CU)
constexpr int kValCap     = kValTable / 2;   // C entries per value window
constexpr int kValLa      = 512;      // A entries of a row whose B cursors live in LDS (10 KB)
constexpr int kValTableSmall = 2048;  // ... of the light shape for rows with few entries (8 KB + 16 KB) ...
constexpr int kValLaSmall = 256;      // ... and its lists per pass (256 work-items)
constexpr int kValTableTiny = 1024, kValLaTiny = 128;    // the lightest shape: 128 work-items (19 KB of LDS: eight workgroups per CU)
constexpr int kValLa2     = 1024;     // ... of the flat value kernel's second shape (1024 work-items, one workgroup per CU)
constexpr int kValLong    = 128;      // B rows at least this long are streamed by a whole wave
constexpr int kHubLa      = 4096;     // A rows up to this long keep cursor + next column in LDS (hub value kernel)

constexpr int kUnitBitsMaxKnob = 18;  // default of the knob spgemm_unit_bits
constexpr int kItemMaxBlocks = 4;     // column blocks per rank item (default of the knob spgemm_item_blocks)

struct SpgemmTuning {
  int win_bits       = 1 << 20;   // columns per LDS bitmap window (128 KB); rows wider than this take several passes
  int val_cap        = kValCap;   // C entries per value window
  int force_unsorted = 0;         // test hook: treat B as unsorted (dense rows accumulate in HBM)
  int emit_chunked   = 0;         // 1 = every dense row walks its bitmap in contiguous runs (the sparse-bitmap path) when emitting entries(C)
#ifdef KK_ABLATE
  int debug          = 0;         // measurement build only: ablation bits for the dense-row kernels
#endif
  int val_la         = kValLa;    // A rows up to this long use the cached-cursor value kernel (<= kValLa)
  int val_shape      = 0;         // value-kernel geometry: 0 = 4096 slots x 512 threads (default), 1..4 alternatives
  int emit_win_bits  = 0;         // numeric: bitmap window of the rows whose bitmap was not kept (0 = win_bits)
  int hub_split      = 1;         // hub rows: one workgroup per pass of kHubLa entries (sums of rows with several passes meet through fp64 atomics)
  int sym_large      = 0;         // symbolic: rows of 2049..8192 products through the 16384-slot hash kernel; 0 (default) = the bitmap kernel (R-MAT scale 20: the hash kernel spent 21 ms on 137 K such rows, symbolic 78 -> 68 ms without it)
  int keep_bitmaps   = 1;         // symbolic keeps the bitmaps of its densest rows for the first numeric call (0 = every row walks its products twice)
  int emit_staged    = 1;         // entries(C) of the stored bitmaps leave through wave-private LDS (whole-line stores); 0 = every lane writes its own run
  int keep_lists     = 1;         // ... and the entry lists of the other dense rows, in a pool behind the bitmaps (0 = those rows walk their products twice)
  int val_la2        = kValLa2;   // ... up to this many entries (above kValLa2: several passes of kValLa2 lists)
  int val_small_cnt  = 65536;     // rows of C with at most this many entries (and at most kValLaSmall entries in the A row) take the flat value kernel's light shape (0 = none;
                                  // R-MAT scale 20 numeric / reuse: 0: 221.7 / 184.7 ms, 8192: 217.6 / 180.7, 32768: 213.8 / 176.6, 65536: 211.1 / 174.6, 131072: 216.4 / 179.5, all: 219.0 / 182.3)
  int val_tiny_cnt   = 32768;     // ... and at most this many (and kValLaTiny entries in the A row) the lightest one: 128 work-items, 1024-slot table (0 = none;
                                  // R-MAT scale 20 numeric / reuse with the light shape at 65536: 0: 212.0 / 175.2 ms, 2048: 210.0 / 172.8, 8192: 208.4 / 171.6, 32768: 207.5 / 170.9)
  int quad_rows      = 1;         // wave-per-row kernels with four rows of the list per wave, 16 lanes each: 1 = for the rows of the wave bin with at most kQuadFlops products /
                                  // kQuadNnz entries (the bin's list is split when it mixes sizes; 7-pt FD 150^3: symbolic 3.54 -> 1.94 ms, numeric 4.24 -> 1.97), 2 = for every row of the
                                  // bin (waves with a larger row do their four rows one after the other: 27-pt FE 100^3 numeric 4.04 -> 5.03 ms, which is why 1 is the default), 0 = never
  int emit_sort      = 1;         // entries(C) of the dense-bin rows with at most kEmitSortCap products: sorted in LDS, 256 work-items per row (0 = the bitmap kernel)
  int val_mid        = 1;         // A rows of kValLa + 1 .. kValLa2 entries through the flat value kernel's 1024-list shape (0 = the hub kernel)
  int hub_chunked    = 1;         // A rows above kHubLa entries: 1 = the LDS hub value kernel in passes of kHubLa entries, 0 = L2 atomics into a k-wide HBM accumulator
  int col_quads      = 4;         // dense-row bitmap kernels read entries(B) as aligned 16-byte quads, 4 or 8 per work-item and step (0 = one 4-byte load per product)
  int val_hub_flat   = 0;         // 1 = A rows above kValLa through the flat value kernel too (measured slower, see numeric_typed)
  int val_kernel     = 2;         // dense rows with short A rows: 2 = flat walk with the lists cut per window group (default), 1 = wave-per-list streaming
  int block          = 1;         // rows of C that are dense (or have more lists than the flat kernel's shapes) through the column-block value kernel (0 = windows only)
  int block_w        = 16384;     // its columns per block (a power of two; 16384 = 128 KB of fp64 sums)
  int items          = 1;         // the column-block class as ITEMS (groups of blocks with a position-indexed accumulator, dense blocks direct); 0 = one workgroup per (row, block)
  int item_blocks    = kItemMaxBlocks;   // column blocks per rank item at most (4 x 16384 columns = 16 KB of packed words)
  int item_cap       = 6144;      // entries of C per rank item (48 KB of sums; with the 16 KB of packed words two workgroups per CU)
  int block_min_pct  = 4;         // ... rows with at least this percentage of the columns (R-MAT scale 20 reuse, items: 20 %: 109.0 ms, 9 %: 101.1, 6 %: 98.1, 4 %: 96.7, 3 %: 96.7, 2 %: 99.4, 1 %: 107.7; one workgroup per (row, block): 6 %: 136.5, 12 %: 117.1, 20 %: 115.9) ...
  int block_la_pct   = 3;         // ... or at least this percentage and more than kValLa lists
  int list_staged    = 1;         // symbolic: the entry lists kept for the numeric phase are written wave by wave, 64 consecutive words per round (0 = every lane writes its own run)
  int nt             = 0;         // value kernels of the dense rows: entries(C) / values(C) through nontemporal loads / stores
  int sort_rows      = 1;         // the row lists of the dense kernels are ordered by size, largest first (0 = the order the binning left)
  int sym_units      = 1;         // symbolic phase of the dense class by units (row, window of 2^unit_bits columns): spgemm_sym_unit_kernel; 0 = one workgroup per row (spgemm_dense_cols_kernel)
  int unit_bits      = kUnitBitsMaxKnob;   // log2 of a unit's columns (6 .. 18; 18 = 32 KB of bitmap, four workgroups of 256 per CU)
  int store_cap_mb   = 0;         // upper bound, in MB, on the store of kept structure (bitmaps / entry lists of the symbolic phase) a product may take; 0 = none beyond the
                                  // share of the free HBM the library takes by itself (0.225).  A host that interleaves its own allocations sets it; rows past it walk their products again
  int pool_keep      = 0;         // the process-wide store of bitmaps / entry lists when the last handle is destroyed: 0 = returned to the device after the process's
                                  // first product, kept once the process has come back for it (see BmPool); 1 = always kept; 2 = always returned
};
static SpgemmTuning g_spgemm;

struct BinLimits { int64_t lim[kNumBins - 1]; };   // size <= lim[b] -> bin b  (lim[0] = 0)
static const BinLimits kSymLimits = {{0, (kSymWaveTable * 2) / 3, kSymBlkS / 2, kSymBlkL / 2}};
static const BinLimits kSymLimitsNoLarge = {{0, (kSymWaveTable * 2) / 3, kSymBlkS / 2, kSymBlkS / 2}};   // knob sym_large 0: rows above 2048 products straight to the bitmap kernel
static const BinLimits kSymLimitsC = {{0, (1024 * 2) / 3, 4096 / 2, 16384 / 2}};     // compressed symbolic (keys + masks: smaller tables)
static const BinLimits kAllDense = {{0, 0, 0, 0}};                                   // every non-empty row in the last bin
static const BinLimits kNumLimits = {{0, kWaveTable / 2, kNumBlkS / 2, (kNumBlkL * 2) / 3}};
// B sorted: everything above the small block table goes to the column + windowed value kernels (no 8192-slot bitonic sort)
static const BinLimits kNumLimitsSorted = {{0, kWaveTable / 2, kWaveTable / 2, kWaveTable / 2}};

__host__ __device__ __forceinline__ int bin_of(int64_t size, const BinLimits& L) {
  if (size <= L.lim[0]) return 0;
  if (size <= L.lim[1]) return 1;
  if (size <= L.lim[2]) return 2;
  if (size <= L.lim[3]) return 3;
  return 4;
}

struct BinOffsets { int64_t off[kNumBins + 1]; };

// ------------------------------------------------------------------------------------------------
// 1. row flops (K8 analogue, sparse/impl/KokkosSparse_spgemm_impl_symbolic.hpp:1108-1185): 8 lanes per row.
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_flops_kernel(int64_t m, const OffT* __restrict__ rmA,
                                                              const int32_t* __restrict__ entA,
                                                              const OffT* __restrict__ rmB, int64_t* __restrict__ flops,
                                                              unsigned long long* __restrict__ stats /*[0]=total,[1]=max*/,
                                                              int32_t* __restrict__ long_list, unsigned long long* __restrict__ long_cnt,
                                                              const OffT* __restrict__ endB = nullptr) {
  __shared__ unsigned long long s_sum, s_max;
  if (threadIdx.x == 0) { s_sum = 0; s_max = 0; }
  __syncthreads();
  const int lane = threadIdx.x & 7;
  long long sum = 0, mx = 0;      // per work-item partials over the grid-stride loop: two global atomics per workgroup
  const int64_t stride = (int64_t)gridDim.x * (kBlock / 8);
  for (int64_t r0 = (int64_t)blockIdx.x * (kBlock / 8); r0 < m; r0 += stride) {     // workgroup-uniform trip count
    const int64_t row = r0 + threadIdx.x / 8;
    long long f = 0;
    const bool mine = row < m && (int64_t)rmA[row + 1] - (int64_t)rmA[row] <= kFlopsLong;     // longer rows: spgemm_flops_long_kernel, from this list
    if (row < m && !mine && lane == 0) long_list[atomicAdd(long_cnt, 1ull)] = (int32_t)row;
    if (mine) {
      // four entries per lane and step, their loads independent (R-MAT scale 20: the longest row, 39,580 entries on these 8 lanes, was
      // 5 of the kernel's 7 ms with one entry per step; rows above kFlopsLong entries now get a whole workgroup)
      const int64_t a_end = (int64_t)rmA[row + 1];
      for (int64_t a = (int64_t)rmA[row] + lane; a < a_end; a += 32) {
        int32_t c[4];
        KK_UNROLL
        for (int u = 0; u < 4; ++u) c[u] = a + 8 * u < a_end ? entA[a + 8 * u] : -1;
        KK_UNROLL
        for (int u = 0; u < 4; ++u) if (c[u] >= 0) f += (long long)(endB ? endB[c[u]] : rmB[c[u] + 1]) - (long long)rmB[c[u]];
      }
    }
    f = group_sum(f, 8);
    if (mine && lane == 0) { flops[row] = f; sum += f; mx = f > mx ? f : mx; }
  }
  sum = group_sum(sum, 64);
  for (int o = 32; o > 0; o >>= 1) { const long long other = __shfl_xor(mx, o, 64); mx = other > mx ? other : mx; }
  if ((threadIdx.x & 63) == 0) {
    if (sum) atomicAdd(&s_sum, (unsigned long long)sum);
    if (mx) atomicMax(&s_max, (unsigned long long)mx);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_sum) atomicAdd(&stats[0], s_sum);
    if (s_max) atomicMax(&stats[1], s_max);
  }
}

// rows of A above kFlopsLong entries (at most nnz(A) / kFlopsLong of them: the launch's size), from the list spgemm_flops_kernel left: a
// workgroup per row.  (Every workgroup looking at kBlock rows and walking the long ones among them put the long rows of a graph, which sit
// together -- R-MAT: at the indices with few bits set, so that rows a power of two apart are no better than consecutive ones --, into a few
// workgroups: 1.0 - 1.3 ms on R-MAT scale 20.)
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_flops_long_kernel(const int32_t* __restrict__ long_list, const unsigned long long* __restrict__ long_cnt,
                                                                   const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                   const OffT* __restrict__ rmB, int64_t* __restrict__ flops,
                                                                   unsigned long long* __restrict__ stats, const OffT* __restrict__ endB = nullptr) {
  __shared__ unsigned long long s_part[kBlock / 64];
  if ((unsigned long long)blockIdx.x >= *long_cnt) return;                       // (uniform)
  const int64_t row = long_list[blockIdx.x], b = (int64_t)rmA[row], e = (int64_t)rmA[row + 1];
  long long f = 0;
  // eight entries per work-item and step, their loads independent (an entry is two dependent trips to memory)
  for (int64_t a = b + threadIdx.x; a < e; a += 8 * kBlock) {
    int32_t c[8];
    KK_UNROLL
    for (int u = 0; u < 8; ++u) c[u] = a + u * kBlock < e ? entA[a + u * kBlock] : -1;
    KK_UNROLL
    for (int u = 0; u < 8; ++u) if (c[u] >= 0) f += (long long)(endB ? endB[c[u]] : rmB[c[u] + 1]) - (long long)rmB[c[u]];
  }
  f = group_sum(f, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = (unsigned long long)f;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long tot = 0;
    for (int w = 0; w < kBlock / 64; ++w) tot += s_part[w];
    flops[row] = (int64_t)tot;
    atomicAdd(&stats[0], tot); atomicMax(&stats[1], tot);
  }
}
// are the rows of a CRS graph column-sorted (non-strict)?  8 lanes per row; *unsorted is set to 1 otherwise.
// are the rows of a CRS graph column-sorted (non-strict)?  *unsorted is set to 1 otherwise.  Rows of up to kSortedLong entries: 8 lanes per
// row (stencils and the body of a graph: 0.04 ms on 7-pt 150^3); longer rows: every workgroup looks at kBlock rows and walks the long ones
// among them with all its work-items (with 8 lanes on every row the hub rows of R-MAT scale 20 made this 1.9 ms of every symbolic phase;
// streaming the entries with a search of the row map at every descent was fast there and 0.3 - 0.5 ms on stencils, whose rows end every
// 7 - 27 entries; a bitmap of the row starts cost an allocation per call).
constexpr int kSortedLong = 256;
template <class OffT>
__global__ __launch_bounds__(kBlock) void rows_sorted_kernel(int64_t n, const OffT* __restrict__ rm,
                                                             const int32_t* __restrict__ ent, int* __restrict__ unsorted) {
  const int lane = threadIdx.x & 7;
  bool bad = false;
  for (int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / 8; row < n; row += (int64_t)gridDim.x * (kBlock / 8)) {
    const int64_t b = (int64_t)rm[row], e = (int64_t)rm[row + 1];
    if (e - b > kSortedLong) continue;
    for (int64_t j = b + lane; j + 1 < e; j += 8) bad |= ent[j] > ent[j + 1];
  }
  if (bad) *unsorted = 1;
}
template <class OffT>
__global__ __launch_bounds__(kBlock) void rows_sorted_long_kernel(int64_t n, const OffT* __restrict__ rm, const int32_t* __restrict__ ent, int* __restrict__ unsorted) {
  __shared__ int s_long[kBlock];
  __shared__ int s_n;
  const int64_t mine = (int64_t)threadIdx.x * gridDim.x + blockIdx.x;       // rows gridDim.x apart (see spgemm_flops_long_kernel)
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  if (mine < n && (int64_t)rm[mine + 1] - (int64_t)rm[mine] > kSortedLong) s_long[atomicAdd(&s_n, 1)] = threadIdx.x;
  __syncthreads();
  bool bad = false;
  const int n_long = s_n;
  for (int i = 0; i < n_long; ++i) {
    const int64_t row = (int64_t)s_long[i] * gridDim.x + blockIdx.x, b = (int64_t)rm[row], e = (int64_t)rm[row + 1];
    for (int64_t j = b + threadIdx.x; j + 1 < e; j += 8 * kBlock) {          // eight independent pairs per work-item and step
      int32_t x0[8], x1[8];
      KK_UNROLL
      for (int u = 0; u < 8; ++u) { const int64_t q = j + u * kBlock; const bool in = q + 1 < e; x0[u] = in ? ent[q] : 0; x1[u] = in ? ent[q + 1] : 0; }
      KK_UNROLL
      for (int u = 0; u < 8; ++u) bad |= x0[u] > x1[u];
    }
  }
  if (bad) *unsorted = 1;
}
// sum of the per-row counts (before the scan) in 64 bits: a 32-bit row_map must not wrap silently
template <class OffT> __global__ void sum_counts_kernel(int64_t m, const OffT* __restrict__ counts, unsigned long long* out) {
  unsigned long long s = 0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += (int64_t)gridDim.x * blockDim.x) s += (unsigned long long)counts[r];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}
// ... and with the largest count (the reference's set_max_result_nnz, impl_symbolic.hpp:1501-1505): out[0] += sum, out[1] = max
template <class OffT> __global__ void sum_max_counts_kernel(int64_t m, const OffT* __restrict__ counts, unsigned long long* out) {
  unsigned long long s = 0, mx = 0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long c = (unsigned long long)counts[r];
    s += c; mx = c > mx ? c : mx;
  }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); const unsigned long long m2 = __shfl_xor(mx, o, 64); mx = m2 > mx ? m2 : mx; }
  __shared__ unsigned long long s_s[16], s_m[16];            // one pair of atomics per workgroup (16,000 waves on two addresses were 0.35 of this kernel's 0.38 ms)
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) { s_s[wave] = s; s_m[wave] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nw; ++w) { s += s_s[w]; mx = s_m[w] > mx ? s_m[w] : mx; }
    if (s) { atomicAdd(out, s); atomicMax(out + 1, mx); }
  }
}
// row size of C from its row_map (numeric binning)
template <class OffT>
__global__ void spgemm_rowsize_kernel(int64_t m, const OffT* __restrict__ rmC, int64_t* __restrict__ sizes) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += (int64_t)gridDim.x * blockDim.x)
    sizes[r] = (int64_t)rmC[r + 1] - (int64_t)rmC[r];
}

// 2. binning: count, then scatter row ids grouped by bin (workgroup-aggregated cursors).
__global__ __launch_bounds__(kBlock) void spgemm_bin_count_kernel(int64_t m, const int64_t* __restrict__ sizes,
                                                                  int64_t cap, BinLimits L,
                                                                  unsigned long long* __restrict__ counts) {
  __shared__ int s_cnt[kNumBins];
  if (threadIdx.x < kNumBins) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r < m) { const int64_t sz = sizes[r] < cap ? sizes[r] : cap; atomicAdd(&s_cnt[bin_of(sz, L)], 1); }
  __syncthreads();
  if (threadIdx.x < kNumBins && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

__global__ __launch_bounds__(kBlock) void spgemm_bin_scatter_kernel(int64_t m, const int64_t* __restrict__ sizes,
                                                                    int64_t cap, BinLimits L, BinOffsets off,
                                                                    unsigned long long* __restrict__ cursors,
                                                                    int32_t* __restrict__ perm) {
  __shared__ int s_cnt[kNumBins];
  __shared__ unsigned long long s_base[kNumBins];
  if (threadIdx.x < kNumBins) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int bin = -1, local = 0;
  if (r < m) { const int64_t sz = sizes[r] < cap ? sizes[r] : cap; bin = bin_of(sz, L); local = atomicAdd(&s_cnt[bin], 1); }
  __syncthreads();
  if (threadIdx.x < kNumBins) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&cursors[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]) : 0ull;
  __syncthreads();
  if (r < m) perm[off.off[bin] + (int64_t)s_base[bin] + local] = (int32_t)r;
}

// ------------------------------------------------------------------------------------------------
// LDS open-addressing helpers
// A plain LDS read comes first: LDS atomics retire at about one lane per clock per CU (the wave-per-row kernels were
// bound by exactly that), reads at 16+ lanes per clock, and most products hit a column that is already in the table.
__device__ __forceinline__ bool hash_insert_key(int* tab, int mask, int key) {   // true if the key is new
  int h = (int)(((unsigned)key * (unsigned)kHashMul) & (unsigned)mask);
  while (true) {
    const int k = tab[h];
    if (k == key) return false;
    if (k == -1) {
      const int old = atomicCAS(&tab[h], -1, key);
      if (old == -1) return true;
      if (old == key) return false;
    }
    h = (h + 1) & mask;
  }
}
template <class VT> __device__ __forceinline__ void hash_accumulate(int* keys, VT* vals, int mask, int key, VT v) {
  int h = (int)(((unsigned)key * (unsigned)kHashMul) & (unsigned)mask);
  while (true) {
    const int k = keys[h];
    if (k == key) { KK_ATOMIC_FADD(&vals[h], v); return; }
    if (k == -1) {
      const int old = atomicCAS(&keys[h], -1, key);
      if (old == -1 || old == key) { KK_ATOMIC_FADD(&vals[h], v); return; }
    }
    h = (h + 1) & mask;
  }
}

// Value-window tables (keys distinct, load <= 1/2, so an empty slot always ends a probe sequence): Fibonacci hashing on
// the high bits and a probe loop with no trip counter -- the wave-wide loop runs as long as its slowest lane, so every
// instruction in it counts.
template <int H> __device__ __forceinline__ int vt_hash(int key) {
  constexpr int kBits = H == 1024 ? 10 : (H == 2048 ? 11 : (H == 4096 ? 12 : 13));
  static_assert(H == 1024 || H == 2048 || H == 4096 || H == 8192, "value table size");
  return (int)(((unsigned)key * 0x9E3779B1u) >> (32 - kBits));
}
template <int H> __device__ __forceinline__ int vt_insert(int* hk, int key) {          // returns the slot
  int hh = vt_hash<H>(key);
  while (atomicCAS(&hk[hh], -1, key) != -1) hh = (hh + 1) & (H - 1);
  return hh;
}
template <int H> __device__ __forceinline__ int vt_find(const int* hk, int key) {       // slot, or -1 if absent
  int hh = vt_hash<H>(key);
  int kq = hk[hh];
  while (kq != key && kq != -1) { hh = (hh + 1) & (H - 1); kq = hk[hh]; }
  return kq == key ? hh : -1;
}

// Visit every product column of A(row,:)*B with `nthreads` cooperating work-items (id tid): sub-groups of 2^s lanes
// share an A entry and stride over that B row.  s is at least sg_log2 (the matrix-wide hint: average B row length) and
// grows for rows of A with few entries so that the sub-groups (nthreads >> s of them) just cover the row -- a row with
// 13 entries handled by 1024 work-items runs 16 sub-groups of 64 lanes instead of leaving 115 of 128 idle.
// Each lane issues kProdUnroll independent B loads per step (a step is latency-bound otherwise): f(a, j, column).
#ifndef KK_PROD_UNROLL
#define KK_PROD_UNROLL 4
#endif
constexpr int kProdUnroll = KK_PROD_UNROLL;
template <class OffT, class F>
__device__ __forceinline__ void for_each_product(int64_t row, const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                 const OffT* __restrict__ rmB, const int32_t* __restrict__ entB, int tid,
                                                 int nthreads, int sg_log2, F f) {
  const int64_t a_beg = (int64_t)rmA[row], a_end = (int64_t)rmA[row + 1];
  while ((int64_t)(nthreads >> (sg_log2 + 1)) >= a_end - a_beg && (2 << sg_log2) <= nthreads) ++sg_log2;
  const int sg = 1 << sg_log2, sub = tid >> sg_log2, nsub = nthreads >> sg_log2, sl = tid & (sg - 1);
  for (int64_t a = a_beg + sub; a < a_end; a += nsub) {
    const int32_t c    = entA[a];
    const int64_t b_end = (int64_t)rmB[c + 1];
    for (int64_t j = (int64_t)rmB[c] + sl; j < b_end; j += (int64_t)sg * kProdUnroll) {
      int col[kProdUnroll];
      KK_UNROLL
      for (int u = 0; u < kProdUnroll; ++u) { const int64_t ju = j + (int64_t)u * sg; col[u] = ju < b_end ? entB[ju] : -1; }
      KK_UNROLL
      for (int u = 0; u < kProdUnroll; ++u) if (col[u] >= 0) f(a, j + (int64_t)u * sg, col[u]);
    }
  }
}
// same with the B value loaded alongside the column: f(a, column, value of B)
template <class OffT, class VT, class F>
__device__ __forceinline__ void for_each_product_v(int64_t row, const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                   const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                                   const VT* __restrict__ valB, int tid, int nthreads, int sg_log2, F f) {
  const int64_t a_beg = (int64_t)rmA[row], a_end = (int64_t)rmA[row + 1];
  while ((int64_t)(nthreads >> (sg_log2 + 1)) >= a_end - a_beg && (2 << sg_log2) <= nthreads) ++sg_log2;
  const int sg = 1 << sg_log2, sub = tid >> sg_log2, nsub = nthreads >> sg_log2, sl = tid & (sg - 1);
  for (int64_t a = a_beg + sub; a < a_end; a += nsub) {
    const int32_t c    = entA[a];
    const int64_t b_end = (int64_t)rmB[c + 1];
    for (int64_t j = (int64_t)rmB[c] + sl; j < b_end; j += (int64_t)sg * kProdUnroll) {
      int col[kProdUnroll];
      VT bv[kProdUnroll];
      KK_UNROLL
      for (int u = 0; u < kProdUnroll; ++u) {
        const int64_t ju = j + (int64_t)u * sg;
        const bool ok    = ju < b_end;
        col[u] = ok ? entB[ju] : -1;
        bv[u]  = ok ? valB[ju] : VT(0);
      }
      KK_UNROLL
      for (int u = 0; u < kProdUnroll; ++u) if (col[u] >= 0) f(a, col[u], bv[u]);
    }
  }
}

// Workgroup-wide FLAT iteration over the products of one row (block-per-row kernels).  The row's A entries are taken
// NT at a time: every work-item looks up one B row (start, length), a workgroup scan turns the lengths into product
// offsets, and then product q of the chunk is handled by work-item q mod NT -- consecutive lanes read consecutive B
// entries, all kProdUnroll loads of a lane are independent, and no lane waits on the entries(A) -> row_map(B) ->
// entries(B) chain more than once per chunk.  (With sub-groups walking "their" B rows the kernels were bound by that
// chain: ~135 G products/s on R-MAT; see DESIGN.md.)  A work-item takes kProdUnroll NEIGHBOURING products per step: one
// binary search for the first, a short walk for the rest (with product q on work-item q mod NT every product paid a full
// search, and the searches were most of the instructions of the R-MAT symbolic kernels).
// Every work-item of the workgroup must call it.  f(a, j, column) with a, j indices into A's / B's entry arrays.
template <int NT> struct FlatScratch {
  long long pre[NT + 1];    // product offset of each A entry of the chunk
  long long b0[NT];         // first B entry of each
  long long wave[NT / 64];
};
struct NoVals {};
template <int NT, class OffT, class VT, class F>
__device__ __forceinline__ void flat_products_impl(int64_t row, const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                   const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                                   const VT* __restrict__ valB, FlatScratch<NT>& sc, F f,
                                                   const OffT* __restrict__ endB = nullptr) {
  constexpr bool kVals = !std::is_same<VT, NoVals>::value;
  const int t = threadIdx.x;
  const int64_t a_beg = (int64_t)rmA[row], a_end = (int64_t)rmA[row + 1];
  for (int64_t chunk = a_beg; chunk < a_end; chunk += NT) {
    const int n = (int)(a_end - chunk < NT ? a_end - chunk : NT);
    long long len = 0, b0 = 0;
    if (t < n) { const int32_t c = entA[chunk + t]; b0 = (long long)rmB[c]; len = (long long)(endB ? endB[c] : rmB[c + 1]) - b0; }
    long long tot;
    const long long excl = block_exclusive_scan_n<long long, NT>(len, &tot, sc.wave);
    if (t < n) { sc.pre[t] = excl; sc.b0[t] = b0; }
    if (t == 0) sc.pre[n] = tot;
    __syncthreads();
    auto find = [&](long long q, int lo) {      // largest s in [lo, n) with pre[s] <= q
      int len2 = n - lo;
      while (len2 > 1) { const int half = len2 >> 1; lo += (sc.pre[lo + half] <= q) ? half : 0; len2 -= half; }
      return lo;
    };
    for (long long base = 0; base < tot; base += (long long)NT * kProdUnroll) {
      // work-item t takes products base + t U .. + U - 1: ONE search per work-item and step, then a short walk (the products of
      // a lane are neighbours, mostly of one B row); across the lanes of a load the addresses are U entries apart
      const long long q0 = base + (long long)t * kProdUnroll;
      int sgc = q0 < tot ? find(q0, 0) : 0;
      int col[kProdUnroll], seg[kProdUnroll];
      long long jj[kProdUnroll];
      typename std::conditional<kVals, VT, int>::type bv[kProdUnroll];
      KK_UNROLL
      for (int u = 0; u < kProdUnroll; ++u) {
        const long long q = q0 + u;
        col[u] = -1; seg[u] = 0; jj[u] = 0; bv[u] = 0;
        if (q < tot) {
          while (q >= sc.pre[sgc + 1]) ++sgc;            // also steps over empty B rows; q < tot = pre[n] ends it
          seg[u] = sgc;
          jj[u]  = sc.b0[sgc] + (q - sc.pre[sgc]);
        }
      }
      // UNCONDITIONAL loads (a product past the end reads entry 0 of B, which exists when tot > 0): the compiler branches around a
      // conditional load and then waits for every load in flight before it issues the next one -- U round trips per step, not one
      KK_UNROLL
      for (int u = 0; u < kProdUnroll; ++u) {
        col[u] = entB[jj[u]];
        if constexpr (kVals) bv[u] = valB[jj[u]];
      }
      KK_UNROLL
      for (int u = 0; u < kProdUnroll; ++u)
        if (q0 + u < tot) {
          if constexpr (kVals) f(chunk + seg[u], col[u], bv[u]);
          else f(chunk + seg[u], (int64_t)jj[u], col[u]);
        }
    }
    __syncthreads();
  }
}
// f(a, j, column)
template <int NT, class OffT, class F>
__device__ __forceinline__ void flat_products(int64_t row, const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                              const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                              FlatScratch<NT>& sc, F f, const OffT* __restrict__ endB = nullptr) {
  flat_products_impl<NT, OffT, NoVals>(row, rmA, entA, rmB, entB, (const NoVals*)nullptr, sc, f, endB);
}
// f(a, column, value of B)
template <int NT, class OffT, class VT, class F>
__device__ __forceinline__ void flat_products_v(int64_t row, const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                                const VT* __restrict__ valB, FlatScratch<NT>& sc, F f) {
  flat_products_impl<NT, OffT, VT>(row, rmA, entA, rmB, entB, valB, sc, f);
}

// Columns only, 16 bytes at a time (the dense-row bitmap kernels: symbolic count and entries(C)).  The flat walk above issues one
// 4-byte load per product; the bitmap kernels are bound by how many loads a work-item keeps in flight (16 waves per CU around one
// 128 KB bitmap: R-MAT scale 20, 2.0e10 products in 57 ms = 350 G products/s, 1.4 TB/s), not by bytes.  Here the unit of work is
// an ALIGNED quad of entries(B): a B row [b0, b1) is covered by the quads (b0 >> 2) .. ((b1 + 3) >> 2) - 1, a scan of the quad
// counts lets work-item q take Q neighbouring quads per step -- one 16-byte load each, four times the products per load
// instruction -- and the lanes of a quad that lie before b0 (they belong to the previous row) or past b1 are masked.  The quad at
// the end of a row is read entry by entry (a 16-byte load there could reach past the end of the array).  f(column).
template <int NT> struct FlatScratchQ {
  int pre[NT + 2];          // quad offset of each A entry of the chunk (32-bit: a chunk whose quads do not fit takes the plain loop below)
  long long b0[NT], b1[NT]; // its B row
  long long wave[NT / 64];
};
template <int NT, int Q, class OffT, class F>
__device__ __forceinline__ void flat_columns_quads(int64_t row, const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                   const OffT* __restrict__ rmB, const int32_t* __restrict__ entB, int64_t nnzB,
                                                   FlatScratchQ<NT>& sc, F f) {
  const int t = threadIdx.x;
  const int64_t a_beg = (int64_t)rmA[row], a_end = (int64_t)rmA[row + 1];
  // Every 16-byte load of the walk is UNCONDITIONAL: a load the compiler has to branch around makes it wait for all loads in flight
  // before it issues the next (the Q quads of a step then cost Q memory round trips instead of one -- how this walk ran until round 4:
  // 2, 4 and 8 quads per step all measured the same).  A quad that does not exist reads quad 0; the one quad that would reach past
  // the end of the array (nnz(B) not a multiple of 4: its last 1..3 entries) reads the last full quad instead and takes its entries
  // from `tail`, loaded once.  nnz(B) >= 4 (the caller's condition for this walk).
  // The walk's own arithmetic is 32-bit (quad offsets inside the chunk, positions inside a quad): the kernel around it issues 0.8
  // vector instructions per product on R-MAT and keeps the vector unit half busy, so instructions count.
  const long long last_full = ((nnzB >> 2) - 1) << 2;              // first entry of the last quad that lies inside the array
  int tail[3];
  KK_UNROLL
  for (int e = 0; e < 3; ++e) { const long long i = last_full + 4 + e; tail[e] = entB[i < nnzB ? i : nnzB - 1]; }
  for (int64_t chunk = a_beg; chunk < a_end; chunk += NT) {
    const int n = (int)(a_end - chunk < NT ? a_end - chunk : NT);
    long long nq = 0, b0 = 0, b1 = 0;
    if (t < n) {
      const int32_t c = entA[chunk + t];
      b0 = (long long)rmB[c]; b1 = (long long)rmB[c + 1];
      if (b1 > b0) nq = ((b1 + 3) >> 2) - (b0 >> 2);
    }
    long long tot64;
    const long long excl = block_exclusive_scan_n<long long, NT>(nq, &tot64, sc.wave);
    if (tot64 > (long long)INT_MAX - 2 * NT * Q) {       // (an A row that names rows of B with 8e9 entries between them: not a real case)
      for (int a = 0; a < n; ++a) {
        const int32_t c = entA[chunk + a];
        for (long long j = (long long)rmB[c] + t; j < (long long)rmB[c + 1]; j += NT) f(entB[j]);
      }
      __syncthreads();
      continue;
    }
    const int tot = (int)tot64;
    if (t < n) { sc.pre[t] = (int)excl; sc.b0[t] = b0; sc.b1[t] = b1; }
    if (t == 0) sc.pre[n] = tot;
    __syncthreads();
    auto find = [&](int q) {            // largest s in [0, n) with pre[s] <= q
      int lo = 0, len2 = n;
      while (len2 > 1) { const int half = len2 >> 1; lo += (sc.pre[lo + half] <= q) ? half : 0; len2 -= half; }
      return lo;
    };
    // (Measured and not kept: two steps in flight -- the quads of step s + 1 located and requested before those of step s are consumed --
    // changed nothing, R-MAT scale 20 symbolic 71.1 -> 71.2 ms; eight quads per step instead of four: 72.3 ms.)
    auto locate = [&](int base, long long (&at)[Q], int (&lo)[Q], int (&hi)[Q]) {
      KK_UNROLL
      for (int u = 0; u < Q; ++u) { at[u] = 0; lo[u] = 0; hi[u] = 0; }                  // no quad: nothing between lo and hi
      if (base >= tot) return;                                                         // uniform
      const int q0 = base + t * Q;
      int seg = q0 < tot ? find(q0) : 0;
      // the B row of the current quad stays in registers: neighbouring quads are mostly of one row (LDS is read when the row changes)
      int pre_next = sc.pre[seg + 1];
      long long sb0 = sc.b0[seg], sb1 = sc.b1[seg];
      long long qbase = ((sb0 >> 2) - sc.pre[seg]) << 2;  // quad q of this row starts at entry qbase + 4 q
      KK_UNROLL
      for (int u = 0; u < Q; ++u) {
        const int q = q0 + u;
        if (q < tot) {
          if (q >= pre_next) {
            do { ++seg; pre_next = sc.pre[seg + 1]; } while (q >= pre_next);     // also steps over empty B rows; q < tot = pre[n] ends it
            sb0 = sc.b0[seg]; sb1 = sc.b1[seg]; qbase = ((sb0 >> 2) - sc.pre[seg]) << 2;
          }
          at[u] = qbase + ((long long)q << 2);
          lo[u] = sb0 > at[u] ? (int)(sb0 - at[u]) : 0;
          hi[u] = sb1 < at[u] + 4 ? (int)(sb1 - at[u]) : 4;
        }
      }
    };
    auto request = [&](const long long (&at)[Q], int4 (&v)[Q]) {
      KK_UNROLL
      for (int u = 0; u < Q; ++u) v[u] = *reinterpret_cast<const int4*>(entB + (at[u] <= last_full ? at[u] : last_full));
    };
    auto consume = [&](const long long (&at)[Q], const int (&lo)[Q], const int (&hi)[Q], int4 (&v)[Q]) {
#ifndef KK_EMU
      // the quads stay whole: left alone, the compiler splits one of them and sinks the 4-byte load of its first entry into the branch
      // that consumes it -- one more round trip per step
      KK_UNROLL
      for (int u = 0; u < Q; ++u) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
#endif
      KK_UNROLL
      for (int u = 0; u < Q; ++u) {
        const bool past = at[u] > last_full;              // the array's last, partial quad
        const int col[4] = {past ? tail[0] : v[u].x, past ? tail[1] : v[u].y, past ? tail[2] : v[u].z, v[u].w};
        KK_UNROLL
        for (int e = 0; e < 4; ++e) if (e >= lo[u] && e < hi[u]) f(col[e]);
      }
    };
    for (int base = 0; base < tot; base += NT * Q) {
      long long at[Q];
      int lo[Q], hi[Q];
      int4 v[Q];
      locate(base, at, lo, hi); request(at, v); consume(at, lo, hi, v);
    }
    __syncthreads();
  }
}

// Inclusive prefix sum of one int per lane across the wave, on the VECTOR unit: six DPP adds (row_shr 1, 2, 4, 8, then lane 15 / lane 31
// of the rows before broadcast into the rows after).  __shfl_up compiles to ds_bpermute_b32, which occupies the LDS unit: the bitmap
// emission kernel issued 700 of them per workgroup pass and kept the LDS unit 84 % busy (profiles/round4/spgemm_s20_sq_counters.txt).
__device__ __forceinline__ int wave_inclusive_scan_i32(int v, int lane) {
#ifdef KK_EMU
  for (int o = 1; o < 64; o <<= 1) { const int nb = __shfl_up(v, (unsigned)o, 64); if (lane >= o) v += nb; }
  return v;
#else
  (void)lane;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1 (zeros shifted in at the start of every row of 16)
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return v;
#endif
}
__device__ __forceinline__ int wave_last_lane_i32(int v) {
#ifdef KK_EMU
  return __shfl(v, 63, 64);
#else
  return __builtin_amdgcn_readlane(v, 63);
#endif
}
__device__ __forceinline__ int wave_first_lane_i32(int v) {
#ifdef KK_EMU
  return __shfl(v, 0, 64);
#else
  return __builtin_amdgcn_readlane(v, 0);
#endif
}
__device__ __forceinline__ int wave_sum_i32(int v, int lane) { return wave_last_lane_i32(wave_inclusive_scan_i32(v, lane)); }
// Inclusive prefix sum inside every ROW of 16 lanes (a DPP row): four adds on the vector unit.  All 64 lanes call it.
__device__ __forceinline__ int row16_inclusive_scan_i32(int v, int lane) {
#ifdef KK_EMU
  for (int o = 1; o < 16; o <<= 1) { const int nb = __shfl_up(v, (unsigned)o, 16); if ((lane & 15) >= o) v += nb; }
  return v;
#else
  (void)lane;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
  return v;
#endif
}
// FOUR ROWS PER WAVE (rows with a few dozen products: a 7-point stencil times itself has 49, the A P of a multigrid set-up 7 .. 100):
// lanes 16 g .. 16 g + 15 walk the products of the g-th row.  The control flow is the WAVE's -- every loop runs as long as the longest
// of the four rows needs (__any), lanes with nothing to do are masked -- so that the prefix sums and ballots inside stay wave-wide
// operations.  f as in wave_flat_products.  sc: the group's own scratch.
struct Group16Scratch { int pre[17]; int pad_[3]; long long b0[16]; };
template <class OffT, class VT, class F>
__device__ __forceinline__ void group16_flat_products(bool active, int64_t row, const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                      const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                                      const VT* __restrict__ valB, int lane, Group16Scratch& sc, F f) {
  constexpr bool kVals = !std::is_same<VT, NoVals>::value;
  constexpr int U = kProdUnroll;
  const int gl = lane & 15;
  int64_t a_beg = 0, a_end = 0;
  if (active) { a_beg = (int64_t)rmA[row]; a_end = (int64_t)rmA[row + 1]; }
  for (int64_t off = 0; __any(a_beg + off < a_end); off += 16) {
    const int64_t chunk = a_beg + off;
    const int n = chunk < a_end ? (int)(a_end - chunk < 16 ? a_end - chunk : 16) : 0;
    long long b0 = 0;
    int len = 0;
    if (gl < n) { const int32_t c = entA[chunk + gl]; b0 = (long long)rmB[c]; len = (int)((long long)rmB[c + 1] - b0); }
    const int inc = row16_inclusive_scan_i32(len, lane);
    KK_WAVE_SYNC();                       // the previous chunk's readers are done
    sc.pre[gl] = inc - len; sc.b0[gl] = b0;
    if (gl == 15) sc.pre[16] = inc;
    KK_WAVE_SYNC();
    const int tot = sc.pre[16];
    for (int base = 0; __any(base < tot); base += 16 * U) {
      int col[U], seg[U];
      long long jj[U];
      typename std::conditional<kVals, VT, int>::type bv[U];
      KK_UNROLL
      for (int u = 0; u < U; ++u) {
        const int q = base + u * 16 + gl;
        seg[u] = 0; jj[u] = 0;
        if (q < tot) {                    // (tot > 0 implies n >= 1)
          int lo = 0, len2 = n;
          while (len2 > 1) { const int half = len2 >> 1; lo += (sc.pre[lo + half] <= q) ? half : 0; len2 -= half; }
          seg[u] = lo;
          jj[u] = sc.b0[lo] + (q - sc.pre[lo]);
        }
      }
      KK_UNROLL
      for (int u = 0; u < U; ++u) {       // unconditional loads (entry 0 of B exists when some group of the wave has products)
        col[u] = entB[jj[u]];
        if constexpr (kVals) bv[u] = valB[jj[u]];
      }
      KK_UNROLL
      for (int u = 0; u < U; ++u)
        if (base + u * 16 + gl < tot) {
          if constexpr (kVals) f(chunk + seg[u], col[u], bv[u]);
          else f(chunk + seg[u], (int64_t)jj[u], col[u]);
        }
    }
  }
}
// Wave-wide flat iteration (wave-per-row kernels): the same idea as flat_products for one wave -- lane l looks up the
// l-th A entry of the row (all B row lookups of up to 64 entries in ONE dependent chain instead of one chain per group
// of entries), a shuffle scan turns the lengths into product offsets kept in wave-private LDS, then lane (q mod 64)
// takes product q.  Every lane of the wave must call it (uniform trip counts; `active` masks idle waves).
struct WaveFlatScratch { int pre[65]; long long b0[64]; };
template <class OffT, class VT, class F>
__device__ __forceinline__ void wave_flat_products(bool active, int64_t row, const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                   const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                                   const VT* __restrict__ valB, int lane, WaveFlatScratch& sc, F f,
                                                   const OffT* __restrict__ endB = nullptr) {
  constexpr bool kVals = !std::is_same<VT, NoVals>::value;
  int64_t a_beg = 0, a_end = 0;
  if (active) { a_beg = (int64_t)rmA[row]; a_end = (int64_t)rmA[row + 1]; }
  for (int64_t chunk = a_beg; chunk < a_end; chunk += 64) {
    const int n = (int)(a_end - chunk < 64 ? a_end - chunk : 64);
    long long b0 = 0;
    int len = 0;
    if (lane < n) { const int32_t c = entA[chunk + lane]; b0 = (long long)rmB[c]; len = (int)((long long)(endB ? endB[c] : rmB[c + 1]) - b0); }
    const int inc = wave_inclusive_scan_i32(len, lane);      // (on the vector unit: a shuffle is an LDS instruction, and the wave-per-row
    const int tot = wave_last_lane_i32(inc);                 //  kernels keep the LDS unit 50-70 % busy on stencil products)
    KK_WAVE_SYNC();                       // the previous chunk's readers are done
    sc.pre[lane] = inc - len; sc.b0[lane] = b0;
    if (lane == 0) sc.pre[64] = tot;
    KK_WAVE_SYNC();
    auto find = [&](int q, int lo) {      // largest s in [lo, n) with pre[s] <= q
      int len2 = n - lo;
      while (len2 > 1) { const int half = len2 >> 1; lo += (sc.pre[lo + half] <= q) ? half : 0; len2 -= half; }
      return lo;
    };
    for (int base = 0; base < tot; base += 64 * kProdUnroll) {
      int col[kProdUnroll], seg[kProdUnroll];
      long long jj[kProdUnroll];
      typename std::conditional<kVals, VT, int>::type bv[kProdUnroll];
      KK_UNROLL
      for (int u = 0; u < kProdUnroll; ++u) {
        const int q = base + u * 64 + lane;
        seg[u] = u == 0 ? 0 : seg[u - 1]; jj[u] = 0;
        if (q < tot) {
          seg[u] = find(q, seg[u]);
          jj[u] = sc.b0[seg[u]] + (q - sc.pre[seg[u]]);
        }
      }
      KK_UNROLL
      for (int u = 0; u < kProdUnroll; ++u) {          // unconditional loads (see flat_products_impl): entry 0 of B exists when tot > 0
        col[u] = entB[jj[u]];
        if constexpr (kVals) bv[u] = valB[jj[u]];
      }
      KK_UNROLL
      for (int u = 0; u < kProdUnroll; ++u)
        if (base + u * 64 + lane < tot) {
          if constexpr (kVals) f(chunk + seg[u], col[u], bv[u]);
          else f(chunk + seg[u], (int64_t)jj[u], col[u]);
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// B compression for the symbolic phase (K9 analogue: sparse/impl/KokkosSparse_spgemm_impl_compression.hpp:400-636; layout note
// SURVEY A5).  Columns are grouped into SETS of 32 consecutive columns: set index = column >> 5, mask bit = column & 31.  Row
// i's (set, mask) pairs are written from the ORIGINAL start row_map(B)[i]; end[i] marks where they stop.  On matrices whose
// rows hold runs of neighbouring columns (stencils: three neighbours in x per grid line) the symbolic phase then inserts a
// third as many keys; on matrices without such runs (R-MAT) compression buys nothing and is dropped: it is kept only when
// it removes at least 15 % of the symbolic work (sparse/impl/KokkosSparse_spgemm_impl_def.hpp:81-127, cut-off 0.85).
// Needs column-sorted rows of B (a set is a run); the masks are OR-ed with atomics into a zeroed array, so lanes may split a
// row anywhere.  8 lanes per row, each a contiguous piece.
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_compress_kernel(int64_t n, const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                                                 int32_t* __restrict__ setB, unsigned* __restrict__ maskB,
                                                                 OffT* __restrict__ endB) {
  const int lane = threadIdx.x & 7;
  for (int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / 8; row < n; row += (int64_t)gridDim.x * (kBlock / 8)) {   // uniform per 8-lane group
    const int64_t b = (int64_t)rmB[row], e = (int64_t)rmB[row + 1], len = e - b;
    const int64_t per = (len + 7) / 8;
    const int64_t j0 = b + lane * per, j1 = (j0 + per < e) ? j0 + per : e;
    int starts = 0;
    for (int64_t j = j0; j < j1; ++j) starts += (j == b || (entB[j] >> 5) != (entB[j - 1] >> 5)) ? 1 : 0;
    int incl = starts;                                          // inclusive scan over the row's 8 lanes
    for (int o = 1; o < 8; o <<= 1) { const int v = __shfl_up(incl, (unsigned)o, 8); if (lane >= o) incl += v; }
    const int total = __shfl(incl, 7, 8);
    int64_t pos = b + (incl - starts) - 1;                      // position of the set the piece's first entry belongs to (if it continues one)
    for (int64_t j = j0; j < j1; ++j) {
      const int c = entB[j];
      if (j == b || (c >> 5) != (entB[j - 1] >> 5)) { ++pos; setB[pos] = c >> 5; }
      atomicOr(&maskB[pos], 1u << (c & 31));
    }
    if (lane == 0) endB[row] = (OffT)(b + total);
  }
}

// keys = set indices, masks OR-ed per key; returns how many mask bits this call added (the row's nnz is their sum: no pass
// over the table afterwards).  A plain read filters the common case of bits that are already there.
__device__ __forceinline__ int hash_insert_or(int* keys, unsigned* masks, int mask, int key, unsigned bits) {
  int h = (int)(((unsigned)key * (unsigned)kHashMul) & (unsigned)mask);
  while (true) {
    const int k = keys[h];
    bool mine = (k == key);
    if (k == -1) { const int old = atomicCAS(&keys[h], -1, key); mine = (old == -1 || old == key); }
    if (mine) {
      if ((masks[h] & bits) == bits) return 0;
      return __popc(bits & ~atomicOr(&masks[h], bits));
    }
    h = (h + 1) & mask;
  }
}
constexpr int kSymWaveTableC = 1024;       // compressed symbolic: keys + masks, 8 KB per wave
constexpr int kSymBlkSC = 4096, kSymBlkLC = 16384;
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_symc_wave_kernel(int64_t nbin, const int32_t* __restrict__ perm,
                                                                  const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                  const OffT* __restrict__ rmB, const OffT* __restrict__ endB,
                                                                  const int32_t* __restrict__ setB, const unsigned* __restrict__ maskB,
                                                                  OffT* __restrict__ counts) {
  constexpr int H = kSymWaveTableC;
  __shared__ int tab[kBlock / 64][H];
  __shared__ unsigned msk[kBlock / 64][H];
  __shared__ WaveFlatScratch s_wf[kBlock / 64];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int64_t idx = (int64_t)blockIdx.x * (kBlock / 64) + w;
  for (int i = lane; i < H; i += 64) { tab[w][i] = -1; msk[w][i] = 0u; }
  __syncthreads();
  const bool active = idx < nbin;
  const int64_t row = active ? (int64_t)perm[idx] : 0;
  int* mytab = tab[w]; unsigned* mymsk = msk[w];
  int cnt = 0;
  wave_flat_products<OffT, NoVals>(active, row, rmA, entA, rmB, setB, (const NoVals*)nullptr, lane, s_wf[w],
                                   [&](int64_t, int64_t j, int c) { cnt += hash_insert_or(mytab, mymsk, H - 1, c, maskB[j]); }, endB);
  cnt = group_sum(cnt, 64);
  if (active && lane == 0) counts[row] = (OffT)cnt;
}
template <class OffT, int H, int NT>
__global__ __launch_bounds__(NT) void spgemm_symc_block_kernel(int64_t nbin, const int32_t* __restrict__ perm,
                                                                   const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                   const OffT* __restrict__ rmB, const OffT* __restrict__ endB,
                                                                   const int32_t* __restrict__ setB, const unsigned* __restrict__ maskB,
                                                                   OffT* __restrict__ counts) {
  KK_DYN_SMEM(int, dyn);                     // [H keys][H masks]
  int* tab = dyn; unsigned* msk = reinterpret_cast<unsigned*>(dyn + H);
  __shared__ int s_count;
  __shared__ FlatScratch<NT> s_flat;
  const int t = threadIdx.x;
  const int64_t row = perm[blockIdx.x];
  for (int i = t; i < H; i += NT) { tab[i] = -1; msk[i] = 0u; }
  if (t == 0) s_count = 0;
  __syncthreads();
  int cnt = 0;
  flat_products<NT, OffT>(row, rmA, entA, rmB, setB, s_flat,
                          [&](int64_t, int64_t j, int c) { cnt += hash_insert_or(tab, msk, H - 1, c, maskB[j]); }, endB);
  cnt = group_sum(cnt, 64);
  if ((t & 63) == 0 && cnt) atomicAdd(&s_count, cnt);
  __syncthreads();
  if (t == 0) counts[row] = (OffT)s_count;
  (void)nbin;
}

// ------------------------------------------------------------------------------------------------
// 3. symbolic kernels
// The table of a row is as large as the row needs (a power of two of at least 3 times its products, 64 .. kSymWaveTable slots):
// a 7-point stencil times itself has 49 products per row, and clearing 2048 slots for them was most of what the row cost the LDS unit
// (3.4 M rows of 7-pt 150^3: LDS 73 % busy, 55 LDS instructions per row of which 32 cleared the table).
template <class OffT>
__device__ __forceinline__ void sym_wave_row(bool active, int64_t row, const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                             const OffT* __restrict__ rmB, const int32_t* __restrict__ entB, OffT* __restrict__ counts,
                                             const int64_t* __restrict__ flops, int* mytab, WaveFlatScratch& wf, int lane) {
  constexpr int H = kSymWaveTable;
  int hrow = H;
  if (flops) {                                             // wave-uniform
    const int64_t need = active ? 3 * flops[row] : 0;       // (1.5 x: uniform random 1e6 x 20, 400 products per row, symbolic 3.6 -> 4.3 ms -- probe collisions)
    hrow = 64;
    while (hrow < H && (int64_t)hrow < need) hrow <<= 1;
  }
  {
    int4* t4 = reinterpret_cast<int4*>(mytab);
    int4 empty; empty.x = -1; empty.y = -1; empty.z = -1; empty.w = -1;
    for (int i = lane; i < hrow / 4; i += 64) t4[i] = empty;
  }
  KK_WAVE_SYNC();                                          // the table is the wave's own: no workgroup barrier
  int cnt = 0;
  const int mask = hrow - 1;
  wave_flat_products<OffT, NoVals>(active, row, rmA, entA, rmB, entB, (const NoVals*)nullptr, lane, wf,
                                   [&](int64_t, int64_t, int c) { cnt += hash_insert_key(mytab, mask, c) ? 1 : 0; });
  cnt = wave_sum_i32(cnt, lane);
  if (active && lane == 0) counts[row] = (OffT)cnt;
}
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_sym_wave_kernel(int64_t nbin, const int32_t* __restrict__ perm,
                                                                 const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                 const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                                                 OffT* __restrict__ counts, const int64_t* __restrict__ flops) {
  __shared__ __attribute__((aligned(16))) int tab[kBlock / 64][kSymWaveTable];
  __shared__ WaveFlatScratch s_wf[kBlock / 64];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int64_t idx = (int64_t)blockIdx.x * (kBlock / 64) + w;
  const bool active = idx < nbin;
  const int64_t row = active ? (int64_t)perm[idx] : 0;
  sym_wave_row<OffT>(active, row, rmA, entA, rmB, entB, counts, flops, tab[w], s_wf[w], lane);
}
// A wave takes FOUR consecutive rows of the bin's list.  When none of them has more than kQuadFlops products, 16 lanes take each
// (group16_flat_products, a 256-slot table per row inside the wave's table); otherwise the wave does the four rows one after the other
// as above.
constexpr int kQuadFlops = 64;
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_sym_quad_kernel(int64_t nbin, const int32_t* __restrict__ perm,
                                                                 const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                 const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                                                 OffT* __restrict__ counts, const int64_t* __restrict__ flops) {
  __shared__ __attribute__((aligned(16))) int tab[kBlock / 64][kSymWaveTable];
  __shared__ WaveFlatScratch s_wf[kBlock / 64];
  __shared__ Group16Scratch s_g[kBlock / 16];
  static_assert(kSymWaveTable >= 4 * 256, "four 256-slot tables inside a wave's table");
  const int t = threadIdx.x, w = t >> 6, lane = t & 63, g = lane >> 4, gl = lane & 15;
  const int64_t first = ((int64_t)blockIdx.x * (kBlock / 64) + w) * 4;         // the wave's first position in the list
  const bool act = first + g < nbin;
  const int64_t row = act ? (int64_t)perm[first + g] : 0;
  const int64_t fl = act ? flops[row] : 0;
  if (!__any(fl > (int64_t)kQuadFlops)) {                  // wave-uniform
    int* gtab = tab[w] + g * 256;
    {
      int4* t4 = reinterpret_cast<int4*>(gtab);
      int4 empty; empty.x = -1; empty.y = -1; empty.z = -1; empty.w = -1;
      for (int i = gl; i < 64; i += 16) t4[i] = empty;
    }
    KK_WAVE_SYNC();
    int cnt = 0;
    group16_flat_products<OffT, NoVals>(act, row, rmA, entA, rmB, entB, (const NoVals*)nullptr, lane, s_g[t >> 4],
                                        [&](int64_t, int64_t, int c) { cnt += hash_insert_key(gtab, 255, c) ? 1 : 0; });
    cnt = row16_inclusive_scan_i32(cnt, lane);
    if (act && gl == 15) counts[row] = (OffT)cnt;
  } else {
    for (int r = 0; r < 4; ++r) {
      const bool a = first + r < nbin;
      const int64_t rw = a ? (int64_t)perm[first + r] : 0;
      sym_wave_row<OffT>(a, rw, rmA, entA, rmB, entB, counts, flops, tab[w], s_wf[w], lane);
      KK_WAVE_SYNC();                                      // the next row clears the table
    }
  }
}

template <class OffT, int H, int NT>
__global__ __launch_bounds__(NT) void spgemm_sym_block_kernel(int64_t nbin, const int32_t* __restrict__ perm,
                                                                  const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                  const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                                                  OffT* __restrict__ counts, int sg_log2) {
  __shared__ int tab[H];
  __shared__ int s_count;
  __shared__ FlatScratch<NT> s_flat;
  const int t = threadIdx.x;
  const int64_t row = perm[blockIdx.x];
  for (int i = t; i < H; i += NT) tab[i] = -1;
  if (t == 0) s_count = 0;
  __syncthreads();
  int cnt = 0;
  flat_products<NT, OffT>(row, rmA, entA, rmB, entB, s_flat,
                          [&](int64_t, int64_t, int c) { cnt += hash_insert_key(tab, H - 1, c) ? 1 : 0; });
  (void)sg_log2;
  cnt = group_sum(cnt, 64);
  if ((t & 63) == 0 && cnt) atomicAdd(&s_count, cnt);
  __syncthreads();
  if (t == 0) counts[row] = (OffT)s_count;
  (void)nbin;
}

// dense rows, columns: one workgroup of 16 waves per row around a k-bit bitmap in LDS (up to 2^20 columns = 128 KB
// per pass; wider products take ceil(k / win_bits) passes over the row's products).  Columns are set with ds_or_b64,
// then the touched word range is walked 1024 words at a time: popcount, workgroup scan, and -- when EMIT -- the set
// bits are written out in ascending order, so entries(C) for the row leave the kernel column-sorted.  EMIT = false is
// the symbolic count; EMIT = true fills entries(C) in the numeric phase (the value kernel below needs them).
typedef unsigned long long kk_u64;
// The set bits of bm[0 .. words) (LDS or HBM), as columns col0 + bit index in ascending order, to entC[pos0 ...]; returns their number.
// Every wave owns a contiguous range of the words and walks it 64 words at a time (lane l owns word base + l): the loads of a wave
// are 512 contiguous bytes, neighbouring lanes write neighbouring pieces of entries(C), the offsets inside a wave come from shuffles
// and only the 16 wave totals cross the workgroup (one barrier).  (One contiguous run of words per work-item made every store of a
// wave 64 separate short runs: 4-byte stores into 64 different sectors; 1024 interleaved words per step cost sixteen workgroup scans
// per row.)  words <= 16384 (2^20 columns).  All kDenseBlock work-items call it; s_wave must be free (a barrier since its last use).
__device__ __forceinline__ int emit_bits_by_wave(const kk_u64* __restrict__ bm, int words, int64_t col0, int64_t pos0, int32_t* __restrict__ entC, int* s_wave) {
  constexpr int NW = kDenseBlock / 64, NB = 16;            // 2^20 columns / 64 / 1024 = 16 steps of 64 words per wave
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wpw = ((words + NW - 1) / NW + 63) & ~63;       // words per wave, a multiple of 64
  const int w0 = wave * wpw, w1 = (w0 + wpw < words) ? w0 + wpw : words;
  kk_u64 w[NB];
  int wsum = 0;
  KK_UNROLL
  for (int i = 0; i < NB; ++i) { const int wd = w0 + i * 64 + lane; w[i] = (i * 64 < wpw && wd < w1) ? bm[wd] : 0ull; wsum += __popcll(w[i]); }
  for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o, 64);
  if (lane == 0) s_wave[wave] = wsum;
  __syncthreads();
  int tot = 0;
  for (int i = 0; i < NW; ++i) { if (i == wave) pos0 += tot; tot += s_wave[i]; }
  KK_UNROLL
  for (int i = 0; i < NB; ++i) {
    if (i * 64 >= wpw) break;                              // uniform
    kk_u64 v = w[i];
    const int pc = __popcll(v);
    int inc = pc;
    for (int o = 1; o < 64; o <<= 1) { const int nb_ = __shfl_up(inc, (unsigned)o, 64); if (lane >= o) inc += nb_; }
    int64_t pos = pos0 + (inc - pc);
    const int64_t c0 = col0 + (int64_t)(w0 + i * 64 + lane) * 64;
    while (v) {
      const int bit = __ffsll(v) - 1;
      entC[pos++]   = (int32_t)(c0 + bit);
      v &= v - 1;
    }
    pos0 += __shfl(inc, 63, 64);
  }
  return tot;
}
// The same through wave-private LDS: a wave's 64 words are taken 16 at a time -- lane l owns the l-th 16-bit piece of the 16 words, so
// the pieces ascend with the lanes --, the pieces' columns are laid down in order in 4 KB of LDS, and the (at most 1024) entries leave
// with store instructions whose 64 lanes write 64 CONSECUTIVE entries.  With the direct form every lane writes its own run of up to
// 64 entries, i.e. every store instruction is 64 four-byte pieces in 64 different lines: on R-MAT scale 20 the stored bitmaps emit
// 8e9 entries = 8e9 L2 write requests in 27.9 ms -- the L2's request rate, not its bandwidth (32 GB at 1.15 TB/s).
// (Measured and not kept: every lane laying down the set bits of its OWN 64-bit word -- no shuffles at all, one prefix sum per 64 words,
// steps above 1024 entries taken in halves or quarters of the lanes: R-MAT scale 20 numeric 193.7 -> 200.9 ms; the 16-bit pieces below
// keep the divergent loop at 16 trips at most.)
template <int NT = kDenseBlock>
__device__ __forceinline__ int emit_bits_by_wave_staged(const kk_u64* __restrict__ bm, int words, int64_t col0, int64_t pos0, int32_t* __restrict__ entC, int* s_wave,
                                                        int32_t* __restrict__ stage /* [1024] of this wave */) {
  constexpr int NW = NT / 64, NB = 16;                     // words <= 1024 NW (2^20 columns for 1024 work-items, 2^18 for 256)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wpw = ((words + NW - 1) / NW + 63) & ~63;
  const int w0 = wave * wpw, w1 = (w0 + wpw < words) ? w0 + wpw : words;
  kk_u64 w[NB];
  int wsum = 0;
  KK_UNROLL
  for (int i = 0; i < NB; ++i) { const int wd = w0 + i * 64 + lane; w[i] = (i * 64 < wpw && wd < w1) ? bm[wd] : 0ull; wsum += __popcll(w[i]); }
  for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o, 64);
  if (lane == 0) s_wave[wave] = wsum;
  __syncthreads();
  int tot = 0;
  for (int i = 0; i < NW; ++i) { if (i == wave) pos0 += tot; tot += s_wave[i]; }
  KK_UNROLL
  for (int i = 0; i < NB; ++i) {
    if (i * 64 >= wpw) break;                              // uniform
    if (__ballot(w[i] != 0ull) == 0ull) continue;          // 64 empty words (uniform)
    for (int sub = 0; sub < 4; ++sub) {                    // 16 words = 64 pieces of 16 bits
      const int src = sub * 16 + (lane >> 2);
      const unsigned lo = __shfl((unsigned)w[i], src, 64), hi = __shfl((unsigned)(w[i] >> 32), src, 64);
      const unsigned half = (lane & 2) ? hi : lo;
      unsigned piece = (lane & 1) ? (half >> 16) : (half & 0xffffu);
      const int pc = __popc(piece);
      const int inc = wave_inclusive_scan_i32(pc, lane);
      const int total = wave_last_lane_i32(inc);
      if (total == 0) continue;                            // uniform
      int at = inc - pc;
      const int c0 = (int)(col0 + ((int64_t)(w0 + i * 64 + src)) * 64 + 16 * (lane & 3));
      while (piece) { const int bit = __ffs((int)piece) - 1; stage[at++] = c0 + bit; piece &= piece - 1; }
      KK_WAVE_SYNC();
      for (int q = lane; q < total; q += 64) entC[pos0 + q] = stage[q];
      KK_WAVE_SYNC();
      pos0 += total;
    }
  }
  return tot;
}
// The same for the bitmaps of UNITS (row of C, window of columns), which are sparse: a unit keeps a bitmap when its PRODUCTS could fill one,
// and on R-MAT a unit's 262144 bits hold 7000 entries on average -- a 16-bit piece holds 0.4, and the four rounds of pieces per 64 words
// (two lane exchanges, a prefix sum, a store of a few dozen entries each) cost 0.86 vector instructions per entry over the whole kernel, 58 % of
// the vector unit's time.  Here a lane lays down the set bits of its OWN word: one prefix sum and one store round per 64 words; only
// rounds that hold more than the wave's 1024 staging slots (every fourth bit set) go by pieces.
// The kernel's code must stay small: the 16 rounds of a wave are unrolled (the words sit in registers), and with the piece form inlined in
// every round the kernel was 124 KB of instructions and 126 registers -- twice the instruction cache two CUs share, four waves per SIMD.
// So the unrolled rounds hold the sparse form only; a pair of rounds with more than 1024 entries is noted (its place in entries(C) in LDS)
// and taken afterwards by a loop that exists once and reads the pair's words again.
template <int NT>
__device__ __forceinline__ int emit_unit_bits_by_wave(const kk_u64* __restrict__ bm, int words, int64_t col0, int64_t pos0, int32_t* __restrict__ entC, int* s_wave,
                                                      int32_t* __restrict__ stage /* [1024] of this wave */, int* __restrict__ later /* [16] of this wave */) {
  constexpr int NW = NT / 64, NB = 16;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wpw = ((words + NW - 1) / NW + 63) & ~63;
  const int w0 = wave * wpw, w1 = (w0 + wpw < words) ? w0 + wpw : words;
  kk_u64 w[NB];
  int wsum = 0;
  KK_UNROLL
  for (int i = 0; i < NB; ++i) { const int wd = w0 + i * 64 + lane; w[i] = (i * 64 < wpw && wd < w1) ? bm[wd] : 0ull; wsum += __popcll(w[i]); }
  for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o, 64);
  if (lane == 0) s_wave[wave] = wsum;
  __syncthreads();
  int tot = 0;
  for (int i = 0; i < NW; ++i) { if (i == wave) pos0 += tot; tot += s_wave[i]; }
  auto lay = [&](kk_u64 v, int at, int cb) {               // the set bits of the lane's own word, ascending, from stage[at]
    unsigned lo32 = (unsigned)v, hi32 = (unsigned)(v >> 32);
    _Pragma("nounroll") while (lo32) { stage[at++] = cb + (__ffs((int)lo32) - 1); lo32 &= lo32 - 1u; }
    _Pragma("nounroll") while (hi32) { stage[at++] = cb + 32 + (__ffs((int)hi32) - 1); hi32 &= hi32 - 1u; }
  };
  auto flush = [&](int64_t at, int n) {                    // stage[0 .. n) -> entries(C), 64 consecutive entries per store instruction
    KK_WAVE_SYNC();
    _Pragma("nounroll") for (int q = lane; q < n; q += 64) entC[at + q] = stage[q];
    KK_WAVE_SYNC();
  };
  // two rounds of 64 words share one prefix sum (their popcounts -- at most 64 per lane, 4096 per round -- packed into the halves of one int)
  unsigned dense = 0u;                                     // (uniform) pairs left for the loop below
  int rel = 0;                                             // entries of this wave before the round
  KK_UNROLL
  for (int i = 0; i < NB; i += 2) {
    if (i * 64 >= wpw) break;                              // uniform (wpw is a multiple of 64: round i + 1 may lie beyond it and is then all zero)
    const int pa = __popcll(w[i]), pb = __popcll(w[i + 1]);
    const int inc = wave_inclusive_scan_i32(pa | (pb << 16), lane);
    const int last = wave_last_lane_i32(inc);
    const int ta = last & 0xffff, tb = last >> 16;
    if (ta + tb == 0) continue;                            // 128 empty words (uniform)
    if (ta + tb <= 1024) {
      lay(w[i], (inc & 0xffff) - pa, (int)(col0 + ((int64_t)(w0 + i * 64 + lane)) * 64));
      lay(w[i + 1], ta + (inc >> 16) - pb, (int)(col0 + ((int64_t)(w0 + i * 64 + 64 + lane)) * 64));
      flush(pos0 + rel, ta + tb);
    } else {
      if (lane == 0) later[i >> 1] = rel;
      dense |= 1u << (i >> 1);
    }
    rel += ta + tb;
  }
  KK_WAVE_SYNC();
  while (dense) {                                          // (uniform)
    const int pr = __ffs((int)dense) - 1;
    dense &= dense - 1u;
    int64_t at = pos0 + later[pr];
    _Pragma("nounroll") for (int half = 0; half < 2; ++half) {
      const int wbase = w0 + pr * 128 + half * 64;
      const kk_u64 v = (wbase + lane < w1) ? bm[wbase + lane] : 0ull;
      const int pcw = __popcll(v);
      const int incw = wave_inclusive_scan_i32(pcw, lane);
      const int totw = wave_last_lane_i32(incw);
      if (totw == 0) continue;                             // uniform
      if (totw <= 1024) {
        lay(v, incw - pcw, (int)(col0 + ((int64_t)(wbase + lane)) * 64));
        flush(at, totw); at += totw;
        continue;
      }
      _Pragma("nounroll") for (int sub = 0; sub < 4; ++sub) {                  // 16 words = 64 pieces of 16 bits at a time
        const int src = sub * 16 + (lane >> 2);
        const unsigned lo = __shfl((unsigned)v, src, 64), hi = __shfl((unsigned)(v >> 32), src, 64);
        const unsigned hf = (lane & 2) ? hi : lo;
        unsigned piece = (lane & 1) ? (hf >> 16) : (hf & 0xffffu);
        const int pc = __popc(piece);
        const int inc = wave_inclusive_scan_i32(pc, lane);
        const int total = wave_last_lane_i32(inc);
        if (total == 0) continue;                          // uniform
        int a = inc - pc;
        const int c0 = (int)(col0 + ((int64_t)(wbase + src)) * 64 + 16 * (lane & 3));
        _Pragma("nounroll") while (piece) { const int bit = __ffs((int)piece) - 1; stage[a++] = c0 + bit; piece &= piece - 1; }
        flush(at, total); at += total;
      }
    }
  }
  return tot;
}
struct BitmapStore {                 // where the symbolic count kernel may leave a row's bitmap (words == 0: nowhere)
  kk_u64* words_out = nullptr;       // [cap][words]
  int32_t* row_slot = nullptr;       // [m], -1 = not stored
  unsigned long long* counter = nullptr;
  long long cap = 0, min_count = 0;
  int words = 0;
  // ... or its ENTRIES, for the rows whose bitmap is not kept (fewer than min_count entries, or no slot left): the set bits in ascending
  // order at pool[pool_off[row] ...], space taken from the cursor row by row
  int32_t* pool = nullptr;
  long long* pool_off = nullptr;     // [m], -1 = not written
  unsigned long long* pool_cursor = nullptr;
  long long pool_cap = 0;
  int list_staged = 1;               // the lists are written wave by wave, 64 consecutive words per round (0 = every lane its own run of words)
};
template <class OffT, bool EMIT, int Q = 4>
__global__ __launch_bounds__(kDenseBlock) void spgemm_dense_cols_kernel(const int32_t* __restrict__ perm,
                                                                        const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                        const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                                                        OffT* __restrict__ counts, const OffT* __restrict__ rmC,
                                                                        int32_t* __restrict__ entC, int64_t k, int win_bits,
                                                                        int sg_log2, int force_chunked, const OffT* __restrict__ endB,
                                                                        const unsigned* __restrict__ maskB, int64_t quads, BitmapStore bs KK_DBG_PARAM) {
  // quads: 0, or nnz(B) -- entries(B) is then read as aligned 16-byte quads (flat_columns_quads)
  // endB / maskB (symbolic count only): B is compressed -- entB holds set indices, a product ORs its 32-column mask into the bitmap
  KK_DYN_SMEM(kk_u64, bm);
  __shared__ int s_min, s_max;
  __shared__ int s_wave[kDenseBlock / 64];
  __shared__ FlatScratchQ<kDenseBlock> s_flatq;
  FlatScratch<kDenseBlock>& s_flat = *reinterpret_cast<FlatScratch<kDenseBlock>*>(&s_flatq);     // the compressed path's scratch: a prefix of the same memory
  static_assert(sizeof(FlatScratch<kDenseBlock>) <= sizeof(FlatScratchQ<kDenseBlock>), "scratch overlay");
  const int t       = threadIdx.x;
  const int64_t row = perm[blockIdx.x];
  int64_t total     = 0;
  int ch_a = 0, ch_z = 0, ch_excl = 0, ch_per = 0;                 // (count only) the work-item's run of words in the last window, its offset and the words per work-item
  (void)sg_log2;
  for (int64_t c0 = 0; c0 < k; c0 += win_bits) {
    const int nbits  = (int)((k - c0 < (int64_t)win_bits) ? k - c0 : (int64_t)win_bits);
    const int nwords = (nbits + 63) >> 6;
    for (int i = t; i < nwords; i += kDenseBlock) bm[i] = 0ull;
    if (t == 0) { s_min = INT_MAX; s_max = -1; }
    __syncthreads();
    int cmin = INT_MAX, cmax = -1;
    if (maskB) {
      flat_products<kDenseBlock, OffT>(row, rmA, entA, rmB, entB, s_flat, [&](int64_t, int64_t j, int sb) {
        const int64_t c64 = (int64_t)sb * 32 - c0;               // windows are multiples of 64 columns: a set never straddles one
        if (c64 >= 0 && c64 < nbits) {
          const int c = (int)c64;
          atomicOr(&bm[c >> 6], (kk_u64)maskB[j] << (c & 63));
          cmin = c < cmin ? c : cmin; cmax = (c + 31 < nbits ? c + 31 : nbits - 1) > cmax ? (c + 31 < nbits ? c + 31 : nbits - 1) : cmax;
        }
      }, endB);
    } else if (!KK_DBG(2)) {
      // 32-bit arithmetic and 32-bit LDS atomics on the halves of the 64-bit words (bit c of the window = bit c & 31 of half c >> 5)
      unsigned* bm32 = reinterpret_cast<unsigned*>(bm);
      const unsigned c0u = (unsigned)c0;                 // c0 < k <= 2^31 - 1
      auto mark = [&](int cb) {
        const unsigned cu = (unsigned)cb - c0u;          // a column before the window wraps to something above nbits
        if (cu < (unsigned)nbits) {
          const int c = (int)cu;
          if (!KK_DBG(256)) atomicOr(&bm32[cu >> 5], 1u << (cu & 31u));
          cmin = c < cmin ? c : cmin; cmax = c > cmax ? c : cmax;
        }
      };
      if (quads) flat_columns_quads<kDenseBlock, Q, OffT>(row, rmA, entA, rmB, entB, quads, s_flatq, mark);
      else flat_products<kDenseBlock, OffT>(row, rmA, entA, rmB, entB, s_flat, [&](int64_t, int64_t, int cb) { mark(cb); });
    }
    if (cmax >= 0) { atomicMin(&s_min, cmin); atomicMax(&s_max, cmax); }
    __syncthreads();
    if (s_max >= 0 && !KK_DBG(4)) {
      const int w_lo = s_min >> 6, w_hi = s_max >> 6, nw = w_hi - w_lo + 1;
      // sparse bitmaps (fewer than 4 columns per touched word on average; always when only counting): every
      // work-item takes one contiguous run of words -- one workgroup scan per window instead of one per 1024 words.
      // Dense bitmaps keep the interleaved walk, whose stores to entries(C) coalesce.  (The wave-contiguous walk of
      // emit_bits_by_wave, which serves the stored bitmaps, measured slower here on the sparse rows: R-MAT scale 20 numeric 261 -> 271 ms.)
      const bool chunked = !EMIT || force_chunked || (int64_t)rmC[row + 1] - (int64_t)rmC[row] < 4 * (int64_t)nw;
      if (chunked) {
        const int per = (nw + kDenseBlock - 1) / kDenseBlock;
        const int a = w_lo + t * per, z = (a + per <= w_hi + 1) ? a + per : w_hi + 1;
        int cnt = 0;
        for (int wd = a; wd < z; ++wd) cnt += __popcll(bm[wd]);
        int tot;
        const int excl = block_exclusive_scan_n<int, kDenseBlock>(cnt, &tot, s_wave);
        ch_a = a; ch_z = z; ch_excl = excl; ch_per = per;
        if (EMIT && !KK_DBG(1)) {
          int64_t pos = (int64_t)rmC[row] + total + excl;
          for (int wd = a; wd < z; ++wd) {
            kk_u64 v = bm[wd];
            while (v) {
              const int bit = __ffsll(v) - 1;
              entC[pos++]   = (int32_t)(c0 + (int64_t)wd * 64 + bit);
              v &= v - 1;
            }
          }
        }
        total += tot;
      } else {
        for (int wb = w_lo; wb <= w_hi; wb += kDenseBlock) {
          const int wd = wb + t;
          kk_u64 v     = (wd <= w_hi) ? bm[wd] : 0ull;
          int tot;
          const int excl = block_exclusive_scan_n<int, kDenseBlock>(__popcll(v), &tot, s_wave);
          if (EMIT && !KK_DBG(1)) {
            int64_t pos = (int64_t)rmC[row] + total + excl;
            while (v) {
              const int bit = __ffsll(v) - 1;
              entC[pos++]   = (int32_t)(c0 + (int64_t)wd * 64 + bit);
              v &= v - 1;
            }
          }
          total += tot;
        }
      }
    }
    __syncthreads();
  }
  if (!EMIT && t == 0) counts[row] = (OffT)total;
  if constexpr (!EMIT) {
    // the symbolic phase keeps the bitmaps of its densest rows (BitmapStore): the first numeric call writes their entries(C)
    // straight from them instead of walking the row's products a second time.  One window only (k <= win_bits).
    bool kept = false;
    if (bs.words && total >= bs.min_count) {
      __shared__ long long s_slot;
      if (t == 0) { const unsigned long long sl = atomicAdd(bs.counter, 1ull); s_slot = sl < (unsigned long long)bs.cap ? (long long)sl : -1; }
      __syncthreads();
      const long long slot = s_slot;
      if (slot >= 0) {
        kk_u64* dst = bs.words_out + (size_t)slot * (size_t)bs.words;
        for (int i = t; i < bs.words; i += kDenseBlock) dst[i] = bm[i];
        if (t == 0) bs.row_slot[row] = (int32_t)slot;
        kept = true;
      }
    }
    // the other rows leave their ENTRIES: the bitmap is still in LDS and every work-item knows where its run of words starts in the row
    // (ch_excl), so the first numeric call copies the list instead of walking the row's products again (R-MAT scale 20: 370 K rows, 47 ms)
    if (bs.pool && !kept && total > 0 && k <= (int64_t)win_bits) {              // workgroup-uniform
      __shared__ long long s_poff;
      if (t == 0) {
        const unsigned long long o = atomicAdd(bs.pool_cursor, (unsigned long long)total);
        s_poff = (o + (unsigned long long)total <= (unsigned long long)bs.pool_cap) ? (long long)o : -1;
      }
      __syncthreads();
      const long long poff = s_poff;
      if (poff >= 0) {
        // (staging the list in LDS -- the product walk's scratch -- and writing it out in whole lines, 6144 entries per round, measured
        // slower: dense_cols<false> 70 -> 115 ms on R-MAT scale 20.  Each work-item writes the entries of its run of words where they stand.)
        if (bs.list_staged) {
          // The count gave every work-item a contiguous run of `per` words, so a WAVE holds 64 per consecutive words and its entries
          // start at the offset of its first lane.  For the emission the wave takes them 64 at a time -- lane l the l-th word of the
          // round: consecutive LDS words (no bank conflict), a prefix sum of the popcounts on the vector unit, and the lanes' entries
          // of a round follow each other in the list: on the sparse bitmaps these rows have (one or two bits per word) the 64 lanes of
          // a store instruction write one or two lines.  (Every lane writing the entries of its own run of `per` words: 64 different
          // lines per store instruction, 17 of the symbolic phase's 62 ms on R-MAT scale 20.  Passing the wave's entries through a small
          // wave-private LDS buffer instead serialises the lanes of a wave whose words are dense: 60 -> 232 ms.)
          const int lane = t & 63;
          const int per = ch_per;
          const int wave_w0 = wave_first_lane_i32(ch_a);               // first word of the wave (the runs ascend with the lanes)
          const int w_end = s_max >> 6;                                // last touched word of the row
          int run = wave_first_lane_i32(ch_excl);
          int32_t* dst = bs.pool + poff;
          for (int i = 0; i < per; ++i) {                              // (uniform)
            const int wd = wave_w0 + i * 64 + lane;
            kk_u64 v = wd <= w_end ? bm[wd] : 0ull;
            const int pc = __popcll(v);
            const int inc = wave_inclusive_scan_i32(pc, lane);
            int pos = run + inc - pc;
            // the two 32-bit halves one after the other: five vector instructions per bit instead of twelve for the 64-bit find-first / clear
            unsigned lo32 = (unsigned)v, hi32 = (unsigned)(v >> 32);
            const int cbase = wd * 64;
            while (lo32) { dst[pos++] = cbase + (__ffs((int)lo32) - 1); lo32 &= lo32 - 1u; }
            while (hi32) { dst[pos++] = cbase + 32 + (__ffs((int)hi32) - 1); hi32 &= hi32 - 1u; }
            run += wave_last_lane_i32(inc);
          }
        } else {
          int32_t* dst = bs.pool + poff + ch_excl;
          for (int wd = ch_a; wd < ch_z; ++wd) {
            kk_u64 v = bm[wd];
            while (v) { const int bit = __ffsll(v) - 1; *dst++ = (int32_t)((int64_t)wd * 64 + bit); v &= v - 1; }
          }
        }
        if (t == 0) bs.pool_off[row] = poff;
      }
    }
  }
}

// entries(C) of a row whose bitmap the symbolic phase kept: no products, the set bits in ascending order
template <class OffT>
__global__ __launch_bounds__(kDenseBlock) void spgemm_emit_bitmap_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ row_slot,
                                                                         const kk_u64* __restrict__ store, int words, const OffT* __restrict__ rmC,
                                                                         int32_t* __restrict__ entC, int staged) {
  __shared__ int s_wave[kDenseBlock / 64];
  __shared__ int32_t s_stage[kDenseBlock / 64][1024];
  const int64_t row = perm[blockIdx.x];
  if (staged) (void)emit_bits_by_wave_staged(store + (size_t)row_slot[row] * (size_t)words, words, 0, (int64_t)rmC[row], entC, s_wave, s_stage[threadIdx.x >> 6]);
  else (void)emit_bits_by_wave(store + (size_t)row_slot[row] * (size_t)words, words, 0, (int64_t)rmC[row], entC, s_wave);
}
// rows of a bin whose size (products) reaches thr -> count[0]; count[1] = a bound on the ENTRIES of all rows of the bin (a row has at
// most as many entries as products, and at most kcols): what the entry lists kept for the numeric phase can need at most
__global__ __launch_bounds__(kBlock) void spgemm_count_ge_kernel(int64_t n, const int32_t* __restrict__ perm, const int64_t* __restrict__ sizes, int64_t thr,
                                                                int64_t kcols, unsigned long long* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t sz = i < n ? sizes[perm[i]] : 0;
  const bool q = i < n && sz >= thr;
  const kk_u64 m = __ballot(q);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned long long)__popcll(m));
  const long long bound = group_sum((long long)(sz < kcols ? sz : kcols), 64);
  if ((threadIdx.x & 63) == 0 && bound) atomicAdd(count + 1, (unsigned long long)bound);
}
// rows of the dense bin: those with a stored bitmap first, the others from the end
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_copy_pool_kernel(const int32_t* __restrict__ perm, const long long* __restrict__ pool_off, const int32_t* __restrict__ pool,
                                                                 const OffT* __restrict__ rmC, int32_t* __restrict__ entC) {
  // entries(C) of a row whose list the symbolic phase left in the pool
  const int64_t row = perm[blockIdx.x];
  const int64_t b = (int64_t)rmC[row], n = (int64_t)rmC[row + 1] - b;
  const int32_t* src = pool + pool_off[row];
  for (int64_t i = threadIdx.x; i < n; i += kBlock) entC[b + i] = src[i];
}
// ------------------------------------------------------------------------------------------------
// The dense class by UNITS (round 6).  A unit is (row of C, WINDOW of 2^wb columns): one workgroup of 256 around a 32 KB bitmap
// (wb = 18), four of them per CU, each independent of every other.  The one-workgroup-per-row kernel above (spgemm_dense_cols_kernel)
// walked all of a row's products once per window of 2^20 columns (R-MAT scale 22, k = 2^22: four times), kept one row per CU in
// flight, and spent 49 vector instructions per product at 54 % vector-ALU utilisation with nothing to overlap the per-row latency
// chains (round-5 profile, R-MAT scale 20: 50 of the symbolic phase's 59 ms).  What makes the windows cheap:
//   * an index of B at window granularity, built once per symbolic phase (spgemm_bidx_kernel): widx[w][j] = entries of row j of B
//     with a column below w 2^wb -- and from it an index of A's ENTRIES (spgemm_aw_kernel): aw[w][a] = rmB[j] + widx[w][j] for
//     j = entries(A)[a].  The piece of list a inside window w is entries(B)[aw[w][a] .. aw[w + 1][a]): a unit reads the bounds of its
//     pieces with coalesced loads and walks ONLY its own piece of every list;
//   * units without products are never launched, the others are ordered by their products (heaviest first) and described by a
//     32-byte HEAD each (row, first list, lists, where its structure goes): head -> bounds -> entries(B) are the only dependent
//     trips to memory of a unit (the first version chased perm -> row_map(A) -> entries(A) -> row_map(B) / widx -> entries(B) and
//     took room in its store by returning atomics: 9 us of latency per unit, 1.4 million units on R-MAT scale 20);
//   * every WAVE owns a contiguous range of the chunk's 16-byte quads and walks it 64 Q quads per step; the list of a step is found by
//     two wave-uniform searches (first / last quad of the step), and a step that lies inside one list and holds none of its partial end
//     quads -- nearly all products of an R-MAT row -- runs with no per-entry condition at all: one address, Q 16-byte loads, and per
//     product two shifts, a mask and one ds_or_b32.  Steps that cross lists take the general form (per-lane search inside the step's
//     lists, entries masked by their list's bounds);
//   * the columns of a piece are inside the window by construction: no range test, no minimum / maximum tracking per product;
//   * the count is a popcount of the unit's words (lane-strided LDS reads, no bank conflicts) kept in registers, from which the
//     unit's structure leaves for the first numeric call (spgemm_emit_unit_kernel): its bitmap when its products could fill one
//     (more than a bitmap's bytes / 4), else its entry list -- at an offset fixed before the launch (a prefix sum over
//     min(4 products, bitmap bytes)): no cursor, no atomics, nothing that can run full.  Per unit, so any k is covered (the per-row
//     store of rounds 3 - 5 needed k <= 2^20).
// counts[row] collects the units of a row by one integer atomic each (order-independent: the result is exact).
#ifndef KK_UQ
#define KK_UQ 4
#endif
constexpr int kUnitNT = 256;
constexpr int kUnitBitsMax = 18;                                              // 2^18 columns = 32 KB of bitmap for 256 work-items: four workgroups per CU
// (Measured and not kept: windows of 2^19 / 2^20 columns around workgroups of 512 / 1024 -- R-MAT scale 20 symbolic 39.1 / 48.7 ms against 40.0, scale 18 6.2 / 8.0
// against 5.4; the template takes the workgroup size, only 256 is instantiated.)
__host__ __device__ constexpr int unit_bits_of(int nt) { return nt >= 1024 ? 20 : (nt >= 512 ? 19 : 18); }
struct alignas(16) UnitHead {
  long long a_beg;          // first entry of the row of A
  long long store_off;      // bytes into the store, -1: the unit keeps nothing
  int n_lists;              // entries of the row of A
  int unit;                 // rank of the row in the class's list * windows + window
  int row;
  int kind;                 // 1: the unit leaves its bitmap, 0: its entry list
};
template <int NT> struct UnitScratch {
  int gpre[NT + 2];                    // offset of every list piece of the chunk in LANE-STEPS (4 aligned quads of one piece)
  int nq[NT];                          // aligned quads the piece touches
  int ends[NT];                        // entries of the piece's first quad before its start | (entries of its last quad that belong to it) << 4
  long long fq[NT];                    // the piece's first quad: entries(B) index >> 2
  long long wave64[NT / 64];
  int wave32[NT / 64];
};
template <class OffT> __device__ __forceinline__ void unit_count_add(OffT* p, int v);
template <> __device__ __forceinline__ void unit_count_add<int32_t>(int32_t* p, int v) { atomicAdd(reinterpret_cast<int*>(p), v); }
template <> __device__ __forceinline__ void unit_count_add<int64_t>(int64_t* p, int v) { atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v); }

// aw[w][a] = rmB[j] + (entries of row j of B below column w 2^wb), j = entries(A)[a]; aw[0][a] = rmB[j], aw[nwin][a] = rmB[j + 1]
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_aw_kernel(int64_t nnzA, const int32_t* __restrict__ entA, const OffT* __restrict__ rmB, int64_t nB, int nwin,
                                                          const unsigned* __restrict__ widx /* [nwin + 1][nB], nullptr when nwin == 1 */, long long* __restrict__ aw /* [nwin + 1][nnzA] */) {
  const int64_t a = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (a >= nnzA) return;
  const int32_t j = entA[a];
  const long long b = (long long)rmB[j];
  aw[a] = b;
  aw[(size_t)nwin * (size_t)nnzA + a] = (long long)rmB[j + 1];
  for (int w = 1; w < nwin; ++w) aw[(size_t)w * (size_t)nnzA + a] = b + (long long)widx[(size_t)w * (size_t)nB + j];
}
// products of every unit of the class: one wave per unit (the bounds of a window's pieces are two coalesced streams).  (One wave per ROW,
// window by window, left the class's heaviest rows -- 40,000 lists on R-MAT scale 20, first in the class's order -- to four waves: 0.9 of 1.1 ms.)
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_uprod_kernel(int64_t nrows, const int32_t* __restrict__ perm, const OffT* __restrict__ rmA, int64_t nnzA, int nwin,
                                                             const long long* __restrict__ aw, long long* __restrict__ uprod /* [nrows * nwin] */) {
  const int lane = threadIdx.x & 63;
  const int64_t u = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (u >= nrows * nwin) return;                                           // (the whole wave)
  const int64_t r = u / nwin;
  const int w = (int)(u - r * nwin);
  const int64_t row = perm[r];
  const int64_t a0 = (int64_t)rmA[row], a1 = (int64_t)rmA[row + 1];
  const long long* lo = aw + (size_t)w * (size_t)nnzA;
  const long long* hi = lo + (size_t)nnzA;
  long long s0 = 0, s1 = 0;
  int64_t a = a0 + lane;
  for (; a + 64 < a1; a += 128) { s0 += hi[a] - lo[a]; s1 += hi[a + 64] - lo[a + 64]; }
  if (a < a1) s0 += hi[a] - lo[a];
  const long long sum = group_sum(s0 + s1, 64);
  if (lane == 0) uprod[u] = sum;
}
// the units with products, in any order (they are ordered by size next); count[0] = how many
__global__ __launch_bounds__(kBlock) void spgemm_unit_compact_kernel(int64_t units, const long long* __restrict__ uprod, int32_t* __restrict__ ulist, unsigned long long* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool q = i < units && uprod[i] > 0;
  const kk_u64 mk = __ballot(q);
  const int lane = threadIdx.x & 63;
  unsigned long long start = 0;
  if (lane == 0 && mk) start = atomicAdd(count, (unsigned long long)__popcll(mk));
  start = __shfl(start, 0, 64);
  if (q) ulist[start + __popcll(mk & ((1ull << lane) - 1ull))] = (int32_t)i;
}
// bytes a unit's structure can take: its entry list (4 bytes per product at most) or, when that could exceed it, its bitmap
__device__ __forceinline__ long long unit_store_bytes(long long prod, long long bm_bytes) {
  const long long lst = (prod * 4 + 15) & ~15ll;
  return lst < bm_bytes ? lst : bm_bytes;
}
// A row's structure is of use only when ALL its units keep theirs: room is given out row by row, in the class's order (heaviest rows first).
// rbytes[r] = what row r's units take together ([nrows] = 0); after an exclusive scan, row r is kept when rbytes[r + 1] <= budget.
__global__ __launch_bounds__(kBlock) void spgemm_unit_rowbytes_kernel(int64_t nrows, int nwin, const long long* __restrict__ uprod, long long bm_bytes, int keep_lists,
                                                                     long long* __restrict__ rbytes /* [nrows + 1] */, unsigned char* __restrict__ rnone /* [nrows]: keeps nothing whatever the room */) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r > nrows) return;
  long long sum = 0;
  bool none = false;
  if (r < nrows)
    for (int w = 0; w < nwin; ++w) {
      const long long prod = uprod[r * nwin + w];
      if (prod > 0) { if (!keep_lists && prod * 4 <= bm_bytes) none = true; sum += unit_store_bytes(prod, bm_bytes); }   // (lists switched off: a row with a list unit keeps nothing)
    }
  rbytes[r] = none ? 0 : sum;
  if (r < nrows) rnone[r] = none ? 1 : 0;
}
__global__ __launch_bounds__(kBlock) void spgemm_unit_ssize_kernel(int64_t n, const int32_t* __restrict__ ulist, const long long* __restrict__ uprod, long long bm_bytes,
                                                                  int nwin, const long long* __restrict__ rscan /* exclusive scan of rbytes */, const unsigned char* __restrict__ rnone, long long budget,
                                                                  long long* __restrict__ soff /* [n + 1] */, unsigned char* __restrict__ drop /* [n]: the unit's row keeps nothing */) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    const int32_t u = ulist[i];
    const bool keep = rscan[(int64_t)(u / nwin) + 1] <= budget && !rnone[u / nwin];
    soff[i] = keep ? unit_store_bytes(uprod[u], bm_bytes) : 0; drop[i] = keep ? 0 : 1;
  }
  if (i == n) soff[i] = 0;
}
// soff: exclusive prefix of the sizes.  Heads in launch order; per unit, for the numeric phase: uoff[unit] = (store offset << 1) | kind, or -1.
// count[0] = units that leave a bitmap, count[1] = bytes of the store in use, count[2] = units without room
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_unit_heads_kernel(int64_t n, const int32_t* __restrict__ ulist, const long long* __restrict__ uprod, const long long* __restrict__ soff,
                                                                  long long bm_bytes, const unsigned char* __restrict__ drop, int nwin, const int32_t* __restrict__ perm, const OffT* __restrict__ rmA,
                                                                  UnitHead* __restrict__ heads, long long* __restrict__ uoff, unsigned long long* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  bool is_bm = false, none = false; long long used = 0;
  if (i < n) {
    const int32_t u = ulist[i];
    const long long prod = uprod[u];
    const long long sz = unit_store_bytes(prod, bm_bytes);
    const int64_t row = perm[u / nwin];
    UnitHead hd;
    hd.a_beg = (long long)rmA[row]; hd.n_lists = (int)((long long)rmA[row + 1] - hd.a_beg); hd.unit = u; hd.row = (int)row;
    hd.kind = prod * 4 > bm_bytes ? 1 : 0;
    hd.store_off = drop[i] ? -1 : soff[i];
    heads[i] = hd;
    uoff[u] = hd.store_off < 0 ? -1 : ((hd.store_off << 1) | (long long)hd.kind);
    is_bm = hd.store_off >= 0 && hd.kind == 1; none = hd.store_off < 0; used = hd.store_off >= 0 ? sz : 0;
  }
  const kk_u64 mb = __ballot(is_bm), mn = __ballot(none);
  used = group_sum(used, 64);
  if ((threadIdx.x & 63) == 0) {
    if (mb) atomicAdd(count, (unsigned long long)__popcll(mb));
    if (used) atomicAdd(count + 1, (unsigned long long)used);
    if (mn) atomicAdd(count + 2, (unsigned long long)__popcll(mn));
  }
}

template <class OffT, int NT, int Q>
__global__ __launch_bounds__(NT, 4) void spgemm_sym_unit_kernel(const UnitHead* __restrict__ heads, int nwin, int wb, int64_t k, int64_t nnzA, const long long* __restrict__ aw,
                                                             const int32_t* __restrict__ entB, int64_t nnzB, OffT* __restrict__ counts, unsigned* __restrict__ ucnt,
                                                             char* __restrict__ store, int bm_words KK_DBG_PARAM) {
  // dynamic LDS: the bitmap (bm_words 64-bit words, a multiple of 16) at offset 0 -- a bit's address is then two instructions from its column, with no
  // base to add --, the scratch behind it
  KK_DYN_SMEM(kk_u64, bm);
  UnitScratch<NT>& sc = *reinterpret_cast<UnitScratch<NT>*>(bm + bm_words);
  constexpr int NW = NT / 64, S = 64 * Q, NB = (((1 << unit_bits_of(NT)) / 64) / NW) / 64;      // 16 rounds of 64 words per wave when counting
  const int t = threadIdx.x, lane = t & 63;
  const int wave = KK_UNIFORM(t >> 6);
  const UnitHead hd = heads[blockIdx.x];
  const unsigned u = (unsigned)hd.unit;
  const unsigned w = u % (unsigned)nwin;
  const int64_t c0 = (int64_t)w << wb;
  const int nbits  = (int)(k - c0 < ((int64_t)1 << wb) ? k - c0 : ((int64_t)1 << wb));
  const int nwords = (nbits + 63) >> 6;                                   // 64-bit words of the unit's bitmap
  {
    int4 zero4; zero4.x = 0; zero4.y = 0; zero4.z = 0; zero4.w = 0;
    int4* z = reinterpret_cast<int4*>(bm);
    for (int i = t; i < bm_words >> 1; i += NT) z[i] = zero4;
  }
  __syncthreads();
  unsigned* bm32 = reinterpret_cast<unsigned*>(bm);
  const unsigned wordmask = (1u << (wb - 5)) - 1u;            // wb >= 6
  auto mark = [&](int cb) {                        // column -> bit of the window (the piece's columns are inside it): shift, mask, shift, ds_or
    if (!KK_DBG(256)) atomicOr(&bm32[((unsigned)cb >> 5) & wordmask], 1u << ((unsigned)cb & 31u));
#if defined(KK_ABLATE) && !defined(KK_EMU)
    else asm volatile("" :: "v"(cb));               // (measurement build: the column, and the load behind it, stay alive without the atomic)
#endif
  };
  const int64_t a_beg = hd.a_beg, a_end = hd.a_beg + hd.n_lists;
  const long long* awlo = aw + (size_t)w * (size_t)nnzA;
  const long long* awhi = awlo + (size_t)nnzA;
  // every 16-byte load is unconditional (see flat_columns_quads): a quad that does not exist reads quad 0, the one that would reach past
  // the end of entries(B) reads the last full quad and takes its entries from `tail`.  nnz(B) >= 4 (the caller's condition).
  const long long last_full = ((nnzB >> 2) - 1) << 2;
  int tail[3];
  KK_UNROLL
  for (int e = 0; e < 3; ++e) { const long long i = last_full + 4 + e; tail[e] = entB[i < nnzB ? i : nnzB - 1]; }
  // The unit of the walk is a LANE-STEP: four consecutive aligned quads (64 bytes) of ONE piece, taken by one lane (a piece's last lane-step is
  // padded: six entries per piece on average against pieces of 500).  A wave-step is 64 lane-steps; every wave owns a contiguous range of the
  // chunk's wave-steps.
  //   * RUNS of interior wave-steps -- all 64 lane-steps inside one piece, none holding one of its partial end quads -- are walked three
  //     steps deep with nothing searched or tested inside (where a run ends is arithmetic on the piece's bounds);
  //   * a general wave-step finds a lane's piece by one search inside the step's pieces (whose range two uniform searches gave) and masks
  //     entries only in a piece's first and last quad.
  // Lanes that set bits in one instruction are 16 entries of a sorted list apart: neighbouring lanes rarely meet in one 32-bit word.  (Measured
  // and not kept: 16 lanes on 16 consecutive quads, four such groups of quads per lane -- whole-line loads, but the lanes of an instruction
  // then sit 4 entries apart and their ds_or_b32 collide in the same words: R-MAT scale 20 symbolic 40.0 -> 44.2 ms.)
  // The first version assigned quads regardless of pieces: a lane's four quads crossed pieces, every quad carried its own piece tracking and
  // every entry two comparisons -- 34 vector instructions per product over the whole kernel at 67 % vector-ALU utilisation (R-MAT scale 20),
  // three quarters of the products in such steps.
  for (int64_t chunk = a_beg; chunk < a_end && !KK_DBG(2); chunk += NT) {
    const int n = (int)(a_end - chunk < NT ? a_end - chunk : NT);
    long long ng = 0, lo = 0, hi = 0;
    int nq = 0;
    if (t < n) {
      lo = awlo[chunk + t]; hi = awhi[chunk + t];
      if (hi > lo) {
        const long long nq64 = ((hi + 3) >> 2) - (lo >> 2);
        ng = (nq64 + 3) >> 2;
        nq = nq64 > (long long)INT_MAX ? INT_MAX : (int)nq64;
      }
    }
    long long tot64;
    const long long excl = block_exclusive_scan_n<long long, NT>(ng, &tot64, sc.wave64);
    if (tot64 > (long long)(INT_MAX >> 3)) {            // (pieces with 1e10 entries between them: not a real case) -- list by list, entry by entry
      if (t < n) sc.fq[t] = lo;
      __syncthreads();
      for (int a = 0; a < n; ++a) { const long long l0 = sc.fq[a], l1 = awhi[chunk + a]; for (long long i = l0 + t; i < l1; i += NT) mark(entB[i]); }
      __syncthreads();
      continue;
    }
    const int tot = (int)tot64;                         // lane-steps of the chunk
    if (t < n) { sc.gpre[t] = (int)excl; sc.nq[t] = nq; sc.fq[t] = lo >> 2; sc.ends[t] = (int)(lo & 3) | ((int)(((hi - 1) & 3) + 1) << 4); }
    if (t == 0) sc.gpre[n] = tot;
    __syncthreads();
    if (tot > 0) {
      constexpr int GS = 64;                            // lane-steps per wave-step
      const int nsteps = (tot + GS - 1) / GS;
      const int s_beg = (int)(((long long)nsteps * wave) / NW), s_end = (int)(((long long)nsteps * (wave + 1)) / NW);
      // largest s in [from, n) with gpre[s] <= q (pieces without quads share their successor's offset: the search steps over them)
      auto ufind = [&](int from, int q) {
        int at = from, len = n - from;
        while (len > 1) { const int half = len >> 1; at += (sc.gpre[at + half] <= q) ? half : 0; len -= half; }
        return at;
      };
      // The wave-steps run TWO DEEP: a step is looked up and its quads requested one step before its bits are set (a wave doing
      // "look up, load, wait, sixteen atomics" one step at a time had nothing outstanding most of the time: with four waves per SIMD the
      // general steps of R-MAT scale 20 ran at 1.7e12 products/s, their latency chain, whatever their instruction count).
      struct Stage { int4 v[4]; int left; int pq0; int en; };            // left: quads of the lane's piece from its first quad on (<= 0: no lane-step)
      int seg = s_beg < s_end ? KK_UNIFORM(ufind(0, s_beg * GS)) : 0;
      int seg_pre = 0, seg_next = -1, sa = 0, sb = 0;                    // of the piece `seg`: its lane-steps, its interior wave-steps [sa, sb)
      long long seg_fq = 0;
      auto issue = [&](int step, Stage& st) {                            // (uniform control flow)
        const int gw0 = step * GS;
        if (gw0 >= seg_next) {                                           // (seg_next = -1: look the piece up again)
          if (gw0 >= sc.gpre[seg + 1]) seg = KK_UNIFORM(ufind(seg, gw0));
          seg_pre = KK_UNIFORM(sc.gpre[seg]); seg_next = KK_UNIFORM(sc.gpre[seg + 1]);
          const int en_ = KK_UNIFORM(sc.ends[seg]), nqs_ = KK_UNIFORM(sc.nq[seg]);
          // interior lane-steps of the piece: [j_lo, j_hi) -- after the one with a partial first quad, before the one with a partial or missing last quad
          const int j_lo = (en_ & 15) ? 1 : 0, j_hi = (nqs_ - ((en_ >> 4) != 4 ? 1 : 0)) >> 2;
          sa = (seg_pre + j_lo + GS - 1) / GS; sb = (seg_pre + j_hi) / GS;
          seg_fq = sc.fq[seg];
        }
        long long aq0 = 0;                                               // the lane's first quad (a lane without a lane-step reads quads 0 .. 3)
        st.left = 0; st.pq0 = 0; st.en = 0x40;
        if (step >= sa && step < sb) {                                   // interior: all 64 lane-steps in this piece, every quad whole
          st.pq0 = (gw0 - seg_pre + lane) << 2; st.left = INT_MAX;
          aq0 = seg_fq + st.pq0;
        } else {
          const int gw1 = (gw0 + GS <= tot ? gw0 + GS : tot) - 1;        // the step's last lane-step
          const int seg_hi = gw1 < seg_next ? seg : KK_UNIFORM(ufind(seg, gw1));
          const int ls = gw0 + lane;
          if (ls <= gw1) {
            int sg = seg, len = seg_hi - seg + 1;                        // the lane's piece: a search inside the step's pieces
            while (len > 1) { const int half = len >> 1; sg += (sc.gpre[sg + half] <= ls) ? half : 0; len -= half; }
            st.pq0 = (ls - sc.gpre[sg]) << 2; st.en = sc.ends[sg];
            st.left = sc.nq[sg] - st.pq0;
            aq0 = sc.fq[sg] + st.pq0;
          }
          if (seg_hi != seg) { seg = seg_hi; seg_next = -1; }             // the next step looks its piece up again
        }
        // every 16-byte load is unconditional; the one quad that would reach past the end of entries(B) -- its last 1 .. 3 entries when nnz(B) is
        // no multiple of 4 -- reads the last full quad instead and takes its entries from `tail`
        const bool any_past = __any(((aq0 + 3) << 2) > last_full);
        KK_UNROLL
        for (int uu = 0; uu < 4; ++uu) st.v[uu] = reinterpret_cast<const int4*>(entB)[any_past && ((aq0 + uu) << 2) > last_full ? (last_full >> 2) : aq0 + uu];
        if (any_past) {                                                  // (uniform; once per product at most)
          KK_UNROLL
          for (int uu = 0; uu < 4; ++uu) if (((aq0 + uu) << 2) > last_full) { st.v[uu].x = tail[0]; st.v[uu].y = tail[1]; st.v[uu].z = tail[2]; }
        }
      };
      auto retire = [&](Stage& st) {
#ifndef KK_EMU
        KK_UNROLL
        for (int uu = 0; uu < 4; ++uu) asm volatile("" : "+v"(st.v[uu].x), "+v"(st.v[uu].y), "+v"(st.v[uu].z), "+v"(st.v[uu].w));     // the quads stay whole (see flat_columns_quads)
#endif
        const int left = st.left, en = st.en;
        if (left >= 4 && !(st.pq0 == 0 && (en & 15)) && !(left == 4 && (en >> 4) != 4)) {     // four whole quads
          KK_UNROLL
          for (int uu = 0; uu < 4; ++uu) { mark(st.v[uu].x); mark(st.v[uu].y); mark(st.v[uu].z); mark(st.v[uu].w); }
        } else {
          KK_UNROLL
          for (int uu = 0; uu < 4; ++uu) {
            if (uu < left) {
              const int elo = (st.pq0 + uu == 0) ? (en & 15) : 0, ehi = (uu == left - 1) ? (en >> 4) : 4;
              const int col[4] = {st.v[uu].x, st.v[uu].y, st.v[uu].z, st.v[uu].w};
              KK_UNROLL
              for (int e = 0; e < 4; ++e) if (e >= elo && e < ehi) mark(col[e]);
            }
          }
        }
      };
      Stage p0, p1;
      int nxt = s_beg, done = s_beg;
      if (nxt < s_end) issue(nxt++, p0);
      if (nxt < s_end) issue(nxt++, p1);
      while (done < s_end) {                                             // (uniform)
        retire(p0); ++done;
        if (nxt < s_end) issue(nxt++, p0);
        if (done < s_end) { retire(p1); ++done; if (nxt < s_end) issue(nxt++, p1); }
      }
    }
    __syncthreads();
  }
  __syncthreads();
  // count: every wave owns a contiguous range of the words and reads it 128 words at a time, lane l the words 2 l and 2 l + 1 (16-byte reads at
  // consecutive addresses); the words stay in registers for what follows.  (64 words per round, a word per lane: the same count, but the entry
  // lists below took a prefix sum and four loop heads per 64 words -- 6.6 of the kernel's 34 ms on R-MAT scale 20, and most of what a unit of
  // a few thousand products costs.)
  static_assert(NB % 2 == 0, "rounds of 128 words");
  constexpr int NB2 = NB / 2;
  const int wpw = ((nwords + NW - 1) / NW + 63) & ~63;
  const int w0 = wave * wpw, w1 = (w0 + wpw < nwords) ? w0 + wpw : nwords;
  kk_u64 wa[NB2], wc[NB2];
  int wsum = 0;
  KK_UNROLL
  for (int i = 0; i < NB2; ++i) {
    const int idx = w0 + i * 128 + 2 * lane;                            // (even; bm holds zeros from nwords to bm_words, a multiple of 16)
    const bool in = i * 128 < wpw && idx < w1;
    wa[i] = in ? bm[idx] : 0ull; wc[i] = in ? bm[idx + 1] : 0ull;
    wsum += __popcll(wa[i]) + __popcll(wc[i]);
  }
  wsum = wave_sum_i32(wsum, lane);
  if (lane == 0) sc.wave32[wave] = wsum;
  __syncthreads();
  int tot = 0, before = 0;
  for (int i = 0; i < NW; ++i) { const int sv = sc.wave32[i]; if (i < wave) before += sv; tot += sv; }
  if (t == 0) { ucnt[u] = (unsigned)tot; if (tot) unit_count_add<OffT>(counts + hd.row, tot); }
  if (tot == 0 || hd.store_off < 0 || KK_DBG(4)) return;                  // (uniform)
  if (hd.kind == 1) {
    kk_u64* dst = reinterpret_cast<kk_u64*>(store + hd.store_off);
    KK_UNROLL
    for (int i = 0; i < NB2; ++i) {
      const int idx = w0 + i * 128 + 2 * lane;
      if (i * 128 < wpw && idx < w1) { dst[idx] = wa[i]; if (idx + 1 < w1) dst[idx + 1] = wc[i]; }
    }
  } else {
    // the unit's entries in ascending order: a prefix sum of the lanes' popcounts on the vector unit places every lane's bits, rounds without a
    // set bit are skipped
    int32_t* dst = reinterpret_cast<int32_t*>(store + hd.store_off);
    int run = before;
    KK_UNROLL
    for (int i = 0; i < NB2; ++i) {
      if (i * 128 >= wpw) break;                                         // uniform
      const kk_u64 va = wa[i], vc = wc[i];
      if (__ballot((va | vc) != 0ull) == 0ull) continue;                 // uniform
      const int pc = __popcll(va) + __popcll(vc);
      const int inc = wave_inclusive_scan_i32(pc, lane);
      int pos = run + inc - pc;
      const int cbase = (int)(c0 + (int64_t)(w0 + i * 128 + 2 * lane) * 64);
      if (!KK_DBG(1)) {
        unsigned b0 = (unsigned)va, b1 = (unsigned)(va >> 32), b2 = (unsigned)vc, b3 = (unsigned)(vc >> 32);
        while (b0) { dst[pos++] = cbase + (__ffs((int)b0) - 1); b0 &= b0 - 1u; }
        while (b1) { dst[pos++] = cbase + 32 + (__ffs((int)b1) - 1); b1 &= b1 - 1u; }
        while (b2) { dst[pos++] = cbase + 64 + (__ffs((int)b2) - 1); b2 &= b2 - 1u; }
        while (b3) { dst[pos++] = cbase + 96 + (__ffs((int)b3) - 1); b3 &= b3 - 1u; }
      }
      run += wave_last_lane_i32(inc);
    }
  }
}

// entries(C) of a unit whose structure the symbolic phase kept (first numeric call): a copy of its list, or the set bits of its bitmap in
// ascending order through wave-private LDS (whole-line stores, emit_bits_by_wave_staged).  Units of rows the numeric phase does not
// treat as dense (at most min_nnz entries: their hash kernels write entries and values together) and of rows with a unit that kept
// nothing (unit_row < 0: the row walks its products, spgemm_dense_cols_kernel<EMIT>) return at once.
template <class OffT, int NT>
__global__ __launch_bounds__(NT) void spgemm_emit_unit_kernel(const UnitHead* __restrict__ heads, const int32_t* __restrict__ unit_row, int nwin, int wb, int64_t k,
                                                              const unsigned* __restrict__ ucnt, const unsigned* __restrict__ ucoff, const char* __restrict__ store,
                                                              const OffT* __restrict__ rmC, int32_t* __restrict__ entC, int64_t min_nnz) {
  __shared__ int s_wave[NT / 64];
  __shared__ int s_later[NT / 64][16];
  __shared__ int32_t s_stage[NT / 64][1024];
  const UnitHead hd = heads[blockIdx.x];
  const unsigned u = (unsigned)hd.unit;
  const unsigned cnt = ucnt[u];
  const unsigned coff = ucoff[u];
  const int32_t ur = unit_row[hd.row];
  const int64_t rbeg = (int64_t)rmC[hd.row], rend = (int64_t)rmC[hd.row + 1];
  if (cnt == 0 || ur < 0 || hd.store_off < 0 || rend - rbeg <= min_nnz) return;
  const int64_t base = rbeg + coff;
  if (hd.kind == 0) {
    const int32_t* src = reinterpret_cast<const int32_t*>(store + hd.store_off);
    _Pragma("unroll 4") for (unsigned i = threadIdx.x; i < cnt; i += NT) entC[base + i] = src[i];
  } else {
    const unsigned w = u % (unsigned)nwin;
    const int64_t c0 = (int64_t)w << wb;
    const int nbits = (int)(k - c0 < ((int64_t)1 << wb) ? k - c0 : ((int64_t)1 << wb));
    (void)emit_unit_bits_by_wave<NT>(reinterpret_cast<const kk_u64*>(store + hd.store_off), (nbits + 63) >> 6, c0, base, entC, s_wave, s_stage[threadIdx.x >> 6], s_later[threadIdx.x >> 6]);
  }
}
// rows of the class whose every unit with entries left its structure: unit_row[row] = rank of the row in the class's list (bit 30: at least one of
// its units left an entry list, not a bitmap); ucoff[unit] = entries of the row's units before it; count[0] = those rows.  unit_row is preset to -1.
__global__ __launch_bounds__(kBlock) void spgemm_unit_rows_kernel(int64_t nrows, const int32_t* __restrict__ perm, int nwin, const unsigned* __restrict__ ucnt,
                                                                 const long long* __restrict__ uoff, unsigned* __restrict__ ucoff, int32_t* __restrict__ unit_row,
                                                                 unsigned long long* __restrict__ count) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  bool ok = r < nrows, lists = false;
  if (ok) {
    bool any = false;
    unsigned run = 0;
    for (int w = 0; w < nwin; ++w) {
      const unsigned c = ucnt[r * nwin + w];
      const long long o = uoff[r * nwin + w];
      ucoff[r * nwin + w] = run; run += c;
      if (c) { any = true; if (o < 0) ok = false; else if ((o & 1) == 0) lists = true; }
    }
    ok = ok && any;
    if (ok) unit_row[perm[r]] = (int32_t)r | (lists ? (1 << 30) : 0);
  }
  const kk_u64 mk = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && mk) atomicAdd(count, (unsigned long long)__popcll(mk));
}
__global__ __launch_bounds__(kBlock) void spgemm_count_flag_kernel(int64_t n, const int32_t* __restrict__ list, const int32_t* __restrict__ unit_row, unsigned long long* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool q = i < n && (unit_row[list[i]] & (1 << 30)) != 0 && unit_row[list[i]] >= 0;
  const kk_u64 mk = __ballot(q);
  if ((threadIdx.x & 63) == 0 && mk) atomicAdd(count, (unsigned long long)__popcll(mk));
}

// entries(C) of a row with FEW PRODUCTS (at most kEmitSortCap, known from the symbolic phase's row flops) but more entries than the
// wave kernel's table holds: the products' columns are laid down in LDS, sorted by a bitonic network and written without their
// duplicates -- 256 work-items and 11 KB of LDS per row, so that eight rows share a CU.  The bitmap kernel gives every such row a
// workgroup of 1024 around a 128 KB bitmap, one row per CU at a time: 15 us per row whatever it holds (1e6 rows of 400 entries, the
// product of two uniform random matrices with 20 entries per row: 60 ms of a 65 ms numeric call whose values take 4 ms).
constexpr int kEmitSortCap = 2048;
__global__ __launch_bounds__(kBlock) void spgemm_flop_class_kernel(int64_t m, const int64_t* __restrict__ flops, int32_t* __restrict__ cls) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < m) cls[i] = flops[i] <= (int64_t)kEmitSortCap ? 0 : -1;
}
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_emit_sort_kernel(const int32_t* __restrict__ perm, const OffT* __restrict__ rmA,
                                                                  const int32_t* __restrict__ entA, const OffT* __restrict__ rmB,
                                                                  const int32_t* __restrict__ entB, const OffT* __restrict__ rmC,
                                                                  int32_t* __restrict__ entC) {
  constexpr int CAP = kEmitSortCap, PER = CAP / kBlock, U = 4;
  __shared__ int s_key[CAP];
  __shared__ int s_pre[kBlock + 1];
  __shared__ long long s_b0[kBlock];
  __shared__ int s_wave[kBlock / 64];
  const int t = threadIdx.x;
  const int64_t row = perm[blockIdx.x];
  const int64_t a_beg = (int64_t)rmA[row], a_end = (int64_t)rmA[row + 1];
  int filled = 0;                                                // products laid down so far (workgroup-uniform)
  for (int64_t chunk = a_beg; chunk < a_end; chunk += kBlock) {
    const int n = (int)(a_end - chunk < kBlock ? a_end - chunk : kBlock);
    int len = 0; long long b0 = 0;
    if (t < n) { const int32_t c = entA[chunk + t]; b0 = (long long)rmB[c]; len = (int)((long long)rmB[c + 1] - b0); }
    int tot;
    const int excl = block_exclusive_scan_n<int, kBlock>(len, &tot, s_wave);
    if (t < n) { s_pre[t] = excl; s_b0[t] = b0; }
    if (t == 0) s_pre[n] = tot;
    __syncthreads();
    for (int base = 0; base < tot; base += kBlock * U) {
      const int q0 = base + t * U;
      int seg = 0;
      if (q0 < tot) { int len2 = n; while (len2 > 1) { const int half = len2 >> 1; seg += (s_pre[seg + half] <= q0) ? half : 0; len2 -= half; } }
      int pre_next = s_pre[seg + 1];
      long long jb = s_b0[seg] - s_pre[seg];                      // product q of this list is entry jb + q of B
      long long jj[U];
      KK_UNROLL
      for (int u = 0; u < U; ++u) {
        const int q = q0 + u;
        jj[u] = 0;
        if (q < tot) {
          if (q >= pre_next) { do { ++seg; pre_next = s_pre[seg + 1]; } while (q >= pre_next); jb = s_b0[seg] - s_pre[seg]; }
          jj[u] = jb + q;
        }
      }
      int col[U];
      KK_UNROLL
      for (int u = 0; u < U; ++u) col[u] = entB[jj[u]];           // unconditional (entry 0 of B exists when tot > 0)
      KK_UNROLL
      for (int u = 0; u < U; ++u) if (q0 + u < tot && filled + q0 + u < CAP) s_key[filled + q0 + u] = col[u];
    }
    filled += tot;
    __syncthreads();
  }
  if (filled > CAP) filled = CAP;                                // (cannot happen: the row's flops were at most CAP)
  int N = 64;
  while (N < filled) N <<= 1;
  for (int i = filled + t; i < N; i += kBlock) s_key[i] = INT_MAX;
  __syncthreads();
  // Bitonic network.  Every wave owns a chunk of C = N / 4 (at least 64) consecutive slots: the steps whose partners are less than C apart
  // stay inside a chunk and need no workgroup barrier -- a wave's LDS operations are carried out in order --, which leaves 3 of the 45 steps of
  // 512 slots (10 of 66 at 2048) with one.  Pair i of a step is (p, p + j), p = i with a 0 inserted at bit log2 j: every work-item has a pair
  // (with q = p ^ j, q > p as the test half of them idled).  (All 45 steps behind barriers: 11.4 ms for the 10^6 rows of 400 products of
  // uniform random 10^6 x 20, whose first numeric call is this kernel.)
  {
    constexpr int NW = kBlock / 64;
    const int C = N / NW > 64 ? N / NW : 64;
    const int lane = t & 63, wave = t >> 6;
    const int cbase = wave * C;
    bool was_global = true;                                      // (the fill above ended with a workgroup barrier)
    for (int kk2 = 2; kk2 <= N; kk2 <<= 1) {
      for (int j = kk2 >> 1; j > 0; j >>= 1) {
        if (j < C) {
          if (was_global) { was_global = false; } else KK_WAVE_SYNC();
          if (cbase < N) {
            for (int i = lane; i < C / 2; i += 64) {
              const int p = cbase + (((i & ~(j - 1)) << 1) | (i & (j - 1))), q = p + j;
              const int x = s_key[p], y = s_key[q];
              const bool up = (p & kk2) == 0;
              if ((x > y) == up) { s_key[p] = y; s_key[q] = x; }
            }
          }
        } else {
          __syncthreads();
          for (int i = t; i < N / 2; i += kBlock) {
            const int p = ((i & ~(j - 1)) << 1) | (i & (j - 1)), q = p + j;
            const int x = s_key[p], y = s_key[q];
            const bool up = (p & kk2) == 0;
            if ((x > y) == up) { s_key[p] = y; s_key[q] = x; }
          }
          __syncthreads();
          was_global = true;
        }
      }
    }
    __syncthreads();
  }
  // the distinct columns, in order: work-item t looks at PER consecutive slots
  int flag[PER], cnt = 0;
  KK_UNROLL
  for (int q = 0; q < PER; ++q) {
    const int p = t * PER + q;
    flag[q] = (p < filled && (p == 0 || s_key[p] != s_key[p - 1])) ? 1 : 0;
    cnt += flag[q];
  }
  int total;
  int pos = block_exclusive_scan_n<int, kBlock>(cnt, &total, s_wave);
  const int64_t base = (int64_t)rmC[row], room = (int64_t)rmC[row + 1] - base;
  KK_UNROLL
  for (int q = 0; q < PER; ++q)
    if (flag[q]) { if (pos < room) entC[base + pos] = s_key[t * PER + q]; ++pos; }
}
template <class SlotT>
__global__ __launch_bounds__(kBlock) void spgemm_split_stored_kernel(int64_t nd, const int32_t* __restrict__ perm_in, const SlotT* __restrict__ row_slot,
                                                                    int32_t* __restrict__ perm_out, unsigned long long* __restrict__ counters /*[2]*/) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int lane  = threadIdx.x & 63;
  int32_t row = 0;
  int cls     = -1;
  if (i < nd) { row = perm_in[i]; cls = row_slot[row] >= 0 ? 0 : 1; }
  for (int c = 0; c < 2; ++c) {
    const kk_u64 m = __ballot(cls == c);
    unsigned long long start = 0;
    if (lane == 0 && m) start = atomicAdd(&counters[c], (unsigned long long)__popcll(m));
    start = __shfl(start, 0, 64);
    if (cls == c) {
      const int64_t r = (int64_t)start + __popcll(m & ((1ull << lane) - 1ull));
      perm_out[c == 0 ? r : nd - 1 - r] = row;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 4. numeric kernels
template <class OffT, class VT>
__device__ __forceinline__ void num_wave_row(bool active, int64_t row, const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                             const VT* __restrict__ valA, const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                             const VT* __restrict__ valB, const OffT* __restrict__ rmC, int32_t* __restrict__ entC,
                                             VT* __restrict__ valC, int* mk, VT* mv, int* ck, unsigned short* cs, WaveFlatScratch& wf, int lane) {
  constexpr int H = kWaveTable;
  // the row's table: a power of two of at least four times its entries (known exactly: row_map of C), 64 .. H slots -- clearing and
  // compacting 512 slots for the 25 entries of a 7-point stencil product was most of the row's LDS work
  int hrow = 64;
  {
    const int64_t need = active ? 4 * ((int64_t)rmC[row + 1] - (int64_t)rmC[row]) : 0;
    while (hrow < H && (int64_t)hrow < need) hrow <<= 1;
  }
  for (int i = lane; i < hrow; i += 64) { mk[i] = -1; mv[i] = VT(0); }
  KK_WAVE_SYNC();                                          // tables and scratch are the wave's own: no workgroup barrier
  {
    const int mask = hrow - 1;
    wave_flat_products<OffT, VT>(active, row, rmA, entA, rmB, entB, valB, lane, wf,
                                 [&](int64_t a, int c, VT bv) { hash_accumulate<VT>(mk, mv, mask, c, valA[a] * bv); });
  }
  KK_WAVE_SYNC();
  // compact the occupied slots (ballot + prefix), then rank-by-counting over the compact list only: keys are unique,
  // so rank = number of smaller keys.  ceil(n/64) * n compares per wave instead of 8 * 512 over the whole table.
  int n = 0;
  for (int s0 = 0; s0 < hrow; s0 += 64) {
    const int key        = mk[s0 + lane];
    const kk_u64 occ     = __ballot(key >= 0);
    if (key >= 0) {
      const int pos = n + __popcll(occ & ((1ull << lane) - 1ull));
      ck[pos] = key;
      cs[pos] = (unsigned short)(s0 + lane);
    }
    n += __popcll(occ);
  }
  KK_WAVE_SYNC();
  if (active) {
    const int64_t base = (int64_t)rmC[row];
    for (int i = lane; i < n; i += 64) {
      const int key = ck[i];
      int rank = 0;
      for (int q = 0; q < n; ++q) rank += (ck[q] < key) ? 1 : 0;
      entC[base + rank] = key;
      valC[base + rank] = mv[cs[i]];
    }
  }
}
template <class OffT, class VT>
__global__ __launch_bounds__(kBlock) void spgemm_num_wave_kernel(int64_t nbin, const int32_t* __restrict__ perm,
                                                                 const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                 const VT* __restrict__ valA, const OffT* __restrict__ rmB,
                                                                 const int32_t* __restrict__ entB, const VT* __restrict__ valB,
                                                                 const OffT* __restrict__ rmC, int32_t* __restrict__ entC,
                                                                 VT* __restrict__ valC, int sg_log2) {
  constexpr int H = kWaveTable;
  __shared__ __attribute__((aligned(16))) int keys[kBlock / 64][H];
  __shared__ __attribute__((aligned(16))) VT vals[kBlock / 64][H];
  __shared__ int ckey[kBlock / 64][H / 2];
  __shared__ WaveFlatScratch s_wf[kBlock / 64];
  __shared__ unsigned short cslot[kBlock / 64][H / 2];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int64_t idx = (int64_t)blockIdx.x * (kBlock / 64) + w;
  const bool active = idx < nbin;
  const int64_t row = active ? (int64_t)perm[idx] : 0;
  (void)sg_log2;
  num_wave_row<OffT, VT>(active, row, rmA, entA, valA, rmB, entB, valB, rmC, entC, valC, keys[w], vals[w], ckey[w], cslot[w], s_wf[w], lane);
}
// Four consecutive rows of the list per wave (see spgemm_sym_quad_kernel): when none of them has more than kQuadNnz entries, 16 lanes
// take each with a 128-slot key + value table inside the wave's; the rows leave sorted by the same rank-by-counting.
constexpr int kQuadNnz = 32;
template <class OffT, class VT>
__global__ __launch_bounds__(kBlock) void spgemm_num_quad_kernel(int64_t nbin, const int32_t* __restrict__ perm,
                                                                 const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                 const VT* __restrict__ valA, const OffT* __restrict__ rmB,
                                                                 const int32_t* __restrict__ entB, const VT* __restrict__ valB,
                                                                 const OffT* __restrict__ rmC, int32_t* __restrict__ entC,
                                                                 VT* __restrict__ valC) {
  constexpr int H = kWaveTable, HG = H / 4;
  static_assert(HG >= 4 * kQuadNnz, "a group's table holds four times its entries");
  __shared__ __attribute__((aligned(16))) int keys[kBlock / 64][H];
  __shared__ __attribute__((aligned(16))) VT vals[kBlock / 64][H];
  __shared__ int ckey[kBlock / 64][H / 2];
  __shared__ WaveFlatScratch s_wf[kBlock / 64];
  __shared__ Group16Scratch s_g[kBlock / 16];
  __shared__ unsigned short cslot[kBlock / 64][H / 2];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63, g = lane >> 4, gl = lane & 15;
  const int64_t first = ((int64_t)blockIdx.x * (kBlock / 64) + w) * 4;
  const bool act = first + g < nbin;
  const int64_t row = act ? (int64_t)perm[first + g] : 0;
  const int64_t base = act ? (int64_t)rmC[row] : 0;
  const int64_t nnz = act ? (int64_t)rmC[row + 1] - base : 0;
  if (!__any(nnz > (int64_t)kQuadNnz)) {                   // wave-uniform
    int* mk = keys[w] + g * HG; VT* mv = vals[w] + g * HG;
    int* ck = ckey[w] + g * (HG / 2); unsigned short* cs = cslot[w] + g * (HG / 2);
    for (int i = gl; i < HG; i += 16) { mk[i] = -1; mv[i] = VT(0); }
    KK_WAVE_SYNC();
    group16_flat_products<OffT, VT>(act, row, rmA, entA, rmB, entB, valB, lane, s_g[t >> 4],
                                    [&](int64_t a, int c, VT bv) { hash_accumulate<VT>(mk, mv, HG - 1, c, valA[a] * bv); });
    KK_WAVE_SYNC();
    int n = 0;                                             // the group's entries so far (the same in its 16 lanes)
    for (int s0 = 0; s0 < HG; s0 += 16) {
      const int key = mk[s0 + gl];
      const unsigned occ = (unsigned)((__ballot(key >= 0) >> (16 * g)) & 0xffffull);
      if (key >= 0) {
        const int pos = n + __popc(occ & ((1u << gl) - 1u));
        if (pos < HG / 2) { ck[pos] = key; cs[pos] = (unsigned short)(s0 + gl); }
      }
      n += __popc(occ);
    }
    KK_WAVE_SYNC();
    if (n > HG / 2) n = HG / 2;                            // (cannot happen: at most kQuadNnz entries)
    if (act) {
      for (int i = gl; i < n; i += 16) {
        const int key = ck[i];
        int rank = 0;
        for (int q = 0; q < n; ++q) rank += (ck[q] < key) ? 1 : 0;
        entC[base + rank] = key;
        valC[base + rank] = mv[cs[i]];
      }
    }
  } else {
    for (int r = 0; r < 4; ++r) {
      const bool a = first + r < nbin;
      const int64_t rw = a ? (int64_t)perm[first + r] : 0;
      num_wave_row<OffT, VT>(a, rw, rmA, entA, valA, rmB, entB, valB, rmC, entC, valC, keys[w], vals[w], ckey[w], cslot[w], s_wf[w], lane);
      KK_WAVE_SYNC();
    }
  }
}

template <class OffT, class VT, int H>
__global__ __launch_bounds__(kBlock) void spgemm_num_block_kernel(int64_t nbin, const int32_t* __restrict__ perm,
                                                                  const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                  const VT* __restrict__ valA, const OffT* __restrict__ rmB,
                                                                  const int32_t* __restrict__ entB, const VT* __restrict__ valB,
                                                                  const OffT* __restrict__ rmC, int32_t* __restrict__ entC,
                                                                  VT* __restrict__ valC, int sg_log2) {
  __shared__ int keys[H];
  __shared__ int slot[H];
  __shared__ VT vals[H];
  __shared__ FlatScratch<kBlock> s_flat;
  const int t = threadIdx.x;
  const int64_t row = perm[blockIdx.x];
  for (int i = t; i < H; i += kBlock) { keys[i] = -1; vals[i] = VT(0); }
  __syncthreads();
  flat_products_v<kBlock, OffT, VT>(row, rmA, entA, rmB, entB, valB, s_flat,
                                    [&](int64_t a, int c, VT bv) { hash_accumulate<VT>(keys, vals, H - 1, c, valA[a] * bv); });
  (void)sg_log2;
  __syncthreads();
  // bitonic network over (key, slot) with empties pushed to the end as INT_MAX
  for (int i = t; i < H; i += kBlock) { slot[i] = i; if (keys[i] < 0) keys[i] = INT_MAX; }
  __syncthreads();
  for (int k = 2; k <= H; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < H; i += kBlock) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const int ka = keys[i], kb = keys[ixj];
          const bool up = ((i & k) == 0);
          if ((ka > kb) == up && ka != kb) { keys[i] = kb; keys[ixj] = ka; const int sa = slot[i]; slot[i] = slot[ixj]; slot[ixj] = sa; }
        }
      }
      __syncthreads();
    }
  const int64_t base = (int64_t)rmC[row];
  const int cnt      = (int)((int64_t)rmC[row + 1] - base);
  for (int i = t; i < cnt; i += kBlock) { entC[base + i] = keys[i]; valC[base + i] = vals[slot[i]]; }
  (void)nbin;
}

// hub rows (more than kValLa entries in the A row, or any dense row when B is not column-sorted): the row is spread
// over many workgroups, which accumulate with L2 atomics into a k-wide accumulator in HBM (one per row of the batch,
// blockIdx.y); entries(C) are already in place, so the sums are then gathered in C order and the touched accumulator
// cells are cleared for the next batch.  Accumulator cells are read with agent-scope loads (served by L2, where the
// atomics were performed).
template <class VT> __device__ __forceinline__ VT load_l2(const VT* p);
template <> __device__ __forceinline__ double load_l2<double>(const double* p) {
  return __longlong_as_double((long long)KK_LOAD_L2(reinterpret_cast<const kk_u64*>(p)));
}
template <> __device__ __forceinline__ float load_l2<float>(const float* p) {
  return __int_as_float((int)KK_LOAD_L2(reinterpret_cast<const unsigned*>(p)));
}
template <class OffT, class VT>
__global__ __launch_bounds__(kBlock) void spgemm_hub_acc_kernel(const int32_t* __restrict__ rows, const OffT* __restrict__ rmA,
                                                                const int32_t* __restrict__ entA, const VT* __restrict__ valA,
                                                                const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                                                const VT* __restrict__ valB, VT* __restrict__ accs, int64_t k,
                                                                int sg_log2) {
  const int64_t row = rows[blockIdx.y];
  VT* acc           = accs + (int64_t)blockIdx.y * k;
  for_each_product_v<OffT, VT>(row, rmA, entA, rmB, entB, valB, (int)(blockIdx.x * kBlock + threadIdx.x), (int)(gridDim.x * kBlock),
                               sg_log2, [&](int64_t a, int c, VT bv) { KK_ATOMIC_FADD(&acc[c], valA[a] * bv); });
}
template <class OffT, class VT>
__global__ __launch_bounds__(kBlock) void spgemm_hub_extract_kernel(const int32_t* __restrict__ rows, const OffT* __restrict__ rmC,
                                                                    const int32_t* __restrict__ entC, VT* __restrict__ valC,
                                                                    VT* __restrict__ accs, int64_t k) {
  const int64_t row  = rows[blockIdx.y];
  VT* acc            = accs + (int64_t)blockIdx.y * k;
  const int64_t base = (int64_t)rmC[row], cnt = (int64_t)rmC[row + 1] - base;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * kBlock) {
    const int c   = entC[base + i];
    valC[base + i] = load_l2<VT>(&acc[c]);
    acc[c]         = VT(0);
  }
}

// dense rows of the numeric phase are split once per handle: rows the windowed LDS value kernel takes first, hub rows
// from the back (wave-aggregated cursors).
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_split_dense_kernel(int64_t nd, const int32_t* __restrict__ perm_in,
                                                                    const OffT* __restrict__ rmA, int64_t la_max, int all_hub,
                                                                    int32_t* __restrict__ perm_out,
                                                                    unsigned long long* __restrict__ counters /*[2]*/) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int lane  = threadIdx.x & 63;
  int32_t row = 0;
  int cls     = -1;
  if (i < nd) { row = perm_in[i]; cls = (all_hub || (int64_t)rmA[row + 1] - (int64_t)rmA[row] > la_max) ? 1 : 0; }
  for (int c = 0; c < 2; ++c) {
    const kk_u64 m = __ballot(cls == c);
    unsigned long long start = 0;
    if (lane == 0 && m) start = atomicAdd(&counters[c], (unsigned long long)__popcll(m));
    start = __shfl(start, 0, 64);
    if (cls == c) {
      const int64_t r = (int64_t)start + __popcll(m & ((1ull << lane) - 1ull));
      perm_out[c == 0 ? r : nd - 1 - r] = row;
    }
  }
}

// ORDER of a list of rows by size, largest first, in quarter-octave classes (a counting sort: histogram, scan of 256 classes, scatter;
// the order inside a class is whatever the atomics give).  Rows that run at the same time then have similar sizes: the heaviest rows
// start first (no long tail behind a late hub row), and rows of similar density walk their windows over similar column ranges at
// similar times -- the segments of B's long rows one of them fetched are still in the XCD's L2 when its neighbours ask for them.
constexpr int kSizeClasses = 256;
__device__ __forceinline__ int size_class_desc(int64_t sz) {      // 0 = the largest sizes
  if (sz < 1) return kSizeClasses - 1;
  const int lg = 63 - __clzll((unsigned long long)sz);           // floor(log2)
  const int frac = lg >= 2 ? (int)((sz >> (lg - 2)) & 3) : 0;    // two bits below the leading one
  const int c = lg * 4 + frac;                                   // 0 .. 255
  return kSizeClasses - 1 - (c < kSizeClasses ? c : kSizeClasses - 1);
}
__global__ __launch_bounds__(kBlock) void spgemm_size_hist_kernel(int64_t n, const int32_t* __restrict__ list, const int64_t* __restrict__ sizes,
                                                                  unsigned* __restrict__ hist) {
  __shared__ unsigned s_h[kSizeClasses];
  for (int i = threadIdx.x; i < kSizeClasses; i += kBlock) s_h[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) atomicAdd(&s_h[size_class_desc(sizes[list[i]])], 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < kSizeClasses; i += kBlock) if (s_h[i]) atomicAdd(&hist[i], s_h[i]);
}
__global__ __launch_bounds__(kSizeClasses) void spgemm_size_scan_kernel(unsigned* __restrict__ hist /* in: counts, out: start of every class; [kSizeClasses]: 1 = leave the order alone */) {
  __shared__ unsigned s_w[kSizeClasses / 64];
  __shared__ int s_lo, s_hi;
  if (threadIdx.x == 0) { s_lo = kSizeClasses; s_hi = -1; }
  __syncthreads();
  const unsigned v = hist[threadIdx.x];
  if (v) { atomicMin(&s_lo, (int)threadIdx.x); atomicMax(&s_hi, (int)threadIdx.x); }
  unsigned tot;
  const unsigned ex = block_exclusive_scan_n<unsigned, kSizeClasses>(v, &tot, s_w);
  hist[threadIdx.x] = ex;
  __syncthreads();
  // rows of (nearly) one size -- a stencil, uniform random rows: at most two neighbouring quarter-octave classes -- keep the order they
  // have: it is the row order, and neighbouring rows of C are neighbours in memory (the atomics of the scatter would scramble them:
  // uniform random 1e6 x 20, numeric reuse 4.19 -> 4.54 ms)
  if (threadIdx.x == 0) hist[kSizeClasses] = (s_hi - s_lo <= 1) ? 1u : 0u;
}
__global__ __launch_bounds__(kBlock) void spgemm_size_scatter_kernel(int64_t n, const int32_t* __restrict__ in, const int64_t* __restrict__ sizes,
                                                                     unsigned* __restrict__ cursor, int32_t* __restrict__ out) {
  // workgroup-aggregated cursors: a class's rows of this workgroup take their places with one global atomic (the rows of a product
  // crowd into a few classes: one global atomic per row was 0.43 ms per call on R-MAT scale 20)
  __shared__ unsigned s_cnt[kSizeClasses], s_base[kSizeClasses];
  if (cursor[kSizeClasses]) {                                    // (uniform) one size: the list stays as it is
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = in[i];
    return;
  }
  for (int64_t i0 = (int64_t)blockIdx.x * kBlock; i0 < n; i0 += (int64_t)gridDim.x * kBlock) {       // uniform trip count
    for (int c = threadIdx.x; c < kSizeClasses; c += kBlock) s_cnt[c] = 0;
    __syncthreads();
    const int64_t i = i0 + threadIdx.x;
    int32_t row = 0; int cls = 0; unsigned local = 0;
    if (i < n) { row = in[i]; cls = size_class_desc(sizes[row]); local = atomicAdd(&s_cnt[cls], 1u); }
    __syncthreads();
    for (int c = threadIdx.x; c < kSizeClasses; c += kBlock) s_base[c] = s_cnt[c] ? atomicAdd(&cursor[c], s_cnt[c]) : 0u;
    __syncthreads();
    if (i < n) out[s_base[cls] + local] = row;
    __syncthreads();
  }
}

// ... and the rows of the first group once more: SMALL ones (at most cnt_max entries in the row of C and la_max in the row of A) first, the
// others from the back.  The small rows get the flat value kernel's light shape (256 work-items, 2048-slot table, 38 KB of LDS: four
// workgroups per CU instead of two): on R-MAT scale 20 245 K of the 454 K rows of this group have 257 .. 8 K entries -- unions of two
// to five lists, 4 % of the products -- and a 512-work-item workgroup around a 4096-slot table spent most of its time per row on set-up.
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_split_small_kernel(int64_t nd, const int32_t* __restrict__ perm_in, const OffT* __restrict__ rmA,
                                                                    const int64_t* __restrict__ sizes, int64_t la_max, int64_t cnt_max,
                                                                    int32_t* __restrict__ perm_out, unsigned long long* __restrict__ counters /*[2]*/) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int lane  = threadIdx.x & 63;
  int32_t row = 0;
  int cls     = -1;
  if (i < nd) { row = perm_in[i]; cls = ((int64_t)rmA[row + 1] - (int64_t)rmA[row] <= la_max && sizes[row] <= cnt_max) ? 0 : 1; }
  for (int c = 0; c < 2; ++c) {
    const kk_u64 m = __ballot(cls == c);
    unsigned long long start = 0;
    if (lane == 0 && m) start = atomicAdd(&counters[c], (unsigned long long)__popcll(m));
    start = __shfl(start, 0, 64);
    if (cls == c) {
      const int64_t r = (int64_t)start + __popcll(m & ((1ull << lane) - 1ull));
      perm_out[c == 0 ? r : nd - 1 - r] = row;
    }
  }
}

// the dense bin as [ other rows | rows for the column-block value kernel ]: A rows above la_min entries or rows of C with at least cnt_min entries
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_split_block_kernel(int64_t nd, const int32_t* __restrict__ perm_in, const OffT* __restrict__ rmA,
                                                                    const int64_t* __restrict__ sizes, int64_t la_min, int64_t cnt_min, int64_t cnt_floor,
                                                                    int32_t* __restrict__ perm_out, unsigned long long* __restrict__ counters /*[2]*/) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int lane  = threadIdx.x & 63;
  int32_t row = 0;
  int cls     = -1;
  if (i < nd) {
    row = perm_in[i];
    const int64_t la = (int64_t)rmA[row + 1] - (int64_t)rmA[row], cnt = sizes[row];
    cls = (cnt >= cnt_min || (la > la_min && cnt >= cnt_floor)) ? 1 : 0;
  }
  for (int c = 0; c < 2; ++c) {
    const kk_u64 m = __ballot(cls == c);
    unsigned long long start = 0;
    if (lane == 0 && m) start = atomicAdd(&counters[c], (unsigned long long)__popcll(m));
    start = __shfl(start, 0, 64);
    if (cls == c) {
      const int64_t r = (int64_t)start + __popcll(m & ((1ull << lane) - 1ull));
      perm_out[c == 0 ? r : nd - 1 - r] = row;
    }
  }
}

// dense rows, values (B rows column-sorted, at most kValLa entries in the A row): entries(C) of the row are already
// in place and sorted (spgemm_dense_cols_kernel<EMIT>), so the row is cut into windows of `cap` consecutive C entries.
// A window's columns are hashed into a clean LDS table (each work-item keeps the slots of its columns in registers).
// The resume point of every A entry (position in B, entries left, A value) lives in LDS, so every B entry is consumed
// exactly once although the row is visited window by window:
//   * B rows of >= kValLong entries: one wave per A entry, two entries in flight, 128 consecutive B entries per step;
//   * shorter B rows: sub-groups of 8..64 lanes (fewer A entries -> wider groups) that each own one A entry at a time
//     and move on when it has nothing left inside the window -- up to 8 independent B streams per wave.
// Columns are probed (known to be present) and accumulated with ds_add.  The sums leave in C order, coalesced, and
// every work-item clears its slots so the table is clean for the next window; the next window's columns are already
// in flight while the current one streams.
template <class OffT, class VT, int H, int NT>
__global__ __launch_bounds__(NT) void spgemm_dense_vals_kernel(const int32_t* __restrict__ perm,
                                                                      const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                      const VT* __restrict__ valA, const OffT* __restrict__ rmB,
                                                                      const int32_t* __restrict__ entB, const VT* __restrict__ valB,
                                                                      const OffT* __restrict__ rmC, const int32_t* __restrict__ entC,
                                                                      VT* __restrict__ valC, int cap KK_DBG_PARAM) {
  __shared__ int hk[H];
  __shared__ VT hv[H];
  __shared__ long long s_cur[kValLa];
  __shared__ int s_rem[kValLa];
  __shared__ VT s_av[kValLa];
  __shared__ unsigned char s_long[kValLa];
  __shared__ int s_whi;
  constexpr int UL = 2, US = 2;       // independent B loads per lane and step (long / short B rows)
  constexpr int EL = 4;               // long B rows in flight per wave
  constexpr int KPT = (H / 2 + NT - 1) / NT;   // C entries of a window per work-item
  constexpr int NW  = NT / 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t row = perm[blockIdx.x];
  const int64_t a0 = (int64_t)rmA[row], la = (int64_t)rmA[row + 1] - a0;
  const int64_t la_c = la < kValLa ? la : kValLa;
  const int64_t base = (int64_t)rmC[row], cnt = (int64_t)rmC[row + 1] - base;
  int sg_log2 = 3;
  while ((int64_t)(NT >> (sg_log2 + 1)) >= la && sg_log2 < 6) ++sg_log2;
  const int sg = 1 << sg_log2, sub = t >> sg_log2, nsub = NT >> sg_log2, sl = t & (sg - 1);
  const int sg_shift   = lane & ~(sg - 1);
  const kk_u64 sg_mask = sg == 64 ? ~0ull : ((1ull << sg) - 1ull);
  // A B row is streamed by a whole wave (UL * 64 entries per step) only when a window is expected to take about a wave's worth
  // of it: the row's entries spread over cnt / cap windows, so len * cap / cnt of them fall into one.  (With the fixed
  // threshold kValLong every window re-read 128 entries of every hub row to consume a handful: 17 of the 23.7 ms of this
  // kernel on R-MAT scale 18 were spent there.)
  const int64_t long_est = 64 * cnt / (cap > 0 ? cap : 1);
  const int long_min = (int)(long_est > kValLong ? (long_est < INT_MAX ? long_est : INT_MAX) : kValLong);
  for (int64_t a = t; a < la_c; a += NT) {
    const int32_t kc = entA[a0 + a];
    const int64_t b0 = (int64_t)rmB[kc];
    const int len    = (int)((int64_t)rmB[kc + 1] - b0);
    s_cur[a] = b0; s_rem[a] = len; s_av[a] = valA[a0 + a]; s_long[a] = len >= long_min ? 1 : 0;
  }
  for (int i = t; i < H; i += NT) { hk[i] = -1; hv[i] = VT(0); }
  auto accumulate = [&](int c, VT v) {
    const int hh = vt_find<H>(hk, c);
    if (hh >= 0) KK_ATOMIC_FADD(&hv[hh], v);
  };
  // one streaming step of a long B row by a whole wave: UL * 64 consecutive entries, columns and values loaded together
  auto load_step = [&](int64_t p, int rem, int* c, VT* v) {
    KK_UNROLL
    for (int u = 0; u < UL; ++u) {
      const int idx = u * 64 + lane;
      const bool ok = idx < rem && !KK_DBG(1024);
      c[u] = ok ? entB[p + idx] : INT_MAX;
      v[u] = ok ? valB[p + idx] : VT(0);
    }
  };
  auto consume_step = [&](const int* c, const VT* v, VT av, int whi) -> int {
    int nin = 0;
    KK_UNROLL
    for (int u = 0; u < UL; ++u) {
      const bool in = c[u] <= whi;
      if (in && !KK_DBG(512)) accumulate(c[u], av * v[u]);
      nin += __popcll(__ballot(in));
    }
    return nin;
  };
  int curk[KPT], slot[KPT];
  KK_UNROLL
  for (int q = 0; q < KPT; ++q) { const int i = t + q * NT; curk[q] = (i < cap && i < cnt) ? entC[base + i] : -1; }
  __syncthreads();
  for (int64_t done = 0; done < cnt; done += cap) {
    const int n = (int)(cnt - done < (int64_t)cap ? cnt - done : (int64_t)cap);
    // build: this window's columns into the (clean) table; the slot of every column stays in a register for the end
    KK_UNROLL
    for (int q = 0; q < KPT; ++q) {
      slot[q] = -1;
      if (curk[q] >= 0) {
        slot[q] = vt_insert<H>(hk, curk[q]);
        if (t + q * NT == n - 1) s_whi = curk[q];
      }
    }
    int nxtk[KPT];        // next window's columns: in flight while this window streams
    KK_UNROLL
    for (int q = 0; q < KPT; ++q) {
      const int i = t + q * NT;
      nxtk[q]     = (i < cap && done + cap + i < cnt) ? entC[base + done + cap + i] : -1;
    }
    __syncthreads();
    const int whi    = s_whi;
    const bool last = done + n >= cnt;
    // long B rows: one wave per A entry, EL entries in flight per wave
    if (!KK_DBG(32)) for (int64_t a = wave; a < la_c; a += EL * NW) {
      bool ok[EL];
      int64_t p[EL];
      int rem[EL];
      VT av[EL];
      int c[EL][UL];
      VT v[EL][UL];
      KK_UNROLL
      for (int e = 0; e < EL; ++e) {
        const int64_t ae = a + e * NW;
        ok[e] = ae < la_c && s_long[ae] != 0;
        if (ok[e]) { p[e] = s_cur[ae]; rem[e] = s_rem[ae]; av[e] = s_av[ae]; load_step(p[e], rem[e], c[e], v[e]); }
      }
      KK_UNROLL
      for (int e = 0; e < EL; ++e) {
        if (!ok[e]) continue;
        int nin = consume_step(c[e], v[e], av[e], whi);
        p[e] += nin; rem[e] -= nin;
        while (nin == UL * 64) { load_step(p[e], rem[e], c[e], v[e]); nin = consume_step(c[e], v[e], av[e], whi); p[e] += nin; rem[e] -= nin; }
        if (!last && lane == 0) { s_cur[a + e * NW] = p[e]; s_rem[a + e * NW] = rem[e]; }
      }
    }
    // short B rows: persistent sub-groups
    int64_t a = sub, p = 0;
    int rem = 0;
    VT av   = VT(0);
    bool have = false;
    auto fetch = [&]() {
      while (a < la_c && s_long[a]) a += nsub;
      have = a < la_c;
      if (have) { p = s_cur[a]; rem = s_rem[a]; av = s_av[a]; }
    };
    fetch();
    if (KK_DBG(64)) have = false;
    while (true) {
      int c[US];
      VT v[US];
      KK_UNROLL
      for (int u = 0; u < US; ++u) {
        const int idx = u * sg + sl;
        const bool ok = have && idx < rem;
        c[u] = ok ? entB[p + idx] : INT_MAX;
        v[u] = ok ? valB[p + idx] : VT(0);
      }
      int nin = 0;
      KK_UNROLL
      for (int u = 0; u < US; ++u) {
        const bool in = c[u] <= whi;
        if (in) accumulate(c[u], av * v[u]);
        nin += __popcll((__ballot(in) >> sg_shift) & sg_mask);
      }
      if (have) {
        p += nin; rem -= nin;
        if (nin < US * sg) {       // this A entry has nothing more inside the window: park it, take the next one
          if (!last && sl == 0) { s_cur[a] = p; s_rem[a] = rem; }
          a += nsub;
          fetch();
        }
      }
      if (__ballot(have) == 0ull) break;
    }
    __syncthreads();
    // sums leave in C order (coalesced); every work-item cleans the slots it filled, so the table is clean again
    KK_UNROLL
    for (int q = 0; q < KPT; ++q) {
      if (slot[q] >= 0) {
        valC[base + done + t + q * NT] = hv[slot[q]];
        hk[slot[q]] = -1; hv[slot[q]] = VT(0);
      }
      curk[q] = nxtk[q];
    }
    __syncthreads();
  }
}

// dense rows, values, second form (default; B rows column-sorted, at most kValLa entries in the A row): the same windows of
// `cap` consecutive C entries hashed into a clean LDS table, but the B side is walked FLAT.  Every list (one per A entry: a
// sorted B row scaled by the A value) is cut at the upper column of each of the next G windows by a binary search -- G * la
// independent searches spread over the workgroup, once per group of G windows, instead of every wave discovering the end of
// "its" list by streaming 128 entries at a time.  With the cuts known a window is: one scan of the lists' in-window counts,
// then product q goes to work-item q / U (U neighbouring products per work-item, all loads independent, nothing read that is
// not consumed), the hash is probed for a column known to be present, ds_add.  The first form (spgemm_dense_vals_kernel) gave
// the long lists of a window to one wave each: on R-MAT three of a row's sixteen lists carry the products, so one to three
// waves of eight worked through sequential 128-entry round trips while the others sat at the barrier (R-MAT scale 20:
// 158 ms for 1.4e10 products, VALU busy 6 %).  LA = lists per pass (A rows above LA take several passes): 512 with 512 work-items
// (78 KB of LDS: two workgroups per CU), 1024 with 1024 work-items and groups of four windows (88 KB: one per CU) for A rows of
// 513..1024 entries, which the cached-cursor hub kernel below serves at half the rate per product (every window re-fetches the
// cache lines of ~5 useful entries per list: 7x read amplification).
// ST = steps of a window's walk a work-item has in flight: the addresses of ST * U products are worked out and all their loads issued
// before the first product is added (a window of 2048 entries holds 3 to 8 thousand products on R-MAT = two to four steps of NT * U,
// each a dependent chain of LDS searches, two HBM loads, probe and add).
template <class OffT, class VT, int H, int NT, int G, int LA, int ST = 1>
__global__ __launch_bounds__(NT) void spgemm_dense_vals2_kernel(const int32_t* __restrict__ perm,
                                                                       const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                       const VT* __restrict__ valA, const OffT* __restrict__ rmB,
                                                                       const int32_t* __restrict__ entB, const VT* __restrict__ valB,
                                                                       const OffT* __restrict__ rmC, const int32_t* __restrict__ entC,
                                                                       VT* __restrict__ valC, int cap_nt, int64_t nnzB KK_DBG_PARAM) {
  static_assert(NT >= LA, "one work-item per list in the scan");
  // bit 30 of the argument: entries(C) / values(C) -- read once, written once -- go through nontemporal loads and stores, so that they do not
  // push the rows of B out of the caches on their way (knob spgemm_nt)
  const int cap = cap_nt & 0xFFFFFF; const bool nt = (cap_nt >> 30) & 1;
  __shared__ int hk[H];
  __shared__ VT hv[H];
  __shared__ long long s_cur[LA];     // first unconsumed entry of list a (index into entries / values of B)
  __shared__ int s_rem[LA];           // entries left in it
  __shared__ VT s_av[LA];
  __shared__ int s_pos[G][LA];        // entries of list a, counted from s_cur[a], with column <= upper column of window g of the group
  __shared__ int s_pre[LA + 1];       // product offsets of the lists inside the current window
  __shared__ int s_whi[G];
  __shared__ int s_wave[NT / 64];
  constexpr int U   = kProdUnroll;
  constexpr int UV  = 8;                       // entries of a list per unit of the vector walk (ST == 0)
  constexpr int STN = ST > 0 ? ST : 1;         // steps in flight of the scalar walk
  constexpr int KPT = (H / 2 + NT - 1) / NT;   // C entries of a window per work-item
  const int t = threadIdx.x;
  const int64_t row = perm[blockIdx.x];
  const int64_t a0 = (int64_t)rmA[row], la = (int64_t)rmA[row + 1] - a0;
  const int64_t base = (int64_t)rmC[row], cnt = (int64_t)rmC[row + 1] - base;
  for (int i = t; i < H; i += NT) { hk[i] = -1; hv[i] = VT(0); }
  // ST == 0: the vector walk -- for rows whose lists are long enough to fill its units of eight entries: at least 32 entries of C per list
  // of the row (workgroup-uniform).  Rows made of many short lists (the product of two uniform random matrices: 20 lists of 20 products
  // each) keep the scalar walk, one step in flight: their units would be a third empty and a work-item would probe eight times in a row
  // (uniform random 1e6 x 20: numeric reuse 4.19 -> 4.54 ms with the vector walk on every row).
  const bool vec = ST == 0 && cnt >= 32 * la;
  // A rows longer than LA are taken LA lists at a time: every pass walks all windows of the row, the first one stores
  // its sums, the others add theirs (the same workgroup, one pass after the other: no atomics)
  for (int64_t ach = 0; ach < la; ach += LA) {
    const int la_c = (int)(la - ach < LA ? la - ach : LA);
    const bool first_pass = ach == 0;
    if (t < la_c) {
      const int32_t kc = entA[a0 + ach + t];
      const int64_t b0 = (int64_t)rmB[kc];
      s_cur[t] = b0; s_rem[t] = (int)((int64_t)rmB[kc + 1] - b0); s_av[t] = valA[a0 + ach + t];
    }
    int curk[KPT], slot[KPT];
    KK_UNROLL
    for (int q = 0; q < KPT; ++q) { const int i = t + q * NT; curk[q] = (i < cap && i < cnt) ? (nt ? KK_NT_LOAD(entC + base + i) : entC[base + i]) : -1; }
    __syncthreads();
    for (int64_t gdone = 0; gdone < cnt; gdone += (int64_t)G * cap) {
      const int ng = (int)((cnt - gdone + cap - 1) / cap < G ? (cnt - gdone + cap - 1) / cap : G);     // windows in this group
      if (t < ng) {
        const int64_t last = gdone + (int64_t)(t + 1) * cap;
        s_whi[t] = last >= cnt ? INT_MAX : entC[base + last - 1];      // the row's last window takes whatever is left
      }
      __syncthreads();
      // cut every list at every window's upper column: ng * la_c independent lower-bound searches
      for (int idx = t; idx < ng * la_c; idx += NT) {
        const int g = idx / la_c, a = idx - g * la_c;
        const int whi = s_whi[g];
        const int rem = s_rem[a];
        int lo = 0;
        if (whi == INT_MAX) lo = rem;
        else if (KK_DBG(2048)) lo = (int)((long long)rem * (g + 1) / (ng + 1));      // measurement build: no searches
        else {
          const int32_t* eb = entB + s_cur[a];
          int hi = rem;
          while (lo < hi) { const int mid = (lo + hi) >> 1; if (eb[mid] <= whi) lo = mid + 1; else hi = mid; }
        }
        s_pos[g][a] = lo;
      }
      __syncthreads();
      for (int g = 0; g < ng; ++g) {
        const int64_t done = gdone + (int64_t)g * cap;
        // build: this window's columns into the (clean) table; the slot of every column stays in a register for the end
        KK_UNROLL
        for (int q = 0; q < KPT; ++q) slot[q] = curk[q] >= 0 ? vt_insert<H>(hk, curk[q]) : -1;
        int nxtk[KPT];        // next window's columns: in flight while this window is walked
        KK_UNROLL
        for (int q = 0; q < KPT; ++q) {
          const int i = t + q * NT;
          nxtk[q]     = (i < cap && done + cap + i < cnt) ? (nt ? KK_NT_LOAD(entC + base + done + cap + i) : entC[base + done + cap + i]) : -1;
        }
        // in-window counts of the lists -> product offsets
        const int from = (t < la_c && g > 0) ? s_pos[g - 1][t] : 0;
        const int n_in = t < la_c ? s_pos[g][t] - from : 0;
        int tot;
        // ST == 0: the unit of the walk is a run of up to UV consecutive entries of ONE list (see below): the scan counts units
        const int excl = block_exclusive_scan_n<int, NT>(vec ? (n_in + UV - 1) / UV : n_in, &tot, s_wave);     // two barriers: the table is built when it returns
        if (t < la_c) s_pre[t] = excl;
        if (t == 0) s_pre[la_c] = tot;
        __syncthreads();
        auto find = [&](int q) {               // largest a in [0, la_c) with pre[a] <= q
          int lo = 0, len2 = la_c;
          while (len2 > 1) { const int half = len2 >> 1; lo += (s_pre[lo + half] <= q) ? half : 0; len2 -= half; }
          return lo;
        };
        if (vec) {
          // VECTOR WALK: work-item u of a step takes unit u -- entries 8 i .. 8 i + 7 of one list's piece inside the window -- with two
          // 16-byte loads of columns and four of values (global loads need 4-byte alignment only): one search per EIGHT products
          // instead of one per four, three times fewer load instructions per product, twice the bytes in flight per work-item, and the
          // lanes of a wave that share a list read one contiguous run (64 lanes = 2 KB of columns).  A list's last unit reads past
          // the list's piece (masked), never past the arrays (the few units that would take the guarded loads).
          if (!KK_DBG(4096)) for (int ubase = 0; ubase < tot; ubase += NT) {
            const int u = ubase + t;
            int cnt_u = 0;
            long long first = 0;
            VT ava = VT(0);
            if (u < tot) {
              const int a = find(u);
              const int fr = g > 0 ? s_pos[g - 1][a] : 0;
              const int k0 = (u - s_pre[a]) * UV;
              cnt_u = s_pos[g][a] - fr - k0; cnt_u = cnt_u < UV ? cnt_u : UV;
              first = s_cur[a] + fr + k0;
              ava = s_av[a];
            }
            int cc[UV];
            VT vv[UV];
            if (first + UV <= (long long)nnzB) {
              typedef int kk_i4 __attribute__((vector_size(16)));
              kk_i4 c4[UV / 4];
              KK_UNROLL
              for (int e = 0; e < UV / 4; ++e) __builtin_memcpy(&c4[e], entB + first + 4 * e, 16);
              KK_UNROLL
              for (int e = 0; e < UV; ++e) vv[e] = valB[first + e];        // (consecutive: the compiler merges them into 16-byte loads)
              KK_UNROLL
              for (int e = 0; e < UV; ++e) cc[e] = c4[e / 4][e % 4];
            } else {
              KK_UNROLL
              for (int e = 0; e < UV; ++e) { const long long j = first + e < (long long)nnzB ? first + e : (long long)nnzB - 1; cc[e] = entB[j]; vv[e] = valB[j]; }
            }
            KK_UNROLL
            for (int e = 0; e < UV; ++e)
              if (e < cnt_u && !KK_DBG(8192)) {
                const int hh = vt_find<H>(hk, cc[e]);
                if (hh >= 0) KK_ATOMIC_FADD(&hv[hh], ava * vv[e]);
              }
          }
        } else
        if (!KK_DBG(4096)) for (int pbase = 0; pbase < tot; pbase += NT * U * STN) {
          int col[STN][U];
          VT bv[STN][U], av[STN][U];
          long long jj[STN][U];
          KK_UNROLL
          for (int s = 0; s < STN; ++s) {
            KK_UNROLL
            for (int u = 0; u < U; ++u) jj[s][u] = -1;
            if (s > 0 && pbase + s * NT * U >= tot) continue;                  // uniform: the window ends before this step
            const int q0 = pbase + (s * NT + t) * U;
            int a = q0 < tot ? find(q0) : 0;
            // the list of the current product stays in registers: neighbouring products are mostly of one list
            int pre_next = s_pre[a + 1];
            long long jb = s_cur[a] + (g > 0 ? s_pos[g - 1][a] : 0) - s_pre[a];      // product q of this list is entry jb + q of B
            VT ava = s_av[a];
            KK_UNROLL
            for (int u = 0; u < U; ++u) {
              const int q = q0 + u;
              if (q < tot) {
                if (q >= pre_next) {
                  do { ++a; pre_next = s_pre[a + 1]; } while (q >= pre_next);   // also steps over lists with nothing in the window; q < tot = pre[la_c] ends it
                  jb = s_cur[a] + (g > 0 ? s_pos[g - 1][a] : 0) - s_pre[a]; ava = s_av[a];
                }
                jj[s][u] = jb + q;
                av[s][u] = ava;
              }
            }
          }
          // every load is UNCONDITIONAL (a product that does not exist reads entry 0 of B: tot > 0, so B has one): a load the compiler
          // has to branch around makes it wait for all loads in flight before the next one is issued -- the walk then pays one memory
          // round trip per load instead of one per step
          KK_UNROLL
          for (int s = 0; s < STN; ++s) {
            KK_UNROLL
            for (int u = 0; u < U; ++u) { const long long j = jj[s][u] >= 0 ? jj[s][u] : 0; col[s][u] = entB[j]; bv[s][u] = valB[j]; }
          }
          KK_UNROLL
          for (int s = 0; s < STN; ++s) {
            KK_UNROLL
            for (int u = 0; u < U; ++u)
              if (jj[s][u] >= 0 && !KK_DBG(8192)) {
                const int hh = vt_find<H>(hk, col[s][u]);
                if (hh >= 0) KK_ATOMIC_FADD(&hv[hh], av[s][u] * bv[s][u]);
              }
          }
        }
        __syncthreads();
        // sums leave in C order (coalesced); every work-item cleans the slots it filled, so the table is clean again
        KK_UNROLL
        for (int q = 0; q < KPT; ++q) {
          if (slot[q] >= 0) {
            VT* out = valC + base + done + t + q * NT;
            const VT sum = first_pass ? hv[slot[q]] : *out + hv[slot[q]];
            if (nt) KK_NT_STORE(out, sum); else *out = sum;
            hk[slot[q]] = -1; hv[slot[q]] = VT(0);
          }
          curk[q] = nxtk[q];
        }
        __syncthreads();
      }
      if (t < la_c) { const int adv = s_pos[ng - 1][t]; s_cur[t] += adv; s_rem[t] -= adv; }
      __syncthreads();
    }
  }
}

// dense rows, values, A rows of kValLa < entries <= kHubLa (B sorted): same windows and table as above, but with
// thousands of A entries most of them have nothing inside a given window, so visiting each one per window (a global
// load each) would dominate.  Here the NEXT unconsumed column of every A entry is cached in LDS next to its cursor:
// a window starts with an LDS-only sweep that lists the entries whose next column falls inside the window, and only
// those are streamed (sub-groups of 16 lanes, persistent over the list).  One workgroup of 16 waves per CU.
// lanes per listed entry and loads per lane and step.  An entry of an A row with thousands of them has a handful of columns inside
// a window (R-MAT scale 20: 4.5 on average), so a step of 2 x 16 entries reads seven times what it uses -- and still 8 lanes x 1
// load measured SLOWER (numeric 351 -> 385 ms): the 128 sub-groups then park and re-fetch twice as many entries per window.
#ifndef KK_HUB_SG
#define KK_HUB_SG 16
#endif
#ifndef KK_HUB_US
#define KK_HUB_US 2
#endif
#ifndef KK_HUB_EL
#define KK_HUB_EL 4          // listed entries a sub-group has in flight
#endif
template <class OffT, class VT>
__global__ __launch_bounds__(kDenseBlock) void spgemm_hub_vals_kernel(const int32_t* __restrict__ perm,
                                                                      const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                      const VT* __restrict__ valA, const OffT* __restrict__ rmB,
                                                                      const int32_t* __restrict__ entB, const VT* __restrict__ valB,
                                                                      const OffT* __restrict__ rmC, const int32_t* __restrict__ entC,
                                                                      VT* __restrict__ valC, int cap_nt, const int32_t* __restrict__ items) {
  const int cap = cap_nt & 0xFFFFFF; const bool nt = (cap_nt >> 30) & 1;      // (see spgemm_dense_vals2_kernel)
  constexpr int H = kValTable, NT = kDenseBlock, KPT = (H / 2 + NT - 1) / NT, SG = KK_HUB_SG, NSUB = NT / SG, US = KK_HUB_US, EL = KK_HUB_EL;
  constexpr unsigned long long kSgMask = (1ull << SG) - 1ull;
  __shared__ int hk[H];
  __shared__ VT hv[H];
  __shared__ long long s_cur[kHubLa];
  __shared__ int s_rem[kHubLa];
  __shared__ int s_next[kHubLa];
  __shared__ unsigned short s_list[kHubLa];
  __shared__ int s_whi, s_nact;
  const int t = threadIdx.x, lane = t & 63, sub = t / SG, sl = t & (SG - 1);
  const int sg_shift = lane & ~(SG - 1);
  // items (optional): workgroup b does ONE pass of one row -- (index into perm, pass) pairs -- and adds its sums with atomics when the
  // row has several (its values were zeroed before the launch): the heaviest row of R-MAT scale 20 (39,580 entries of A = 10 passes
  // over 237 windows) kept one CU busy for most of the kernel's 65 ms
  const int64_t row = perm[items ? items[2 * blockIdx.x] : (int)blockIdx.x];
  const int only_pass = items ? items[2 * blockIdx.x + 1] : -1;
  const int64_t a00 = (int64_t)rmA[row], la_all = (int64_t)rmA[row + 1] - a00;
  const int64_t base = (int64_t)rmC[row], cnt = (int64_t)rmC[row + 1] - base;
  for (int i = t; i < H; i += NT) { hk[i] = -1; hv[i] = VT(0); }
  // A rows longer than kHubLa are taken kHubLa entries at a time: every pass walks all windows of the row, the first one stores
  // its sums, the others add theirs (the same workgroup, one pass after the other: no atomics).  These rows used to accumulate
  // through L2 atomics into a k-wide HBM accumulator (spgemm_hub_acc_kernel: R-MAT scale 20, 211 rows, 8.8e8 products in 44 ms).
  const bool shared_row = only_pass >= 0 && la_all > kHubLa;        // other workgroups add to the same values
  for (int64_t ach = only_pass >= 0 ? (int64_t)only_pass * kHubLa : 0; ach < la_all; ach += kHubLa) {
  const bool first_pass = ach == 0;
  const int64_t a0 = a00 + ach;
  const int la     = (int)(la_all - ach < kHubLa ? la_all - ach : kHubLa);
  for (int a = t; a < la; a += NT) {
    const int32_t kc = entA[a0 + a];
    const int64_t b0 = (int64_t)rmB[kc];
    const int len    = (int)((int64_t)rmB[kc + 1] - b0);
    s_cur[a] = b0; s_rem[a] = len; s_next[a] = len > 0 ? entB[b0] : INT_MAX;
  }
  if (t == 0) s_nact = 0;
  int curk[KPT], slot[KPT];
  KK_UNROLL
  for (int q = 0; q < KPT; ++q) { const int i = t + q * NT; curk[q] = (i < cap && i < cnt) ? entC[base + i] : -1; }
  __syncthreads();
  for (int64_t done = 0; done < cnt; done += cap) {
    const int n = (int)(cnt - done < (int64_t)cap ? cnt - done : (int64_t)cap);
    KK_UNROLL
    for (int q = 0; q < KPT; ++q) {
      slot[q] = -1;
      if (curk[q] >= 0) {
        slot[q] = vt_insert<H>(hk, curk[q]);
        if (t + q * NT == n - 1) s_whi = curk[q];
      }
    }
    int nxtk[KPT];
    KK_UNROLL
    for (int q = 0; q < KPT; ++q) {
      const int i = t + q * NT;
      nxtk[q]     = (i < cap && done + cap + i < cnt) ? (nt ? KK_NT_LOAD(entC + base + done + cap + i) : entC[base + done + cap + i]) : -1;
    }
    __syncthreads();
    const int whi = s_whi;
    // LDS-only sweep: which A entries have something inside this window?
    for (int a0i = 0; a0i < la; a0i += NT) {
      const int a      = a0i + t;
      const bool act   = a < la && s_next[a] <= whi;
      const kk_u64 m   = __ballot(act);
      int start = 0;
      if (lane == 0 && m) start = atomicAdd(&s_nact, __popcll(m));
      start = __shfl(start, 0, 64);
      if (act) s_list[start + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)a;
    }
    __syncthreads();
    const int nact = s_nact;
    // Every sub-group walks EL listed entries at a time: their first steps are requested together (with one entry per sub-group
    // and step, a window was a chain of nact / NSUB dependent round trips -- 44 of them for an A row of 2800 entries, R-MAT scale 20).
    // The trip counts are the same for every wave (ballots inside).
    for (int it = 0; it * NSUB * EL < nact; ++it) {
      int a[EL], rem[EL], c[EL][US];
      int64_t p[EL];
      VT av[EL], v[EL][US];
      bool have[EL];
      KK_UNROLL
      for (int e = 0; e < EL; ++e) {
        const int li = (it * EL + e) * NSUB + sub;
        have[e] = li < nact;
        a[e] = 0; p[e] = 0; rem[e] = 0; av[e] = VT(0);
        if (have[e]) { a[e] = s_list[li]; p[e] = s_cur[a[e]]; rem[e] = s_rem[a[e]]; av[e] = valA[a0 + a[e]]; }
      }
      KK_UNROLL
      for (int e = 0; e < EL; ++e) {
        KK_UNROLL
        for (int u = 0; u < US; ++u) {
          const int idx = u * SG + sl;
          const bool ok = have[e] && idx < rem[e];
          c[e][u] = ok ? entB[p[e] + idx] : INT_MAX;
          v[e][u] = ok ? valB[p[e] + idx] : VT(0);
        }
      }
      // consume entry by entry (called with constant indices so that everything stays in registers)
      auto consume = [&](bool open, int ae, int64_t pe, int reme, VT ave, int (&ce)[US], VT (&ve)[US]) {
        while (true) {
          int nin = 0;
          KK_UNROLL
          for (int u = 0; u < US; ++u) {
            const bool in = ce[u] <= whi;
            if (in) {
              const int hh = vt_find<H>(hk, ce[u]);
              if (hh >= 0) KK_ATOMIC_FADD(&hv[hh], ave * ve[u]);
            }
            nin += __popcll((__ballot(in) >> sg_shift) & kSgMask);
          }
          if (open) {
            if (nin < US * SG) {     // done with this entry for the window: the first column left out becomes its next column
              KK_UNROLL
              for (int u = 0; u < US; ++u) if (u * SG + sl == nin) s_next[ae] = ce[u];
              if (sl == 0) { s_cur[ae] = pe + nin; s_rem[ae] = reme - nin; }
              open = false;
            } else { pe += nin; reme -= nin; }
          }
          if (__ballot(open) == 0ull) break;
          KK_UNROLL
          for (int u = 0; u < US; ++u) {        // rare: more than US * SG entries of one B row inside the window
            const int idx = u * SG + sl;
            const bool ok = open && idx < reme;
            ce[u] = ok ? entB[pe + idx] : INT_MAX;
            ve[u] = ok ? valB[pe + idx] : VT(0);
          }
        }
      };
      static_assert(EL == 4 || EL == 8, "explicit calls below");
      consume(have[0], a[0], p[0], rem[0], av[0], c[0], v[0]);
      consume(have[1], a[1], p[1], rem[1], av[1], c[1], v[1]);
      consume(have[2], a[2], p[2], rem[2], av[2], c[2], v[2]);
      consume(have[3], a[3], p[3], rem[3], av[3], c[3], v[3]);
      if constexpr (EL == 8) {
        consume(have[4], a[4], p[4], rem[4], av[4], c[4], v[4]);
        consume(have[5], a[5], p[5], rem[5], av[5], c[5], v[5]);
        consume(have[6], a[6], p[6], rem[6], av[6], c[6], v[6]);
        consume(have[7], a[7], p[7], rem[7], av[7], c[7], v[7]);
      }
    }
    __syncthreads();
    KK_UNROLL
    for (int q = 0; q < KPT; ++q) {
      if (slot[q] >= 0) {
        VT* out = valC + base + done + t + q * NT;
        if (shared_row) KK_ATOMIC_FADD(out, hv[slot[q]]);        // global_atomic_add_f64 into the row's (zeroed) values
        else { const VT sum = first_pass ? hv[slot[q]] : *out + hv[slot[q]]; if (nt) KK_NT_STORE(out, sum); else *out = sum; }
        hk[slot[q]] = -1; hv[slot[q]] = VT(0);
      }
      curk[q] = nxtk[q];
    }
    if (t == 0) s_nact = 0;
    __syncthreads();
  }
  if (only_pass >= 0) break;
  }
}
// passes per listed row (host builds the (row, pass) items from it) and zeroing of the rows that several workgroups add to
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_hub_passes_kernel(int64_t n, const int32_t* __restrict__ perm, const OffT* __restrict__ rmA, int32_t* __restrict__ passes) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) { const int64_t la = (int64_t)rmA[perm[i] + 1] - (int64_t)rmA[perm[i]]; passes[i] = (int32_t)((la + kHubLa - 1) / kHubLa); }
}
template <class OffT, class VT>
__global__ __launch_bounds__(kBlock) void spgemm_zero_rows_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ idx, const OffT* __restrict__ rmC, VT* __restrict__ valC) {
  const int64_t row = perm[idx[blockIdx.x]];
  for (int64_t i = (int64_t)rmC[row] + threadIdx.x; i < (int64_t)rmC[row + 1]; i += kBlock) valC[i] = VT(0);
}


// ------------------------------------------------------------------------------------------------
// DENSE rows, values, by COLUMN BLOCKS (round 5; the reference's dense accumulator where it pays, impl_speed.hpp:28-150 and the
// selection rule impl_kkmem.hpp:1259-1300, re-cut for 160 KB of LDS).  The columns are cut into blocks of W (16384: a direct-indexed
// fp64 accumulator of 128 KB); a work unit is (row of C, column block) and is INDEPENDENT of every other unit: no cursors carried from
// window to window, no hash, no probe, no search at run time --
//   * an index of B, built once per handle and B (spgemm_bidx_kernel): bidx[cb][j] = entries of row j of B with a column below cb W, so
//     the piece of list (A entry) a inside block cb is rmB[j] + bidx[cb][j] .. rmB[j] + bidx[cb + 1][j]: two gathers, no binary search
//     (the windowed kernels cut every list at every window by a lower-bound search in HBM: 13 of the 28 ms of the 1024-list shape);
//   * an index of C for the class's rows (spgemm_cidx_kernel): where block cb starts in the row's sorted entries(C);
//   * the products of a unit are walked in units of eight entries of one list (16-byte loads, one search in LDS per eight products)
//     and added with ds_add_f64 at acc[column - cb W]; the sums leave by reading entries(C) of the block and gathering from acc.
// An A row of any length is taken NT lists at a time into the same accumulator (no passes over windows, no atomics on values(C): the
// hub rows' sums are deterministic up to the order of the LDS adds), and the launch is block-major: the workgroups in flight at any
// time work on the same few column blocks, whose pieces of B's long rows stay in the XCDs' L2.
// For rows at least 1/32 dense (or with more lists than the flat kernel's shapes hold): R-MAT scale 20, 37 % of the products.
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_bidx_kernel(int64_t nB, int nblk, int wshift, const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                                             unsigned* __restrict__ bidx /* [nblk + 1][nB] */) {
  // thread = (block boundary cb, row j), j fastest: coalesced row_map reads and index writes
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (int64_t)(nblk + 1) * nB) return;
  const int cb = (int)(i / nB);
  const int64_t j = i - (int64_t)cb * nB;
  const int64_t b0 = (int64_t)rmB[j];
  const int len = (int)((int64_t)rmB[j + 1] - b0);
  const long long bound = (long long)cb << wshift;               // first column of block cb
  int lo = 0, hi = len;
  if (cb == 0) hi = 0;
  else if (cb == nblk) lo = len;
  else while (lo < hi) { const int mid = (lo + hi) >> 1; if ((long long)entB[b0 + mid] < bound) lo = mid + 1; else hi = mid; }
  bidx[i] = (unsigned)lo;
}
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_cidx_kernel(int64_t nrows, const int32_t* __restrict__ perm, int nblk, int wshift, const OffT* __restrict__ rmC,
                                                             const int32_t* __restrict__ entC, unsigned* __restrict__ cidx /* [nrows][nblk + 1] */) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nrows * (int64_t)(nblk + 1)) return;
  const int64_t r = i / (nblk + 1);
  const int cb = (int)(i - r * (nblk + 1));
  const int64_t row = perm[r], b0 = (int64_t)rmC[row];
  const long long len = (long long)rmC[row + 1] - b0;
  const long long bound = (long long)cb << wshift;
  long long lo = 0, hi = len;
  if (cb == 0) hi = 0;
  else if (cb == nblk) lo = len;
  else while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((long long)entC[b0 + mid] < bound) lo = mid + 1; else hi = mid; }
  cidx[i] = (unsigned)lo;
}
template <class OffT, class VT, int NT>
__global__ __launch_bounds__(NT) void spgemm_block_vals_kernel(int64_t nrows, const int32_t* __restrict__ perm, int nblk, int wshift, int64_t nB,
                                                               const unsigned* __restrict__ bidx, const unsigned* __restrict__ cidx,
                                                               const OffT* __restrict__ rmA, const int32_t* __restrict__ entA, const VT* __restrict__ valA,
                                                               const OffT* __restrict__ rmB, const int32_t* __restrict__ entB, const VT* __restrict__ valB,
                                                               int64_t nnzB, const OffT* __restrict__ rmC, const int32_t* __restrict__ entC, VT* __restrict__ valC) {
  KK_DYN_SMEM(VT, acc);                      // [W]
  __shared__ long long s_p0[NT];             // first entry of list a inside the block (index into entries / values of B)
  __shared__ int s_n[NT];                    // entries of it inside the block
  __shared__ int s_pre[NT + 1];              // unit offsets of the lists
  __shared__ VT s_av[NT];
  __shared__ int s_wave[NT / 64];
  constexpr int UV = 8;
  const int t = threadIdx.x;
  // block-major (the grid is rows x blocks, x fastest): all rows of block 0, then all rows of block 1, ...
  const int cb = (int)blockIdx.y;
  const int64_t r = (int64_t)blockIdx.x;
  (void)nrows;
  const unsigned e0 = cidx[r * (nblk + 1) + cb], e1 = cidx[r * (nblk + 1) + cb + 1];
  if (e0 == e1) return;                      // no entry of C here: no product either (uniform)
  const int64_t row = perm[r];
  const int W = 1 << wshift;
  const int c0 = cb << wshift;
  for (int i = t; i < W; i += NT) acc[i] = VT(0);
  const int64_t a0 = (int64_t)rmA[row], la = (int64_t)rmA[row + 1] - a0;
  const unsigned* bx0 = bidx + (int64_t)cb * nB;
  const unsigned* bx1 = bx0 + nB;
  for (int64_t ach = 0; ach < la; ach += NT) {
    const int la_c = (int)(la - ach < NT ? la - ach : NT);
    int n_in = 0;
    if (t < la_c) {
      const int32_t kc = entA[a0 + ach + t];
      const unsigned f = bx0[kc], l = bx1[kc];
      n_in = (int)(l - f);
      s_p0[t] = (long long)rmB[kc] + f; s_n[t] = n_in; s_av[t] = valA[a0 + ach + t];
    }
    int tot;
    const int excl = block_exclusive_scan_n<int, NT>((n_in + UV - 1) / UV, &tot, s_wave);    // (its barriers also publish the zeroed accumulator)
    if (t < la_c) s_pre[t] = excl;
    if (t == 0) s_pre[la_c] = tot;
    __syncthreads();
    auto find = [&](int q) {               // largest a in [0, la_c) with pre[a] <= q
      int lo = 0, len2 = la_c;
      while (len2 > 1) { const int half = len2 >> 1; lo += (s_pre[lo + half] <= q) ? half : 0; len2 -= half; }
      return lo;
    };
    for (int ubase = 0; ubase < tot; ubase += NT) {
      const int u = ubase + t;
      int cnt_u = 0;
      long long first = 0;
      VT ava = VT(0);
      if (u < tot) {
        const int a = find(u);
        const int k0 = (u - s_pre[a]) * UV;
        cnt_u = s_n[a] - k0; cnt_u = cnt_u < UV ? cnt_u : UV;
        first = s_p0[a] + k0;
        ava = s_av[a];
      }
      int cc[UV];
      VT vv[UV];
      if (first + UV <= (long long)nnzB) {
        typedef int kk_i4 __attribute__((vector_size(16)));
        kk_i4 c4[UV / 4];
        KK_UNROLL
        for (int e = 0; e < UV / 4; ++e) __builtin_memcpy(&c4[e], entB + first + 4 * e, 16);
        KK_UNROLL
        for (int e = 0; e < UV; ++e) vv[e] = valB[first + e];
        KK_UNROLL
        for (int e = 0; e < UV; ++e) cc[e] = c4[e / 4][e % 4];
      } else {
        KK_UNROLL
        for (int e = 0; e < UV; ++e) { const long long j = first + e < (long long)nnzB ? first + e : (long long)nnzB - 1; cc[e] = entB[j]; vv[e] = valB[j]; }
      }
      KK_UNROLL
      for (int e = 0; e < UV; ++e)
        if (e < cnt_u) KK_ATOMIC_FADD(&acc[cc[e] - c0], ava * vv[e]);
    }
    __syncthreads();                       // the next chunk of lists overwrites the descriptors; after the last one: the sums are complete
  }
  const int64_t base = (int64_t)rmC[row];
  for (unsigned i = e0 + t; i < e1; i += NT) valC[base + i] = acc[entC[base + i] - c0];
}

// ITEMS.  The column-block kernel above gives every (row, 16384-column block) a workgroup and 128 KB of sums whatever the block holds;
// on rows that are 6 .. 12 % dense a block holds one or two thousand entries and the windowed kernels were faster (R-MAT scale 20
// reuse: 117 ms with the rows above 12 % in blocks, 136.5 with the rows above 6 %).  Here the unit is an ITEM built from the index of
// the row's entries(C):
//   * consecutive blocks of a row are grouped while the group holds at most `cap` entries (and at most kItemMaxBlocks blocks): a RANK
//     item.  Its accumulator is indexed by the entry's POSITION in the row: a packed word per 32 columns of the group -- the columns'
//     bits in the low half, the number of entries before them in the high half -- turns a product's column into its position with one
//     8-byte LDS read and a popcount, and the sums leave as one contiguous run of values(C).  16 KB of words + 8 bytes per entry:
//     two workgroups per CU, and a sparse stretch of a row costs one item, not four;
//   * a block that alone holds more than `cap` entries (more than 37 % dense) stays a DIRECT item: the column-indexed accumulator.
// Items are ordered by their first block, so the launch is block-major as before.  (row index in the class, first block | end block << 16)
__global__ __launch_bounds__(kBlock) void spgemm_items_build_kernel(int64_t nrows, int nblk, const unsigned* __restrict__ cidx, unsigned cap, int maxblk, int fill,
                                                                   unsigned* __restrict__ n_rank /* [nrows + 1] counts, then offsets */, unsigned* __restrict__ n_direct,
                                                                   int2* __restrict__ out_rank, int2* __restrict__ out_direct) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= nrows) return;
  const unsigned* cx = cidx + r * (nblk + 1);
  unsigned nr = 0, nd = 0;
  const unsigned o_r = fill ? n_rank[r] : 0u, o_d = fill ? n_direct[r] : 0u;
  int cb = 0;
  while (cb < nblk) {
    const unsigned e0 = cx[cb], e1 = cx[cb + 1];
    if (e1 == e0) { ++cb; continue; }                                  // no entry of C in this block: no product either
    if (e1 - e0 > cap) {
      if (fill) out_direct[o_d + nd] = int2{(int)r, cb | ((cb + 1) << 16)};
      ++nd; ++cb; continue;
    }
    int g1 = cb + 1;
    while (g1 < nblk && g1 - cb < maxblk && cx[g1 + 1] - e0 <= cap) ++g1;
    if (fill) out_rank[o_r + nr] = int2{(int)r, cb | (g1 << 16)};
    ++nr; cb = g1;
  }
  if (!fill) { n_rank[r] = nr; n_direct[r] = nd; }
}
// counting sort of items by their first block: histogram, (serial) scan of the nblk + 1 counters, scatter
__global__ __launch_bounds__(kBlock) void spgemm_items_hist_kernel(int64_t n, const int2* __restrict__ items, int nblk, unsigned* __restrict__ hist) {
  // (workgroup-aggregated: the items of a product crowd into a few dozen first blocks)
  __shared__ unsigned s_h[4096];
  for (int b = threadIdx.x; b < nblk; b += kBlock) s_h[b] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) atomicAdd(&s_h[items[i].y & 0xffff], 1u);
  __syncthreads();
  for (int b = threadIdx.x; b < nblk; b += kBlock) if (s_h[b]) atomicAdd(&hist[b], s_h[b]);
}
__global__ void spgemm_items_scan_kernel(int nblk, unsigned* __restrict__ hist) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { unsigned run = 0; for (int b = 0; b < nblk; ++b) { const unsigned c = hist[b]; hist[b] = run; run += c; } }
}
__global__ __launch_bounds__(kBlock) void spgemm_items_scatter_kernel(int64_t n, const int2* __restrict__ items, int nblk, unsigned* __restrict__ cursor, int2* __restrict__ out) {
  __shared__ unsigned s_h[4096], s_b[4096];
  for (int64_t i0 = (int64_t)blockIdx.x * kBlock; i0 < n; i0 += (int64_t)gridDim.x * kBlock) {      // uniform trip count
    for (int b = threadIdx.x; b < nblk; b += kBlock) s_h[b] = 0;
    __syncthreads();
    const int64_t i = i0 + threadIdx.x;
    int2 it = int2{0, 0}; unsigned local = 0;
    if (i < n) { it = items[i]; local = atomicAdd(&s_h[it.y & 0xffff], 1u); }
    __syncthreads();
    for (int b = threadIdx.x; b < nblk; b += kBlock) s_b[b] = s_h[b] ? atomicAdd(&cursor[b], s_h[b]) : 0u;
    __syncthreads();
    if (i < n) out[s_b[it.y & 0xffff] + local] = it;
    __syncthreads();
  }
}

// What an item's workgroup needs before it can ask for anything else, in one 32-byte record (built once per symbolic phase from the ordered
// items): its blocks, its entries of C and where they start, its row of A.  (Read from the item, the class list, the index of C and the two
// row maps these were three dependent trips to memory at the start of every one of 1.5 million workgroups on R-MAT scale 20.)
struct alignas(16) ItemHead { int cbs; int ne; long long base; long long a0; int la; int pad; };
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_items_head_kernel(int64_t n, const int2* __restrict__ items, const int32_t* __restrict__ perm, int nblk,
                                                                  const unsigned* __restrict__ cidx, const OffT* __restrict__ rmA, const OffT* __restrict__ rmC,
                                                                  ItemHead* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int2 it = items[i];
  const int64_t r = it.x;
  const int cb0 = it.y & 0xffff, cb1 = (int)((unsigned)it.y >> 16);
  const unsigned e0 = cidx[r * (nblk + 1) + cb0], e1 = cidx[r * (nblk + 1) + cb1];
  const int64_t row = perm[r];
  ItemHead hd;
  hd.cbs = it.y; hd.ne = (int)(e1 - e0); hd.base = (long long)rmC[row] + e0; hd.a0 = (long long)rmA[row]; hd.la = (int)((int64_t)rmA[row + 1] - (int64_t)rmA[row]); hd.pad = 0;
  out[i] = hd;
}

template <class OffT, class VT, int NT, bool RANK>
__global__ __launch_bounds__(NT) void spgemm_item_vals_kernel(const ItemHead* __restrict__ items, int wshift, int64_t nB,
                                                              const unsigned* __restrict__ bidx,
                                                              const int32_t* __restrict__ entA, const VT* __restrict__ valA,
                                                              const OffT* __restrict__ rmB, const int32_t* __restrict__ entB, const VT* __restrict__ valB,
                                                              int64_t nnzB, const int32_t* __restrict__ entC, VT* __restrict__ valC, int maxblk) {
  KK_DYN_SMEM(char, smem);                   // RANK: [packed words of maxblk blocks][sums by position];  direct: [sums by column of one block]
  __shared__ long long s_p0[NT];             // first entry of list a inside the item's columns (index into entries / values of B)
  __shared__ int s_n[NT];                    // entries of it inside them
  __shared__ int s_pre[NT + 1];              // unit offsets of the lists
  __shared__ VT s_av[NT];
  __shared__ int s_wave[NT / 64];
  constexpr int UV = 8;
  const int t = threadIdx.x;
  const ItemHead hd = items[blockIdx.x];
  const int cb0 = hd.cbs & 0xffff, cb1 = (int)((unsigned)hd.cbs >> 16);
  const int ne = hd.ne;
  const int64_t base = hd.base;
  const int c0 = cb0 << wshift;
  kk_u64* rank = reinterpret_cast<kk_u64*>(smem);
  VT* sums = RANK ? reinterpret_cast<VT*>(smem + ((size_t)maxblk << (wshift - 5)) * 8) : reinterpret_cast<VT*>(smem);
  const int64_t a0 = hd.a0, la = hd.la;
  const unsigned* bx0 = bidx + (int64_t)cb0 * nB;
  const unsigned* bx1 = bidx + (int64_t)cb1 * nB;
  // the first chunk of lists is requested NOW: entries(A) -> index of B / row_map(B) is a chain of two round trips that needs nothing of
  // the accumulator's set-up below and completes behind it
  long long pf_p0 = 0; int pf_n = 0; VT pf_av = VT(0);
  if ((int64_t)t < la) {
    const int32_t kc = entA[a0 + t];
    const unsigned f = bx0[kc], l = bx1[kc];
    pf_n = (int)(l - f); pf_p0 = (long long)rmB[kc] + f; pf_av = valA[a0 + t];
  }
  if constexpr (RANK) {
    // (the item's entries of C -- twelve per work-item at the default capacity -- are asked for before the accumulator is cleared, not
    // behind the barrier that follows it)
    constexpr int EP = 12;
    int ec[EP];
    KK_UNROLL
    for (int e = 0; e < EP; ++e) { const int i = t + e * NT; ec[e] = i < ne ? entC[base + i] : 0; }
    const int nw = (cb1 - cb0) << (wshift - 5);
    for (int i = t; i < nw; i += NT) rank[i] = 0ull;
    for (int i = t; i < ne; i += NT) sums[i] = VT(0);
    __syncthreads();
    unsigned* r32 = reinterpret_cast<unsigned*>(rank);
    KK_UNROLL
    for (int e = 0; e < EP; ++e) if (t + e * NT < ne) { const unsigned c = (unsigned)(ec[e] - c0); atomicOr(&r32[(c >> 5) * 2], 1u << (c & 31u)); }
    for (int i = t + EP * NT; i < ne; i += NT) { const unsigned c = (unsigned)(entC[base + i] - c0); atomicOr(&r32[(c >> 5) * 2], 1u << (c & 31u)); }
    __syncthreads();
    // entries before every word: a contiguous run of words per work-item, one workgroup scan
    const int per = (nw + NT - 1) / NT;
    const int a = t * per, z = a + per < nw ? a + per : nw;
    int cnt = 0;
    for (int w = a; w < z; ++w) cnt += __popc(r32[2 * w]);
    int tot;
    int run = block_exclusive_scan_n<int, NT>(cnt, &tot, s_wave);
    for (int w = a; w < z; ++w) { r32[2 * w + 1] = (unsigned)run; run += __popc(r32[2 * w]); }
  } else {
    const int W = 1 << wshift;
    for (int i = t; i < W; i += NT) sums[i] = VT(0);
  }
  for (int64_t ach = 0; ach < la; ach += NT) {
    const int la_c = (int)(la - ach < NT ? la - ach : NT);
    int n_in = 0;
    if (ach == 0) {
      if (t < la_c) { n_in = pf_n; s_p0[t] = pf_p0; s_n[t] = pf_n; s_av[t] = pf_av; }
    } else if (t < la_c) {
      const int32_t kc = entA[a0 + ach + t];
      const unsigned f = bx0[kc], l = bx1[kc];
      n_in = (int)(l - f);
      s_p0[t] = (long long)rmB[kc] + f; s_n[t] = n_in; s_av[t] = valA[a0 + ach + t];
    }
    int tot;
    const int excl = block_exclusive_scan_n<int, NT>((n_in + UV - 1) / UV, &tot, s_wave);    // (its barriers also publish the accumulator / the words)
    if (t < la_c) s_pre[t] = excl;
    if (t == 0) s_pre[la_c] = tot;
    __syncthreads();
    auto find = [&](int q) {               // largest a in [0, la_c) with pre[a] <= q
      int lo = 0, len2 = la_c;
      while (len2 > 1) { const int half = len2 >> 1; lo += (s_pre[lo + half] <= q) ? half : 0; len2 -= half; }
      return lo;
    };
    for (int ubase = 0; ubase < tot; ubase += NT) {
      const int u = ubase + t;
      int cnt_u = 0;
      long long first = 0;
      VT ava = VT(0);
      if (u < tot) {
        const int a = find(u);
        const int k0 = (u - s_pre[a]) * UV;
        cnt_u = s_n[a] - k0; cnt_u = cnt_u < UV ? cnt_u : UV;
        first = s_p0[a] + k0;
        ava = s_av[a];
      }
      int cc[UV];
      VT vv[UV];
      if (first + UV <= (long long)nnzB) {
        typedef int kk_i4 __attribute__((vector_size(16)));
        kk_i4 c4[UV / 4];
        KK_UNROLL
        for (int e = 0; e < UV / 4; ++e) __builtin_memcpy(&c4[e], entB + first + 4 * e, 16);
        KK_UNROLL
        for (int e = 0; e < UV; ++e) vv[e] = valB[first + e];
        KK_UNROLL
        for (int e = 0; e < UV; ++e) cc[e] = c4[e / 4][e % 4];
      } else {
        KK_UNROLL
        for (int e = 0; e < UV; ++e) { const long long j = first + e < (long long)nnzB ? first + e : (long long)nnzB - 1; cc[e] = entB[j]; vv[e] = valB[j]; }
      }
      KK_UNROLL
      for (int e = 0; e < UV; ++e)
        if (e < cnt_u) {
          const unsigned c = (unsigned)(cc[e] - c0);
          if constexpr (RANK) {
            const kk_u64 pw = rank[c >> 5];
            const int pos = (int)(pw >> 32) + __popc((unsigned)pw & ((1u << (c & 31u)) - 1u));
            KK_ATOMIC_FADD(&sums[pos], ava * vv[e]);
          } else KK_ATOMIC_FADD(&sums[c], ava * vv[e]);
        }
    }
    __syncthreads();                       // the next chunk of lists overwrites the descriptors; after the last one: the sums are complete
  }
  if constexpr (RANK) { for (int i = t; i < ne; i += NT) valC[base + i] = sums[i]; }
  else { for (int i = t; i < ne; i += NT) valC[base + i] = sums[entC[base + i] - c0]; }
}
// ------------------------------------------------------------------------------------------------
}  // namespace kk

struct kkamd_spgemm_handle {
  int64_t m = 0, n = 0, k = 0;
  int offset_type = 0;
  bool symbolic_called = false, numeric_called = false, numeric_bins_ready = false;
  int64_t c_nnz = 0, mults = 0, max_row_flops = 0, max_row_nnz = 0;
  const void *rmA = nullptr, *rmB = nullptr;
  int64_t* d_sizes = nullptr;      // [m] scratch: row flops (symbolic) then nnz per C row (numeric)
  int32_t* d_perm  = nullptr;      // [m] rows grouped by numeric bin
  kk::BinOffsets num_off{};
  int sg_log2 = 0;
  int64_t nnzA = 0, nnzB = 0;
  bool b_sorted = false;           // rows of B column-sorted: dense rows may use the windowed LDS value kernel
  bool dense_lds = false;          // decided when the numeric bins are made
  int32_t* d_hub_items = nullptr; int64_t n_hub_items = 0;      // (index, pass) pairs of the hub rows: one workgroup each
  int32_t* d_hub_multi = nullptr; int64_t n_hub_multi = 0;      // indices of the hub rows with several passes (zeroed before the launch)
  bool hub_from_mid = false;       // the dense bin is cut [<= kValLa | <= kValLa2 | rest]: flat kernel twice, hub kernel for the rest
  int64_t n_wave_quad = 0;         // leading rows of the numeric wave bin that share waves four at a time (at most kQuadNnz entries each)
  int64_t n_dense_tiny = 0;        // leading rows of the dense bin that take the flat value kernel's lightest shape,
  int64_t n_dense_small = 0;       // rows after them that take its light shape (few entries in C and in A)
  int64_t n_dense_lds = 0;         // leading rows of the dense bin taken by the LDS value kernel,
  int64_t n_dense_hub_lds = 0;     // then rows for the LDS hub kernel; the rest accumulate in HBM
  // options (kkamd_spgemm_set; the reference's SPGEMMHandle / KokkosKernelsHandle setters)
  int algorithm = 0;               // 0 hash accumulators in LDS (SPGEMM_KK and its aliases), 1 dense accumulator numeric (SPGEMM_KK_DENSE)
  int compression = 0;             // symbolic phase: 0 never compress B (default: measured 3.19 -> 3.75 ms on 27-pt 100^3 A*A although it removes
                                   // 64 % of the insertions -- the phase is not insertion-bound here), 1 compress and keep it if it pays, 2 always keep it
  double compression_cutoff = 0.85;  // kept when compressed work <= cutoff * original work (impl_compression.hpp:718)
  int verbose = 0;
  int requested_algorithm = 4;     // the SPGEMMAlgorithm the caller named (SPGEMM_DEFAULT until set)
  std::map<std::string, double> hints;   // accepted-and-ignored tuning hints of the reference, by key
  // bitmaps of the densest rows, kept by the symbolic phase for the first numeric call (freed once entries(C) are written)
  void* d_bm_store = nullptr; int32_t* d_row_slot = nullptr; unsigned long long* d_bm_counter = nullptr;
  int64_t bm_cap = 0, bm_stored = 0; int bm_words = 0; bool bm_pooled = false;
  int64_t bitmaps_used = 0;        // rows of the last numeric call whose entries(C) came from a stored bitmap
  int32_t* d_emit_perm = nullptr; int64_t n_emit_stored = 0;    // the dense bin as [rows with a stored bitmap | the others]
  // entry lists of the dense rows whose bitmap is not kept, left by the symbolic phase in the tail of the same buffer
  int32_t* d_ent_pool = nullptr; long long* d_pool_off = nullptr; int64_t pool_cap = 0, pool_used = 0;
  int32_t* d_emit_perm2 = nullptr; int64_t n_emit_pooled = 0;   // "the others" as [rows with a pooled list | rows that walk their products]
  int64_t pooled_used = 0;         // rows of the last numeric call whose entries(C) were copied from the pool
  int32_t* d_flop_cls = nullptr;   // [m] 0: the row has at most kEmitSortCap products, -1: more (left by the symbolic phase)
  int32_t* d_emit_perm3 = nullptr; int64_t n_emit_sort = 0;     // rows that walk their products as [sorted in LDS | through the bitmap kernel] ...
  const int32_t* emit3_src = nullptr; int64_t emit3_n = 0;      // ... of this list
  int64_t sorted_used = 0;         // rows of the last numeric call whose entries(C) were sorted in LDS
  bool entries_valid = false;      // entries(C) as the last numeric call left them are still what entC_ptr holds (numeric reuse)
  bool entries_reused = false;     // the last numeric call kept them
  const void *entC_ptr = nullptr, *rmC_ptr = nullptr;
  // column-block value kernel: rows of the dense bin's tail, the index of B (kept while B's arrays are the same) and of the rows' entries(C)
  int64_t n_dense_block = 0;
  unsigned* d_bidx = nullptr; const void* bidx_rmB = nullptr; const void* bidx_entB = nullptr; int64_t bidx_nB = 0; int bidx_nblk = 0, bidx_wshift = 0;
  unsigned* d_cidx = nullptr; bool cidx_ready = false;
  kk::ItemHead* d_items_rank = nullptr; kk::ItemHead* d_items_direct = nullptr; int64_t n_items_rank = 0, n_items_direct = 0; bool items_ready = false;
  int idx_nblk = 0, idx_wshift = 0, items_cap = 0, items_blocks = 0;      // what the two indices / the items were built for (the knobs are process-wide and may change between calls)
  // the dense class by units (spgemm_sym_unit_kernel): the class's rows in the order the units were numbered, entries and store offset of every unit;
  // d_row_slot holds unit_row (rank of a row whose units all kept their structure, or -1)
  bool unit_mode = false; int unit_nwin = 0, unit_wb = 0; int64_t unit_rows = 0, unit_rows_kept = 0, unit_bitmaps = 0;
  int64_t last_units = 0, last_unit_bitmaps = 0, last_unit_rows_kept = 0;      // of the last symbolic phase (kkamd_spgemm_get 19 - 21; they outlive the store)
  int32_t* d_unit_perm = nullptr; unsigned* d_ucnt = nullptr; unsigned* d_ucoff = nullptr; long long* d_uoff = nullptr;
  kk::UnitHead* d_heads = nullptr; int64_t n_heads = 0;       // the units with products, in launch order
  void* d_unit_block = nullptr;    // one allocation behind d_unit_perm, d_ucnt, d_ucoff, d_uoff, d_row_slot and d_heads (six hipMalloc were 0.3 ms of a phase)
  // a second stream for the symbolic phase: the kernels of the rows with few products (wave / block hash kernels) run beside the dense class
  hipStream_t aux = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // 64 counters on the device and their pinned copy on the host: what a phase reads back (statistics of the row flops, sortedness of B, bin counts;
  // sum and maximum of the row counts) travels in ONE copy per decision point -- a product of a multigrid setup is a dozen launches of a few
  // microseconds each, and every separate read-back (a temporary, a copy, a synchronisation, a hipFree) cost it 30 - 40 us.  d_scan_ws: workspace of the row_map scan
  unsigned long long* d_small = nullptr; unsigned long long* h_small = nullptr; int small_device = -1;
  void* d_scan_ws = nullptr; size_t scan_ws_bytes = 0;
  int64_t sizes_cap = 0;           // rows d_sizes / d_perm hold
  bool tmp_taken = false;          // the running symbolic phase holds the process-wide buffer of temporaries (take_tmp)
  bool compressed = false;         // what the last symbolic call did
  int64_t compressed_mults = 0;
};

namespace kk {

static int pick_sg_log2(int64_t nnzB, int64_t n) {
  const int64_t avg = n > 0 ? nnzB / n : 1;
  int l = 0;
  while ((1 << l) < 64 && (1 << (l + 1)) <= avg) ++l;    // largest power of two <= average B row length
  return l;
}

constexpr int kSmallSlots = 64, kSmallStats = 0, kSmallFlag = 2, kSmallLong = 3, kSmallBins = 8, kSmallSum = 24;      // slots of the handle's counters
// The counters of destroyed handles wait here for the next handle of the same device (a product of a multigrid setup creates a handle, runs two
// phases of a few hundred microseconds and destroys it: a hipMalloc + hipHostMalloc and their frees per handle were a fifth of that);
// kkamd_release_scratch frees them.
struct SmallPool { struct Item { unsigned long long* d; unsigned long long* h; int device; }; std::vector<Item> items; std::mutex m; };
static SmallPool& small_pool() { static SmallPool pool; return pool; }
static bool ensure_small(kkamd_spgemm_handle* h) {
  if (h->d_small && h->h_small) return true;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
  {
    SmallPool& sp = small_pool(); std::lock_guard<std::mutex> g(sp.m);
    for (size_t i = 0; i < sp.items.size(); ++i)
      if (sp.items[i].device == dev) { h->d_small = sp.items[i].d; h->h_small = sp.items[i].h; h->small_device = dev; sp.items.erase(sp.items.begin() + (long)i); return true; }
  }
  if (!h->d_small && hipMalloc((void**)&h->d_small, kSmallSlots * sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); h->d_small = nullptr; return false; }
  if (!h->h_small && hipHostMalloc((void**)&h->h_small, kSmallSlots * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h->h_small = nullptr; return false; }
  h->small_device = dev;
  return true;
}
static void park_small(kkamd_spgemm_handle* h) {                 // (handle destruction; the handle's streams have been waited for)
  if (!h->d_small || !h->h_small) { if (h->d_small) (void)hipFree(h->d_small); if (h->h_small) (void)hipHostFree(h->h_small); h->d_small = nullptr; h->h_small = nullptr; return; }
  SmallPool& sp = small_pool(); std::lock_guard<std::mutex> g(sp.m);
  if (sp.items.size() < 16) sp.items.push_back({h->d_small, h->h_small, h->small_device});
  else { (void)hipFree(h->d_small); (void)hipHostFree(h->h_small); }
  h->d_small = nullptr; h->h_small = nullptr;
}
static void release_small_pool() {
  SmallPool& sp = small_pool(); std::lock_guard<std::mutex> g(sp.m);
  for (auto& it : sp.items) { (void)hipFree(it.d); (void)hipHostFree(it.h); }
  sp.items.clear();
}
// d_cnt: 2 kNumBins counters owned by the caller (zeroed by the caller when `counted`); h_cnt: where their copy on the host goes / is.
// counted: spgemm_bin_count_kernel has run and h_cnt holds its result.  With d_cnt the function neither allocates nor waits after the scatter.
static int make_bins(int64_t m, const int64_t* d_sizes, int64_t cap, const BinLimits& L, int32_t* d_perm,
                     BinOffsets* off, hipStream_t st, unsigned long long* d_cnt = nullptr, unsigned long long* h_cnt_in = nullptr, bool counted = false) {
  DevBuf cnt_b;                              // frees itself on every return
  const bool own = d_cnt == nullptr;
  if (own) {
    KK_HIP(cnt_b.alloc(sizeof(unsigned long long) * 2 * kNumBins));
    d_cnt = cnt_b.as<unsigned long long>();
  }
  const unsigned grid = (unsigned)ceil_div(m, kBlock);
  unsigned long long h_loc[kNumBins];
  unsigned long long* h_cnt = h_cnt_in ? h_cnt_in : h_loc;
  if (!counted) {
    KK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long) * 2 * kNumBins, st));
    KK_LAUNCH(spgemm_bin_count_kernel, grid, kBlock, 0, st, m, d_sizes, cap, L, d_cnt);
    KK_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(unsigned long long) * kNumBins, hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
  }
  off->off[0] = 0;
  for (int b = 0; b < kNumBins; ++b) off->off[b + 1] = off->off[b] + (int64_t)h_cnt[b];
  KK_LAUNCH(spgemm_bin_scatter_kernel, grid, kBlock, 0, st, m, d_sizes, cap, L, *off, d_cnt + kNumBins, d_perm);
  hipError_t e = hipGetLastError();
  hipError_t e2 = own ? hipStreamSynchronize(st) : hipSuccess;       // (the temporary is freed on return)
  if (e != hipSuccess || e2 != hipSuccess) return fail(KKAMD_ERR_HIP, "spgemm binning failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  return KKAMD_OK;
}

// one workgroup per dense row; dynamic LDS = the bitmap window
template <class OffT, bool EMIT>
static int launch_dense_cols(int64_t nrows, const int32_t* perm, const OffT* rmA, const int32_t* entA, const OffT* rmB,
                             const int32_t* entB, OffT* counts, const OffT* rmC, int32_t* entC, int64_t k, int64_t nnzB, int sg, hipStream_t st,
                             const OffT* endB = nullptr, const unsigned* maskB = nullptr, BitmapStore bs = BitmapStore(), int64_t win_cap = 0) {
  int64_t win = g_spgemm.win_bits;
  if (win_cap > 0 && win > win_cap) win = win_cap;        // lighter rows: smaller windows, more workgroups per CU, several passes
  if (win > k) win = ceil_div(k, 64) * 64;
  const size_t smem = (size_t)(win / 8);
  const int64_t quads = (g_spgemm.col_quads && ((uintptr_t)entB % 16 == 0) && nnzB >= 4 && !maskB) ? nnzB : (int64_t)0;
#ifndef KK_EMU
#define KK_DC_ATTR(QQ) KK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spgemm_dense_cols_kernel<OffT, EMIT, QQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem))
#else
#define KK_DC_ATTR(QQ) (void)0
#endif
#define KK_DC(QQ)                                                                                                                    \
  do {                                                                                                                               \
    KK_DC_ATTR(QQ);                                                                                                                  \
    KK_LAUNCH((spgemm_dense_cols_kernel<OffT, EMIT, QQ>), (unsigned)nrows, kDenseBlock, smem, st, perm, rmA, entA, rmB, entB, counts,  \
              rmC, entC, k, (int)win, sg, g_spgemm.emit_chunked, endB, maskB, quads, bs KK_DBG_ARG);                                  \
  } while (0)
  KK_DC(4);                                                   // (8 quads per work-item and step measured slower, 72.3 against 71.1 ms in round 4: not instantiated any more)
#undef KK_DC
#undef KK_DC_ATTR
  return KKAMD_OK;
}

// The bitmap store is GBs (11.4 on R-MAT scale 20) and lives for one symbolic -> numeric hand-over.  hipMalloc / hipFree of
// such a buffer is not free: every third or so symbolic phase took 1.3-1.6 s instead of 66 ms with an allocation per handle.  The
// buffer is therefore kept in a process-wide pool between uses (one user at a time; a second concurrent handle allocates its own);
// kkamd_release_scratch() gives it back.
struct BmPool { void* p = nullptr; size_t bytes = 0; bool in_use = false; int device = -1; int live_handles = 0; bool released_once = false, sticky = false; std::mutex m;
                size_t high_water = 0; };          // the most the process-wide buffer has held (kkamd_spgemm_get 23)
// Who keeps it (knob "spgemm_pool_keep" 0, the default): a process's FIRST product returns the buffer to the device when its last handle is
// destroyed -- a one-shot caller gets its memory back without knowing about the pool.  A process that comes back for the buffer after such
// a release is a repeat user: from then on the buffer outlives the handles (until kkamd_release_scratch), because giving GBs back and
// asking for them again is not cheap on this runtime: hipMalloc of 14 GB returns in 0.3 ms most of the time and in 0.6 - 1.5 s every second
// or third time after a hipFree (tools/probes/probe_malloc.hip, profiles/round5/probe_malloc.txt) -- with a release per handle the
// symbolic phase of R-MAT scale 20 took 1.7 s in two of three repetitions.  1 = always keep, 2 = always return with the last handle.
// kkamd_release_scratch() gives it back at any time.  Its size: the bitmaps of the rows
// that can qualify (at most an eighth of the HBM that was free when it was sized) plus the entry lists the other dense rows can need (at
// most a tenth).  Kokkos-based hosts: INTEGRATION.md registers kkamd_release_scratch with
// Kokkos::push_finalize_hook; the C++ drop-in's Kokkos::finalize() calls it.
static BmPool& bm_pool() { static BmPool pool; return pool; }
// The temporaries of a symbolic phase with a dense class (the two indices, the units' products and order: 0.7 GB at R-MAT scale 20) come
// from a second process-wide buffer under the same policy: five allocations at the start of the phase and their hipFree at its end -- each
// waits for the device and unmaps hundreds of MB -- were 1.7 of the phase's 36 ms.  One user at a time; anyone else allocates as before.
struct TmpPool { void* p = nullptr; size_t bytes = 0; bool in_use = false; int device = -1; std::mutex m; };
static TmpPool& tmp_pool() { static TmpPool pool; return pool; }
static void* take_tmp(size_t need) {
  TmpPool& pool = tmp_pool();
  std::lock_guard<std::mutex> g(pool.m);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (pool.in_use || (pool.p && pool.device != dev)) return nullptr;
  if (pool.bytes < need) {
    if (pool.p) { (void)hipFree(pool.p); pool.p = nullptr; pool.bytes = 0; }
    if (hipMalloc(&pool.p, need) != hipSuccess) { (void)hipGetLastError(); pool.p = nullptr; return nullptr; }
    pool.bytes = need; pool.device = dev;
  }
  pool.in_use = true;
  return pool.p;
}
static void give_tmp() { TmpPool& pool = tmp_pool(); std::lock_guard<std::mutex> g(pool.m); pool.in_use = false; }
int release_bitmap_pool();
int release_bitmap_pool() {
  release_small_pool();
  {
    TmpPool& tp = tmp_pool();
    std::lock_guard<std::mutex> g(tp.m);
    if (!tp.in_use && tp.p) { (void)hipFree(tp.p); tp.p = nullptr; tp.bytes = 0; }
  }
  BmPool& pool = bm_pool();
  std::lock_guard<std::mutex> g(pool.m);
  if (!pool.in_use && pool.p) { (void)hipFree(pool.p); pool.p = nullptr; pool.bytes = 0; }
  return KKAMD_OK;
}
// `need` bytes for a handle's store: the pool's buffer when it is free (grown when it is less than half of what is wanted), else
// an allocation of the handle's own.  Returns the bytes obtained (possibly fewer than asked for: the caller lowers its row cap).
static size_t take_bitmap_store(kkamd_spgemm_handle* h, size_t need) {
  if (g_spgemm.store_cap_mb > 0 && need > (size_t)g_spgemm.store_cap_mb << 20) need = (size_t)g_spgemm.store_cap_mb << 20;
  BmPool& pool = bm_pool();
  std::lock_guard<std::mutex> g(pool.m);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (!pool.in_use && (pool.p == nullptr || pool.device == dev)) {      // the pool belongs to the device that filled it first
    if (pool.p == nullptr && pool.released_once) pool.sticky = true;      // back for more after a release: a repeat user (see BmPool)
    if (pool.bytes < need / 2 + 1) {
      if (pool.p) { (void)hipFree(pool.p); pool.p = nullptr; pool.bytes = 0; }
      if (hipMalloc(&pool.p, need) != hipSuccess) { (void)hipGetLastError(); pool.p = nullptr; return 0; }
      pool.bytes = need; pool.device = dev;
      if (need > pool.high_water) pool.high_water = need;
    }
    pool.in_use = true; h->bm_pooled = true; h->d_bm_store = pool.p;
    return pool.bytes < need ? pool.bytes : need;
  }
  if (hipMalloc(&h->d_bm_store, need) != hipSuccess) { (void)hipGetLastError(); h->d_bm_store = nullptr; return 0; }
  h->bm_pooled = false;
  return need;
}
static void free_bitmap_store(kkamd_spgemm_handle* h) {
  if (h->d_bm_store) {
    if (h->bm_pooled) { BmPool& pool = bm_pool(); std::lock_guard<std::mutex> g(pool.m); pool.in_use = false; }
    else (void)hipFree(h->d_bm_store);
  }
  h->bm_pooled = false;
  if (h->d_unit_block) {                                  // the unit arrays are pieces of one allocation (row_slot among them)
    (void)hipFree(h->d_unit_block); h->d_unit_block = nullptr; h->d_row_slot = nullptr;
  } else {
    if (h->d_unit_perm) (void)hipFree(h->d_unit_perm);
    if (h->d_ucnt) (void)hipFree(h->d_ucnt);
    if (h->d_ucoff) (void)hipFree(h->d_ucoff);
    if (h->d_uoff) (void)hipFree(h->d_uoff);
    if (h->d_heads) (void)hipFree(h->d_heads);
  }
  h->d_unit_perm = nullptr; h->d_ucnt = nullptr; h->d_ucoff = nullptr; h->d_uoff = nullptr; h->d_heads = nullptr; h->n_heads = 0;
  h->unit_mode = false; h->unit_rows = 0; h->unit_rows_kept = 0; h->unit_bitmaps = 0;
  if (h->d_row_slot) (void)hipFree(h->d_row_slot);
  if (h->d_bm_counter) (void)hipFree(h->d_bm_counter);
  if (h->d_emit_perm) (void)hipFree(h->d_emit_perm);
  if (h->d_emit_perm2) (void)hipFree(h->d_emit_perm2);
  if (h->d_emit_perm3) (void)hipFree(h->d_emit_perm3);
  h->d_emit_perm3 = nullptr; h->emit3_src = nullptr; h->emit3_n = 0; h->n_emit_sort = 0;
  if (h->d_pool_off) (void)hipFree(h->d_pool_off);
  h->d_bm_store = nullptr; h->d_row_slot = nullptr; h->d_bm_counter = nullptr; h->d_emit_perm = nullptr;
  h->d_emit_perm2 = nullptr; h->d_pool_off = nullptr; h->d_ent_pool = nullptr; h->pool_cap = 0; h->pool_used = 0; h->n_emit_pooled = 0;
  h->bm_cap = 0; h->bm_stored = 0; h->bm_words = 0; h->n_emit_stored = 0;
}
static double words_mb(int words) { return (double)words * 8.0 / 1048576.0; }

// the rows of list[0 .. n) whose size (sizes[row]) is at most cnt_max are moved to its front (the others follow in reverse order); *n_small = how many
template <class OffT>
static int split_list_by_size(int32_t* list, int64_t n, const OffT* rmA, const int64_t* sizes, int64_t cnt_max, int64_t* n_small, hipStream_t st) {
  *n_small = 0;
  if (n <= 0) return KKAMD_OK;
  DevBuf tmp_b, cnt_b;
  unsigned long long h_cnt[2] = {0, 0};
  KK_HIP(tmp_b.alloc(sizeof(int32_t) * (size_t)n));
  KK_HIP(cnt_b.alloc(2 * sizeof(unsigned long long)));
  int32_t* d_tmp = tmp_b.as<int32_t>(); unsigned long long* d_cnt = cnt_b.as<unsigned long long>();
  KK_HIP(hipMemsetAsync(d_cnt, 0, 2 * sizeof(unsigned long long), st));
  KK_HIP(hipMemcpyAsync(d_tmp, list, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
  KK_LAUNCH((spgemm_split_small_kernel<OffT>), (unsigned)ceil_div(n, kBlock), kBlock, 0, st, n, (const int32_t*)d_tmp, rmA, sizes, (int64_t)INT64_MAX, cnt_max, list, d_cnt);
  KK_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  *n_small = (int64_t)h_cnt[0];
  return KKAMD_OK;
}

// hipFree waits for the whole device: a temporary freed in the middle of a phase would wait for the kernels running beside it on the
// phase's second stream.  Such temporaries are handed to a Deferred list and freed when the phase is over.
struct Deferred {
  std::vector<void*> ptrs;
  void take(DevBuf& b) { if (b.p) ptrs.push_back(b.release()); }
  ~Deferred() { for (void* q : ptrs) (void)hipFree(q); }
};
// list[0 .. n) reordered by sizes[row], largest first (quarter-octave classes; see spgemm_size_hist_kernel)
// scratch: 4 n + 4 (kSizeClasses + 1) bytes of the caller's (then nothing is allocated, nothing waited for)
static size_t order_list_scratch_bytes(int64_t n) { return sizeof(int32_t) * (size_t)n + sizeof(unsigned) * (kSizeClasses + 1) + 16; }
static int order_list_by_size(int32_t* list, int64_t n, const int64_t* sizes, hipStream_t st, Deferred* later = nullptr, void* scratch = nullptr) {
  if (n < 4096 || !g_spgemm.sort_rows) return KKAMD_OK;          // (a short list finishes in one wave of workgroups whatever its order)
  DevBuf tmp_b, hist_b;
  struct Hand { Deferred* d; DevBuf &a, &b; ~Hand() { if (d) { d->take(a); d->take(b); } } } hand{later, tmp_b, hist_b};
  int32_t* d_tmp; unsigned* d_hist;
  if (scratch) { d_hist = reinterpret_cast<unsigned*>(scratch); d_tmp = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(scratch) + ((sizeof(unsigned) * (kSizeClasses + 1) + 15) & ~(size_t)15)); }
  else {
    KK_HIP(tmp_b.alloc(sizeof(int32_t) * (size_t)n));
    KK_HIP(hist_b.alloc(sizeof(unsigned) * (kSizeClasses + 1)));
    d_tmp = tmp_b.as<int32_t>(); d_hist = hist_b.as<unsigned>();
  }
  KK_HIP(hipMemsetAsync(d_hist, 0, sizeof(unsigned) * (kSizeClasses + 1), st));
  KK_HIP(hipMemcpyAsync(d_tmp, list, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
  const int64_t nbk = ceil_div(n, kBlock);
  const unsigned grid = (unsigned)(nbk < 1024 ? nbk : 1024);
  KK_LAUNCH(spgemm_size_hist_kernel, grid, kBlock, 0, st, n, (const int32_t*)d_tmp, sizes, d_hist);
  KK_LAUNCH(spgemm_size_scan_kernel, 1, kSizeClasses, 0, st, d_hist);
  KK_LAUNCH(spgemm_size_scatter_kernel, grid, kBlock, 0, st, n, (const int32_t*)d_tmp, sizes, d_hist, list);
  if (!scratch && !later) KK_HIP(hipStreamSynchronize(st));          // the scratch buffers go out of scope
  return KKAMD_OK;
}

// the handle's second stream and its two events, created at first use; false when the runtime refuses
static bool ensure_aux(kkamd_spgemm_handle* h) {
  if (h->aux) return true;
  if (hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    if (h->aux) { (void)hipStreamDestroy(h->aux); h->aux = nullptr; }
    if (h->ev_fork) { (void)hipEventDestroy(h->ev_fork); h->ev_fork = nullptr; }
    if (h->ev_join) { (void)hipEventDestroy(h->ev_join); h->ev_join = nullptr; }
    return false;
  }
  return true;
}
// The dense class of the symbolic phase by units (spgemm_sym_unit_kernel).  *ran = false: the prerequisites do not hold (B with unsorted
// rows under more than one window, entries(B) not 16-byte aligned, fewer than four entries, more units than a grid holds, no memory
// for the indices) and the caller takes the one-workgroup-per-row kernel.
template <class OffT>
static int symbolic_units(kkamd_spgemm_handle* h, int64_t nrows, const int32_t* list, int64_t m, int64_t n, int64_t k, const OffT* rmA, const int32_t* entA,
                          const OffT* rmB, const int32_t* entB, int64_t nnzB, OffT* rmC, hipStream_t st, bool* ran, Deferred* later) {
  *ran = false;
  const auto t_start = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {                       // (verbose: where the host-side time of the phase goes; adds stream synchronisations)
    if (h->verbose < 2) return;
    (void)hipStreamSynchronize(st);
    KK_VERBOSE("\t\tunits %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  };
  int wb = g_spgemm.unit_bits;
  if (wb > kUnitBitsMax) wb = kUnitBitsMax;
  if (wb < 6) wb = 6;
  const int64_t nwin64 = ceil_div(k, (int64_t)1 << wb);
  const int64_t nnzA = h->nnzA;
  if (!g_spgemm.sym_units || nnzB < 4 || ((uintptr_t)entB % 16) != 0 || nrows <= 0 || nrows >= ((int64_t)1 << 30)) return KKAMD_OK;
  if (nwin64 > 1 && !h->b_sorted) return KKAMD_OK;
  if (nwin64 > 4096 || nrows * nwin64 >= ((int64_t)1 << 31)) return KKAMD_OK;
  const int nwin = (int)nwin64;
  const int64_t units = nrows * nwin;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return KKAMD_OK; }
  size_t pooled = 0;                                       // what the pool holds is not "used" memory -- when it is free and on this device
  { BmPool& pool = bm_pool(); std::lock_guard<std::mutex> g(pool.m); int dev_ = -1;
    if (!pool.in_use && pool.p && hipGetDevice(&dev_) == hipSuccess && dev_ == pool.device) pooled = pool.bytes; }
  const size_t widx_bytes = nwin > 1 ? sizeof(unsigned) * (size_t)(nwin + 1) * (size_t)n : 0;
  const size_t aw_bytes = sizeof(long long) * (size_t)(nwin + 1) * (size_t)nnzA;
  const size_t unit_bytes = (size_t)units * (8 + 4 + 4 + 8 + 4 + 8 + 32) + (size_t)nrows * 4 + (size_t)m * 4;
  if (widx_bytes + aw_bytes + unit_bytes > (free_b + pooled) / 8) return KKAMD_OK;
  free_bitmap_store(h);
  // temporaries of this phase (free themselves): the two indices, products and launch order of the units; kept for the numeric phase: heads, counts, offsets
  DevBuf widx_b, aw_b, uprod_b, ulist_b, soff_b, cnt_b;
  struct Hand { Deferred* d; DevBuf *b[6]; ~Hand() { if (d) for (DevBuf* x : b) d->take(*x); } } hand{later, {&widx_b, &aw_b, &uprod_b, &ulist_b, &soff_b, &cnt_b}};
  // ... from the process-wide buffer when it is free (take_tmp), else allocated here
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t soff_max = sizeof(long long) * (size_t)(units + 1 + scan_workspace_items(units + 1) + nrows + 1 + scan_workspace_items(nrows + 1)) + (size_t)units + (size_t)nrows;
  const size_t arena_need = al(aw_bytes) + al(widx_bytes) + al(sizeof(long long) * (size_t)units) + al(sizeof(int32_t) * (size_t)units) + al(64) + al(soff_max) + al(order_list_scratch_bytes(units));
  char* arena = h->tmp_taken ? nullptr : reinterpret_cast<char*>(take_tmp(arena_need));
  size_t apos = 0;
  auto carve = [&](size_t b) -> void* { void* q = arena + apos; apos += al(b); return q; };
  void *a_aw = nullptr, *a_wx = nullptr, *a_up = nullptr, *a_ul = nullptr, *a_cnt = nullptr, *a_so = nullptr, *a_ord = nullptr;
  if (arena) {
    h->tmp_taken = true;
    a_aw = carve(aw_bytes); a_wx = carve(widx_bytes); a_up = carve(sizeof(long long) * (size_t)units); a_ul = carve(sizeof(int32_t) * (size_t)units);
    a_cnt = carve(64); a_so = carve(soff_max); a_ord = carve(order_list_scratch_bytes(units));
  }
  if ((!arena && (aw_b.alloc(aw_bytes) != hipSuccess || uprod_b.alloc(sizeof(long long) * (size_t)units) != hipSuccess || ulist_b.alloc(sizeof(int32_t) * (size_t)units) != hipSuccess ||
      cnt_b.alloc(8 * sizeof(unsigned long long)) != hipSuccess || (widx_bytes && widx_b.alloc(widx_bytes) != hipSuccess))) ||
      hipMalloc(&h->d_unit_block, al(sizeof(UnitHead) * (size_t)units) + al(sizeof(long long) * (size_t)units) + 2 * al(sizeof(unsigned) * (size_t)units) +
                                  al(sizeof(int32_t) * (size_t)nrows) + al(sizeof(int32_t) * (size_t)m)) != hipSuccess) {
    (void)hipGetLastError(); h->d_unit_block = nullptr; free_bitmap_store(h); return KKAMD_OK;
  }
  {
    char* q = reinterpret_cast<char*>(h->d_unit_block);
    h->d_heads = reinterpret_cast<UnitHead*>(q); q += al(sizeof(UnitHead) * (size_t)units);        // (room for every unit; those with products get a head)
    h->d_uoff = reinterpret_cast<long long*>(q); q += al(sizeof(long long) * (size_t)units);
    h->d_ucnt = reinterpret_cast<unsigned*>(q); q += al(sizeof(unsigned) * (size_t)units);
    h->d_ucoff = reinterpret_cast<unsigned*>(q); q += al(sizeof(unsigned) * (size_t)units);
    h->d_unit_perm = reinterpret_cast<int32_t*>(q); q += al(sizeof(int32_t) * (size_t)nrows);
    h->d_row_slot = reinterpret_cast<int32_t*>(q);
  }
  unsigned* d_wx = arena ? (unsigned*)a_wx : widx_b.as<unsigned>(); long long* d_aw = arena ? (long long*)a_aw : aw_b.as<long long>();
  long long* d_up = arena ? (long long*)a_up : uprod_b.as<long long>(); int32_t* d_ul = arena ? (int32_t*)a_ul : ulist_b.as<int32_t>();
  unsigned long long* d_cnt = arena ? (unsigned long long*)a_cnt : cnt_b.as<unsigned long long>();
  KK_HIP(hipMemcpyAsync(h->d_unit_perm, list, sizeof(int32_t) * (size_t)nrows, hipMemcpyDeviceToDevice, st));     // the numeric phase bins the rows again in d_perm
  KK_HIP(hipMemsetAsync(h->d_uoff, 0xFF, sizeof(long long) * (size_t)units, st));
  KK_HIP(hipMemsetAsync(h->d_ucnt, 0, sizeof(unsigned) * (size_t)units, st));
  KK_HIP(hipMemsetAsync(h->d_row_slot, 0xFF, sizeof(int32_t) * (size_t)m, st));
  KK_HIP(hipMemsetAsync(d_cnt, 0, 8 * sizeof(unsigned long long), st));
  const int32_t* d_perm_u = h->d_unit_perm;
  if (nwin > 1) KK_LAUNCH((spgemm_bidx_kernel<OffT>), (unsigned)ceil_div((int64_t)(nwin + 1) * n, kBlock), kBlock, 0, st, n, nwin, wb, rmB, entB, d_wx);
  KK_LAUNCH((spgemm_aw_kernel<OffT>), (unsigned)ceil_div(nnzA, kBlock), kBlock, 0, st, nnzA, entA, rmB, n, nwin, (const unsigned*)d_wx, d_aw);
  KK_LAUNCH((spgemm_uprod_kernel<OffT>), (unsigned)ceil_div(nrows * nwin, kBlock / 64), kBlock, 0, st, nrows, d_perm_u, rmA, nnzA, nwin, (const long long*)d_aw, d_up);
  KK_LAUNCH(spgemm_unit_compact_kernel, (unsigned)ceil_div(units, kBlock), kBlock, 0, st, units, (const long long*)d_up, d_ul, d_cnt);
  lap("allocations + indices + compaction");
  unsigned long long h_n = 0;
  KK_HIP(hipMemcpyAsync(&h_n, d_cnt, sizeof h_n, hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  const int64_t nu = (int64_t)h_n;                           // units with products
  h->unit_mode = true; h->unit_nwin = nwin; h->unit_wb = wb; h->unit_rows = nrows; h->n_heads = nu;
  h->last_units = nu;
  if (nu == 0) { *ran = true; return KKAMD_OK; }
  int rc;
  if ((rc = order_list_by_size(d_ul, nu, (const int64_t*)d_up, st, later, a_ord))) return rc;        // heaviest units first
  // where every unit's structure goes: a prefix sum over min(4 products, bitmap bytes); the store is at most 0.225 of the free HBM (an eighth
  // for bitmaps and a tenth for lists until round 5).  Room is given out by ROW, heaviest rows first (a row's structure is of use only when
  // all its units keep theirs): the rows past it keep nothing and walk their products again in the numeric phase
  const int64_t win_cols = nwin > 1 ? ((int64_t)1 << wb) : k;
  const int words = (int)ceil_div(win_cols, (int64_t)64);
  const int bm_words = (words + 15) & ~15;
  const long long bm_bytes = (long long)words * 8;
  long long budget = 0, total_need = 0;
  const int64_t so_items = nu + 1 + scan_workspace_items(nu + 1), rb_items = nrows + 1 + scan_workspace_items(nrows + 1);
  if (!arena) KK_HIP(soff_b.alloc(sizeof(long long) * (size_t)(so_items + rb_items) + (size_t)nu + (size_t)nrows));
  long long* d_so = arena ? (long long*)a_so : soff_b.as<long long>();
  long long* d_rb = d_so + so_items;
  unsigned char* d_drop = reinterpret_cast<unsigned char*>(d_rb + rb_items);
  unsigned char* d_rnone = d_drop + nu;
  KK_LAUNCH(spgemm_unit_rowbytes_kernel, (unsigned)ceil_div(nrows + 1, kBlock), kBlock, 0, st, nrows, nwin, (const long long*)d_up, bm_bytes, g_spgemm.keep_lists, d_rb, d_rnone);
  if ((rc = exclusive_scan_inplace<long long>(d_rb, nrows + 1, st, d_rb + nrows + 1))) return rc;
  if (g_spgemm.keep_bitmaps && k >= 64) {
    KK_HIP(hipMemcpyAsync(&total_need, d_rb + nrows, sizeof(long long), hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
    size_t want = (size_t)((double)(free_b + pooled) * 0.225);
    if ((long long)want > total_need) want = (size_t)total_need;
    want = (want + 255) & ~(size_t)255;
    const size_t got = want ? take_bitmap_store(h, want) : 0;
    budget = (long long)got;
  }
  KK_LAUNCH(spgemm_unit_ssize_kernel, (unsigned)ceil_div(nu + 1, kBlock), kBlock, 0, st, nu, (const int32_t*)d_ul, (const long long*)d_up, bm_bytes, nwin,
            (const long long*)d_rb, (const unsigned char*)d_rnone, budget, d_so, d_drop);
  if ((rc = exclusive_scan_inplace<long long>(d_so, nu + 1, st, d_so + nu + 1))) return rc;
  lap("order, sizes, store");
  UnitHead* d_hd = h->d_heads;
  KK_LAUNCH((spgemm_unit_heads_kernel<OffT>), (unsigned)ceil_div(nu, kBlock), kBlock, 0, st, nu, (const int32_t*)d_ul, (const long long*)d_up, (const long long*)d_so,
            bm_bytes, (const unsigned char*)d_drop, nwin, d_perm_u, rmA, d_hd, h->d_uoff, d_cnt + 1);
  char* d_store = (char*)h->d_bm_store;
  unsigned* d_uc = h->d_ucnt;
#ifndef KK_EMU
#define KK_UNIT_ATTR(NTT) KK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spgemm_sym_unit_kernel<OffT, NTT, KK_UQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)bm_words * 8 + sizeof(UnitScratch<NTT>))))
#else
#define KK_UNIT_ATTR(NTT) (void)0
#endif
#define KK_UNIT(NTT)                                                                                                                                        \
  do {                                                                                                                                                      \
    KK_UNIT_ATTR(NTT);                                                                                                                                      \
    KK_LAUNCH((spgemm_sym_unit_kernel<OffT, NTT, KK_UQ>), (unsigned)nu, NTT, (size_t)bm_words * 8 + sizeof(UnitScratch<NTT>), st, (const UnitHead*)d_hd, nwin, wb, k, nnzA, \
              (const long long*)d_aw, entB, nnzB, rmC, d_uc, d_store, bm_words KK_DBG_ARG);                                                                 \
  } while (0)
  lap("heads");
  KK_UNIT(256);
  lap("unit kernel");
#undef KK_UNIT
#undef KK_UNIT_ATTR
  {
    const long long* d_uo = h->d_uoff; unsigned* d_co = h->d_ucoff; int32_t* d_ur = h->d_row_slot;
    KK_LAUNCH(spgemm_unit_rows_kernel, (unsigned)ceil_div(nrows, kBlock), kBlock, 0, st, nrows, d_perm_u, nwin, (const unsigned*)d_uc, d_uo, d_co, d_ur, d_cnt + 4);
  }
  unsigned long long h_c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  KK_HIP(hipMemcpyAsync(h_c, d_cnt, sizeof h_c, hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));                              // (also: the temporaries go out of scope)
  h->unit_bitmaps = (int64_t)h_c[1]; h->pool_used = (int64_t)h_c[2]; h->unit_rows_kept = (int64_t)h_c[4];
  h->bm_words = words;
  h->last_unit_bitmaps = h->unit_bitmaps; h->last_unit_rows_kept = h->unit_rows_kept;
  h->bm_stored = h->unit_rows_kept;                              // kkamd_spgemm_get 13: rows whose structure the symbolic phase holds
  if (h->verbose)
    KK_VERBOSE("\tkkamd spgemm symbolic: dense class by units: %lld rows x %d windows of 2^%d columns = %lld units with products; kept for the numeric phase: %lld unit bitmaps, "
               "%.1f MB of structure in all (%.1f MB wanted, %lld units without room), %lld rows complete\n",
               (long long)nrows, nwin, wb, (long long)nu, (long long)h->unit_bitmaps, 1e-6 * (double)h->pool_used, 1e-6 * (double)total_need, (long long)h_c[3], (long long)h->unit_rows_kept);
  if (h->unit_rows_kept == 0) free_bitmap_store(h);              // nothing to hand over: the store and the unit arrays go back at once
  *ran = true;
  return KKAMD_OK;
}

template <class OffT>
static int symbolic_typed(kkamd_spgemm_handle* h, int64_t m, int64_t n, int64_t k, const void* rmA_, const int32_t* entA,
                          const void* rmB_, const int32_t* entB, void* rmC_, int64_t nnzB, int64_t* c_nnz, hipStream_t st) {
  const OffT* rmA = (const OffT*)rmA_;
  const OffT* rmB = (const OffT*)rmB_;
  OffT* rmC       = (OffT*)rmC_;
  const auto t_start = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (h->verbose < 2) return;
    (void)hipStreamSynchronize(st);
    KK_VERBOSE("\t\tsymbolic %-25s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  };
  KK_HIP(hipMemsetAsync(rmC, 0, sizeof(OffT) * (size_t)(m + 1), st));
  if (!ensure_small(h)) return fail(KKAMD_ERR_ALLOC, "kkamd_spgemm_symbolic: out of memory for the handle's counters");
  // the process-wide buffer of temporaries goes back when the phase is over and its kernels have run (both streams)
  struct TmpGuard { kkamd_spgemm_handle* h; hipStream_t st; ~TmpGuard() { if (h->tmp_taken) { (void)hipStreamSynchronize(st); if (h->aux) (void)hipStreamSynchronize(h->aux); give_tmp(); h->tmp_taken = false; } } } tmp_guard{h, st};
  // Without B compression (the default) nothing between the row flops and the bins depends on the host: flops, sortedness of B and the bin
  // counts are queued together and read back in one copy.
  const bool merged = !h->compression;
  const BinLimits& symL = g_spgemm.sym_large ? kSymLimits : kSymLimitsNoLarge;
  DevBuf stats_b;                            // frees itself on every return
  unsigned long long* d_stats = h->d_small + kSmallStats;
  if (merged) KK_HIP(hipMemsetAsync(h->d_small, 0, kSmallSlots * sizeof(unsigned long long), st));
  else {
    KK_HIP(stats_b.alloc(2 * sizeof(unsigned long long)));
    d_stats = stats_b.as<unsigned long long>();
    KK_HIP(hipMemsetAsync(d_stats, 0, 2 * sizeof(unsigned long long), st));
  }
  // rows of A above kFlopsLong entries are listed by the first kernel and walked by the second, a workgroup each (the list borrows d_perm: the bins come later)
  int32_t* d_long = h->d_perm; unsigned long long* d_long_cnt = h->d_small + kSmallLong;
  const int64_t n_long_cap = h->nnzA / kFlopsLong + 1;
  const unsigned n_long_max = (unsigned)(n_long_cap < m ? n_long_cap : m);
  {
    const int64_t nbk = ceil_div(m * 8, kBlock);
    if (!merged) KK_HIP(hipMemsetAsync(d_long_cnt, 0, sizeof(unsigned long long), st));
    KK_LAUNCH((spgemm_flops_kernel<OffT>), (unsigned)(nbk < 4096 ? nbk : 4096), kBlock, 0, st, m, rmA, entA, rmB, h->d_sizes, d_stats, d_long, d_long_cnt);
    KK_LAUNCH((spgemm_flops_long_kernel<OffT>), n_long_max, kBlock, 0, st, (const int32_t*)d_long, (const unsigned long long*)d_long_cnt, rmA, entA, rmB, h->d_sizes, d_stats);
  }
  unsigned long long h_stats[2] = {0, 0};
  if (merged) {
    int* d_flag = reinterpret_cast<int*>(h->d_small + kSmallFlag);
    if (n > 0 && nnzB > 1) {
      const int64_t nbk = ceil_div(n * 8, kBlock);
      KK_LAUNCH((rows_sorted_kernel<OffT>), (unsigned)(nbk < 8192 ? nbk : 8192), kBlock, 0, st, n, rmB, entB, d_flag);
      KK_LAUNCH((rows_sorted_long_kernel<OffT>), (unsigned)ceil_div(n, kBlock), kBlock, 0, st, n, rmB, entB, d_flag);
    }
    const int64_t* d_fl = h->d_sizes;
    KK_LAUNCH(spgemm_bin_count_kernel, (unsigned)ceil_div(m, kBlock), kBlock, 0, st, m, d_fl, k, symL, h->d_small + kSmallBins);
    KK_HIP(hipMemcpyAsync(h->h_small, h->d_small, kSmallSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
    h_stats[0] = h->h_small[kSmallStats]; h_stats[1] = h->h_small[kSmallStats + 1];
    h->b_sorted = *reinterpret_cast<const int*>(h->h_small + kSmallFlag) == 0;
  } else {
    KK_HIP(hipMemcpyAsync(h_stats, d_stats, sizeof h_stats, hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
  }
  h->mults = (int64_t)h_stats[0]; h->max_row_flops = (int64_t)h_stats[1];
  h->sg_log2 = pick_sg_log2(nnzB, n); h->nnzB = nnzB;
  lap("row flops");
  if (h->d_flop_cls) { (void)hipFree(h->d_flop_cls); h->d_flop_cls = nullptr; }
  // (only when some row can land in the numeric phase's dense bin at all: more products than the wave kernel's table takes entries)
  if (g_spgemm.emit_sort && h->max_row_flops > (int64_t)(kWaveTable / 2) && hipMalloc((void**)&h->d_flop_cls, sizeof(int32_t) * (size_t)m) == hipSuccess) {
    int32_t* d_cls = h->d_flop_cls; const int64_t* d_fl = h->d_sizes;
    KK_LAUNCH(spgemm_flop_class_kernel, (unsigned)ceil_div(m, kBlock), kBlock, 0, st, m, d_fl, d_cls);
  } else { (void)hipGetLastError(); h->d_flop_cls = nullptr; }
  int rc;
  // sortedness of B decides how the numeric phase handles dense rows, and whether B can be compressed
  if (!merged) {
    DevBuf flag; int h_flag = 0;
    KK_HIP(flag.alloc(sizeof(int)));
    int* d_flag = flag.as<int>();
    KK_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), st));
    if (n > 0 && nnzB > 1) {
      const int64_t nbk = ceil_div(n * 8, kBlock);
      KK_LAUNCH((rows_sorted_kernel<OffT>), (unsigned)(nbk < 8192 ? nbk : 8192), kBlock, 0, st, n, rmB, entB, d_flag);
      KK_LAUNCH((rows_sorted_long_kernel<OffT>), (unsigned)ceil_div(n, kBlock), kBlock, 0, st, n, rmB, entB, d_flag);
    }
    KK_HIP(hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
    h->b_sorted = h_flag == 0;
  }
  lap("flop classes + sortedness");
  // B compression (a18): 32-column sets with bit masks; kept when it removes >= 15 % of the symbolic insertions
  DevBuf setB_b, maskB_b, endB_b;
  h->compressed = false; h->compressed_mults = h->mults;
  if (h->compression && h->b_sorted && nnzB > 0 && h->mults > 0 &&
      setB_b.alloc(sizeof(int32_t) * (size_t)nnzB) == hipSuccess && maskB_b.alloc(sizeof(unsigned) * (size_t)nnzB) == hipSuccess &&
      endB_b.alloc(sizeof(OffT) * (size_t)n) == hipSuccess) {
    int32_t* d_set = setB_b.as<int32_t>(); unsigned* d_mask = maskB_b.as<unsigned>(); OffT* d_end = endB_b.as<OffT>();
    KK_HIP(hipMemsetAsync(d_mask, 0, sizeof(unsigned) * (size_t)nnzB, st));
    const int64_t nbk = ceil_div(n * 8, kBlock);
    KK_LAUNCH((spgemm_compress_kernel<OffT>), (unsigned)(nbk < 65536 ? nbk : 65536), kBlock, 0, st, n, rmB, entB, d_set, d_mask, d_end);
    KK_HIP(hipMemsetAsync(d_stats, 0, 2 * sizeof(unsigned long long), st));
    DevBuf cflops;                                              // compressed insertions per row, into a scratch copy first
    if (cflops.alloc(sizeof(int64_t) * (size_t)m) == hipSuccess) {
      int64_t* d_cf = cflops.as<int64_t>();
      const int64_t nbf = ceil_div(m * 8, kBlock);
      KK_HIP(hipMemsetAsync(d_long_cnt, 0, sizeof(unsigned long long), st));
      KK_LAUNCH((spgemm_flops_kernel<OffT>), (unsigned)(nbf < 4096 ? nbf : 4096), kBlock, 0, st, m, rmA, entA, rmB, d_cf, d_stats, d_long, d_long_cnt, (const OffT*)d_end);
      KK_LAUNCH((spgemm_flops_long_kernel<OffT>), n_long_max, kBlock, 0, st, (const int32_t*)d_long, (const unsigned long long*)d_long_cnt, rmA, entA, rmB, d_cf, d_stats, (const OffT*)d_end);
      KK_HIP(hipMemcpyAsync(h_stats, d_stats, sizeof h_stats, hipMemcpyDeviceToHost, st));
      KK_HIP(hipStreamSynchronize(st));
      h->compressed_mults = (int64_t)h_stats[0];
      if (h->compression == 2 || (double)h->compressed_mults <= h->compression_cutoff * (double)h->mults) {
        KK_HIP(hipMemcpyAsync(h->d_sizes, d_cf, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToDevice, st));    // bin by the compressed work
        h->compressed = true;
      }
    }
  } else {
    (void)hipGetLastError();
  }
  stats_b.reset(); d_stats = nullptr;
  if (h->verbose)
    KK_VERBOSE("\tkkamd spgemm symbolic: m %lld n %lld k %lld, multiplications %lld (max per row %lld), B %s, compression %s (%.3f of the work)\n",
           (long long)m, (long long)n, (long long)k, (long long)h->mults, (long long)h->max_row_flops, h->b_sorted ? "sorted" : "unsorted",
           h->compressed ? "kept" : (h->compression ? "dropped" : "off"), h->mults ? (double)h->compressed_mults / (double)h->mults : 1.0);

  BinOffsets off;
  const int sg = h->sg_log2;
  auto nb = [&](int b) { return off.off[b + 1] - off.off[b]; };
  if (h->compressed) {
    const int32_t* setB = setB_b.as<int32_t>(); const unsigned* maskB = maskB_b.as<unsigned>(); const OffT* endB = endB_b.as<OffT>();
    if ((rc = make_bins(m, h->d_sizes, (k + 31) / 32, kSymLimitsC, h->d_perm, &off, st))) return rc;     // a C row has at most k / 32 sets
    if (nb(1)) KK_LAUNCH((spgemm_symc_wave_kernel<OffT>), (unsigned)ceil_div(nb(1), kBlock / 64), kBlock, 0, st, nb(1),
                         (const int32_t*)(h->d_perm + off.off[1]), rmA, entA, rmB, endB, setB, maskB, rmC);
    if (nb(2)) KK_LAUNCH((spgemm_symc_block_kernel<OffT, kSymBlkSC, kBlock>), (unsigned)nb(2), kBlock, (size_t)kSymBlkSC * 8, st, nb(2),
                         (const int32_t*)(h->d_perm + off.off[2]), rmA, entA, rmB, endB, setB, maskB, rmC);
    if (nb(3)) {
#ifndef KK_EMU
      KK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spgemm_symc_block_kernel<OffT, kSymBlkLC, kDenseBlock>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kSymBlkLC * 8));
#endif
      KK_LAUNCH((spgemm_symc_block_kernel<OffT, kSymBlkLC, kDenseBlock>), (unsigned)nb(3), kDenseBlock, (size_t)kSymBlkLC * 8, st, nb(3),
                (const int32_t*)(h->d_perm + off.off[3]), rmA, entA, rmB, endB, setB, maskB, rmC);
    }
    if (nb(4)) {
      if ((rc = launch_dense_cols<OffT, false>(nb(4), h->d_perm + off.off[4], rmA, entA, rmB, setB, rmC, (const OffT*)nullptr,
                                               (int32_t*)nullptr, k, (int64_t)0, sg, st, endB, maskB))) return rc;
    }
  } else {
    if ((rc = make_bins(m, h->d_sizes, k, symL, h->d_perm, &off, st, merged ? h->d_small + kSmallBins : nullptr, merged ? h->h_small + kSmallBins : nullptr, merged))) return rc;   // a C row cannot exceed k columns
    // When there is a dense class, the kernels of the other rows go to a second stream and run beside it (they are bound by their
    // latency chains and leave most of the chip idle: 5 of the symbolic phase's 40 ms on R-MAT scale 20 when they ran first on the one stream).
    // The caller's stream waits for them before the counts are scanned.
    Deferred later;                                            // (declared before the join below: freed after the second stream has been waited for)
    hipStream_t sx = st;
    bool forked = false;
    int64_t nq = 0;
    int32_t* wlist = h->d_perm + off.off[1];
    if (nb(1)) {
      // the rows of this bin with at most kQuadFlops products share waves four at a time; when the product mixes them with larger rows
      // the bin's list is split first (small rows to the front)
      if (g_spgemm.quad_rows == 2 || (g_spgemm.quad_rows == 1 && h->max_row_flops <= (int64_t)kQuadFlops)) nq = nb(1);
      else if (g_spgemm.quad_rows == 1 && (rc = split_list_by_size<OffT>(wlist, nb(1), rmA, (const int64_t*)h->d_sizes, (int64_t)kQuadFlops, &nq, st))) return rc;
    }
    if (nb(4) && (nb(1) || nb(2) || nb(3))) {
      if (ensure_aux(h) && hipEventRecord(h->ev_fork, st) == hipSuccess && hipStreamWaitEvent(h->aux, h->ev_fork, 0) == hipSuccess) { sx = h->aux; forked = true; }
      else (void)hipGetLastError();
    }
    if (nb(1)) {
      if (nq)
        KK_LAUNCH((spgemm_sym_quad_kernel<OffT>), (unsigned)ceil_div(nq, 4 * (kBlock / 64)), kBlock, 0, sx, nq,
                  (const int32_t*)wlist, rmA, entA, rmB, entB, rmC, (const int64_t*)h->d_sizes);
      if (nb(1) - nq)
        KK_LAUNCH((spgemm_sym_wave_kernel<OffT>), (unsigned)ceil_div(nb(1) - nq, kBlock / 64), kBlock, 0, sx, nb(1) - nq,
                  (const int32_t*)(wlist + nq), rmA, entA, rmB, entB, rmC, (const int64_t*)h->d_sizes);
    }
    if (nb(2)) KK_LAUNCH((spgemm_sym_block_kernel<OffT, kSymBlkS, kBlock>), (unsigned)nb(2), kBlock, 0, sx, nb(2),
                         (const int32_t*)(h->d_perm + off.off[2]), rmA, entA, rmB, entB, rmC, sg);
    if (nb(3)) KK_LAUNCH((spgemm_sym_block_kernel<OffT, kSymBlkL, kDenseBlock>), (unsigned)nb(3), kDenseBlock, 0, sx, nb(3),
                         (const int32_t*)(h->d_perm + off.off[3]), rmA, entA, rmB, entB, rmC, sg);
    if (forked) {                                              // (recorded now; the caller's stream waits for it after the dense class)
      if (hipEventRecord(h->ev_join, h->aux) != hipSuccess) { (void)hipGetLastError(); KK_HIP(hipStreamSynchronize(h->aux)); forked = false; }
    }
    lap("bins + small-row kernels queued");
    struct Join { kkamd_spgemm_handle* h; hipStream_t st; bool on; ~Join() { if (on) { if (hipStreamWaitEvent(st, h->ev_join, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(h->aux); } } } } join{h, st, forked};
    if (nb(4)) {
      if ((rc = order_list_by_size(h->d_perm + off.off[4], nb(4), (const int64_t*)h->d_sizes, st, &later))) return rc;      // by products, largest first
      // keep the bitmaps of rows with at least k / 32 entries (the bitmap is then no larger than the row's entries) when one LDS
      // window covers the columns and an eighth of the free HBM holds them: R-MAT scale 20, 87 K rows (83 % of the products), 11 GB
      BitmapStore bs;
      free_bitmap_store(h);
      bool by_units = false;
      lap("class ordered");
      if ((rc = symbolic_units<OffT>(h, nb(4), h->d_perm + off.off[4], m, n, k, rmA, entA, rmB, entB, nnzB, rmC, st, &by_units, &later))) return rc;
      lap("dense class");
      if (by_units) {}
      else {
      if (g_spgemm.keep_bitmaps && k <= (int64_t)g_spgemm.win_bits && k >= 4096) {
        size_t free_b = 0, total_b = 0;
        const int words = (int)ceil_div(k, (int64_t)64);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
          size_t pooled = 0;                                       // what the pool holds is not "used" memory -- when it is free and on this device
          { BmPool& pool = bm_pool(); std::lock_guard<std::mutex> g(pool.m); int dev_ = -1;
            if (!pool.in_use && pool.p && hipGetDevice(&dev_) == hipSuccess && dev_ == pool.device) pooled = pool.bytes; }
          int64_t cap = (int64_t)((free_b + pooled) / 8) / ((int64_t)words * 8);
          unsigned long long list_bound = ~0ull;                   // entries the lists of the bin's rows can hold between them
          {
            // no more slots than rows that can qualify: a row's products bound its entries, so only rows of the bin with at least
            // k / 32 products can have k / 32 entries
            DevBuf qc;
            unsigned long long h_qq[2] = {0, 0};
            if (qc.alloc(2 * sizeof(unsigned long long)) == hipSuccess && hipMemsetAsync(qc.p, 0, 2 * sizeof(unsigned long long), st) == hipSuccess) {
              unsigned long long* d_q = qc.as<unsigned long long>(); const int32_t* d_bin = h->d_perm + off.off[4]; const int64_t* d_fl = h->d_sizes;
              KK_LAUNCH(spgemm_count_ge_kernel, (unsigned)ceil_div(nb(4), kBlock), kBlock, 0, st, nb(4), d_bin, d_fl, k / 32, k, d_q);
              if (hipMemcpyAsync(h_qq, d_q, sizeof h_qq, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); h_qq[0] = (unsigned long long)nb(4); h_qq[1] = ~0ull; }
            } else { (void)hipGetLastError(); h_qq[0] = (unsigned long long)nb(4); h_qq[1] = ~0ull; }
            if (cap > (int64_t)h_qq[0]) cap = (int64_t)h_qq[0];
            list_bound = h_qq[1];
          }
          if (cap > nb(4)) cap = nb(4);
          // behind the bitmaps, room for the ENTRY LISTS of the rows whose bitmap is not kept: what those lists can hold at most (4 bytes
          // per product of the bin's rows, a row capped at k), but no more than a tenth of the free HBM (the caller still has entries(C)
          // and values(C) to allocate between the phases)
          const size_t store_need = cap >= 1 ? (size_t)cap * (size_t)words * 8 : 0;
          size_t pool_need = g_spgemm.keep_lists ? (((size_t)((double)(free_b + pooled) / 10.0)) & ~(size_t)255) : 0;
          if (list_bound != ~0ull && (unsigned long long)pool_need / 4 > list_bound) pool_need = ((size_t)list_bound * 4 + 255) & ~(size_t)255;
          size_t got = cap >= 1 ? take_bitmap_store(h, store_need + pool_need) : 0;
          if (got == 0 && cap >= 1 && pool_need > 0) got = take_bitmap_store(h, store_need);      // no room for both: the bitmaps alone
          const size_t got_store = got < store_need ? got : store_need;
          cap = (int64_t)(got_store / ((size_t)words * 8));
          const int64_t pool_cap = (int64_t)((got - got_store) / sizeof(int32_t));
          if (cap >= 1 && hipMalloc((void**)&h->d_row_slot, sizeof(int32_t) * (size_t)m) == hipSuccess &&
              hipMalloc((void**)&h->d_bm_counter, 2 * sizeof(unsigned long long)) == hipSuccess &&
              hipMemsetAsync(h->d_row_slot, 0xFF, sizeof(int32_t) * (size_t)m, st) == hipSuccess && hipMemsetAsync(h->d_bm_counter, 0, 2 * sizeof(unsigned long long), st) == hipSuccess) {
            h->bm_cap = cap; h->bm_words = words;
            bs.words_out = (kk_u64*)h->d_bm_store; bs.row_slot = h->d_row_slot; bs.counter = h->d_bm_counter; bs.cap = cap; bs.min_count = k / 32; bs.words = words;
            if (pool_cap > 0 && hipMalloc((void**)&h->d_pool_off, sizeof(long long) * (size_t)m) == hipSuccess &&
                hipMemsetAsync(h->d_pool_off, 0xFF, sizeof(long long) * (size_t)m, st) == hipSuccess) {
              h->d_ent_pool = (int32_t*)((char*)h->d_bm_store + got_store); h->pool_cap = pool_cap;
              bs.pool = h->d_ent_pool; bs.pool_off = h->d_pool_off; bs.pool_cursor = h->d_bm_counter + 1; bs.pool_cap = pool_cap; bs.list_staged = g_spgemm.list_staged;
            } else (void)hipGetLastError();
          } else { (void)hipGetLastError(); free_bitmap_store(h); }
        } else (void)hipGetLastError();
      }
      if ((rc = launch_dense_cols<OffT, false>(nb(4), h->d_perm + off.off[4], rmA, entA, rmB, entB, rmC, (const OffT*)nullptr,
                                               (int32_t*)nullptr, k, nnzB, sg, st, nullptr, nullptr, bs))) return rc;
      if (bs.words) {
        unsigned long long h_c[2] = {0, 0};
        KK_HIP(hipMemcpyAsync(h_c, h->d_bm_counter, sizeof h_c, hipMemcpyDeviceToHost, st));
        KK_HIP(hipStreamSynchronize(st));
        const unsigned long long h_n = h_c[0];
        h->bm_stored = (int64_t)(h_n < (unsigned long long)h->bm_cap ? h_n : (unsigned long long)h->bm_cap);
        h->pool_used = bs.pool ? (int64_t)h_c[1] : 0;              // (the cursor runs past the capacity when rows did not fit: those rows have no list)
        if (h->verbose && bs.pool) KK_VERBOSE("\tkkamd spgemm symbolic: entry lists kept for the numeric phase: %.1f of %.1f MB\n", 4e-6 * (double)h->pool_used, 4e-6 * (double)h->pool_cap);
        if (h->bm_stored == 0 && h->pool_used == 0) free_bitmap_store(h);
        if (h->verbose) KK_VERBOSE("\tkkamd spgemm symbolic: bitmaps of %lld rows kept for the numeric phase (%.1f MB)\n", (long long)h->bm_stored, (double)h->bm_stored * words_mb(h->bm_words));
      }
      }
    }
  }
  if (h->verbose)
    KK_VERBOSE("\tkkamd spgemm symbolic bins (rows): empty %lld, wave %lld, block-small %lld, block-large %lld, bitmap %lld\n",
           (long long)nb(0), (long long)nb(1), (long long)nb(2), (long long)nb(3), (long long)nb(4));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(KKAMD_ERR_HIP, "spgemm symbolic launch failed: %s", hipGetErrorString(e));
  // sum (in 64 bits: a 32-bit row_map must not wrap silently) and maximum of the counts, then the scan with the handle's workspace: queued
  // together, one copy back
  {
    const size_t need = sizeof(OffT) * (size_t)scan_workspace_items(m + 1);
    if (h->scan_ws_bytes < need) {
      if (h->d_scan_ws) { (void)hipFree(h->d_scan_ws); h->d_scan_ws = nullptr; h->scan_ws_bytes = 0; }
      KK_HIP(hipMalloc(&h->d_scan_ws, need));
      h->scan_ws_bytes = need;
    }
  }
  KK_HIP(hipMemsetAsync(h->d_small + kSmallSum, 0, 2 * sizeof(unsigned long long), st));
  {
    const int64_t nbk = ceil_div(m, kBlock);
    KK_LAUNCH((sum_max_counts_kernel<OffT>), (unsigned)(nbk < 4096 ? nbk : 4096), kBlock, 0, st, m, (const OffT*)rmC, h->d_small + kSmallSum);
  }
  lap("before the scan");
  rc = exclusive_scan_inplace<OffT>(rmC, m + 1, st, (OffT*)h->d_scan_ws);
  unsigned long long total = 0;
  if (rc == KKAMD_OK) {
    hipError_t e1 = hipMemcpyAsync(h->h_small + kSmallSum, h->d_small + kSmallSum, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
    hipError_t e2 = hipStreamSynchronize(st);
    if (e1 != hipSuccess || e2 != hipSuccess) rc = fail(KKAMD_ERR_HIP, "spgemm symbolic failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    total = h->h_small[kSmallSum]; h->max_row_nnz = (int64_t)h->h_small[kSmallSum + 1];
    if (rc == KKAMD_OK && sizeof(OffT) == 4 && total > (unsigned long long)INT32_MAX)
      return fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_symbolic: nnz(C) = %llu overflows 32-bit offsets; use 64-bit offsets", total);
  }
  if (rc) return rc;
  *c_nnz = (int64_t)total;
  return KKAMD_OK;
}

template <class OffT, class VT>
static int numeric_typed(kkamd_spgemm_handle* h, int64_t m, int64_t k, const void* rmA_, const int32_t* entA, const void* valA_,
                         const void* rmB_, const int32_t* entB, const void* valB_, const void* rmC_, int32_t* entC,
                         void* valC_, hipStream_t st) {
  const OffT* rmA = (const OffT*)rmA_; const OffT* rmB = (const OffT*)rmB_; const OffT* rmC = (const OffT*)rmC_;
  const VT* valA  = (const VT*)valA_;  const VT* valB  = (const VT*)valB_;  VT* valC = (VT*)valC_;
  int rc;
  if (!h->numeric_bins_ready) {
    const int64_t nbk = ceil_div(m, kBlock);
    KK_LAUNCH((spgemm_rowsize_kernel<OffT>), (unsigned)(nbk < 65536 ? nbk : 65536), kBlock, 0, st, m, rmC, h->d_sizes);
    // SPGEMM_KK_DENSE (a21, sparse/impl/KokkosSparse_spgemm_impl_speed.hpp:28-150): every row accumulates into a k-wide dense
    // accumulator (here: in HBM, one per concurrently processed row) instead of an LDS hash table
    const bool dense_alg = h->algorithm == 1;
    h->dense_lds = h->b_sorted && !g_spgemm.force_unsorted && !dense_alg;
    if ((rc = make_bins(m, h->d_sizes, INT64_MAX, dense_alg ? kAllDense : (h->dense_lds ? kNumLimitsSorted : kNumLimits), h->d_perm, &h->num_off, st, ensure_small(h) ? h->d_small + kSmallBins : nullptr, h->h_small ? h->h_small + kSmallBins : nullptr, false))) return rc;
    h->n_dense_lds = 0; h->n_dense_hub_lds = 0;
    const int64_t nd = h->num_off.off[5] - h->num_off.off[4];
    if (nd > 0) {
      // dense bin -> [ A row <= kValLa | A row <= kHubLa | the rest ]; everything is "the rest" when B is not sorted
      DevBuf tmp_b, cnt_b;                     // free themselves on every early return below
      unsigned long long h_cnt[2] = {0, 0};
      KK_HIP(tmp_b.alloc(sizeof(int32_t) * (size_t)nd));
      KK_HIP(cnt_b.alloc(2 * sizeof(unsigned long long)));
      int32_t* d_tmp = tmp_b.as<int32_t>(); unsigned long long* d_cnt = cnt_b.as<unsigned long long>();
      int32_t* seg = h->d_perm + h->num_off.off[4];
      // the tail of the bin: rows for the column-block value kernel (dense rows, and rows with more lists than the flat shapes hold)
      h->n_dense_block = 0; h->cidx_ready = false; h->items_ready = false;
      if (g_spgemm.block && h->dense_lds && g_spgemm.val_kernel == 2 && k >= 64) {
        int wshift = 6;
        while ((1 << wshift) < g_spgemm.block_w) ++wshift;
        const int64_t nblk = ceil_div(k, (int64_t)1 << wshift);
        size_t free_b = 0, total_b = 0;
        const bool fits = nblk <= 65535 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && (size_t)(nblk + 1) * (size_t)h->n * 4 <= free_b / 16;
        (void)hipGetLastError();
        if (fits) {
          KK_HIP(hipMemsetAsync(d_cnt, 0, 2 * sizeof(unsigned long long), st));
          KK_HIP(hipMemcpyAsync(d_tmp, seg, sizeof(int32_t) * (size_t)nd, hipMemcpyDeviceToDevice, st));
          const int64_t cnt_min = (k * g_spgemm.block_min_pct + 99) / 100, cnt_floor = (k * g_spgemm.block_la_pct + 99) / 100;
          KK_LAUNCH((spgemm_split_block_kernel<OffT>), (unsigned)ceil_div(nd, kBlock), kBlock, 0, st, nd, (const int32_t*)d_tmp, rmA, (const int64_t*)h->d_sizes,
                    (int64_t)kValLa, cnt_min > 1 ? cnt_min : 1, cnt_floor > 1 ? cnt_floor : 1, seg, d_cnt);
          KK_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost, st));
          KK_HIP(hipStreamSynchronize(st));
          h->n_dense_block = (int64_t)h_cnt[1];
          if (h->n_dense_block * kDenseBlock >= ((int64_t)1 << 32)) h->n_dense_block = 0;      // (a grid dimension holds 2^32 work-items; the split stays: every row is still in the bin once)
          // the class's own index of C ((blocks + 1) x rows x 4 bytes) and its items (at most one 32-byte head and two 8-byte records per row and block)
          // must fit beside the index of B: a small block width or a large k with many class rows makes them GBs.  No room: the rows keep the windowed kernels.
          if ((double)(nblk + 1) * (double)h->n_dense_block * (4.0 + 48.0) + (double)(nblk + 1) * (double)h->n * 4.0 > (double)free_b / 8.0) h->n_dense_block = 0;
          if ((rc = order_list_by_size(seg + (nd - h->n_dense_block), h->n_dense_block, (const int64_t*)h->d_sizes, st))) return rc;
        }
      }
      int64_t lo = 0, len = nd - h->n_dense_block;
      // (B sorted: the hub value kernel takes A rows of any length, kHubLa entries per pass; hub_chunked 0 = rows above kHubLa accumulate in HBM)
      // flat value kernel (default): [A row <= kValLa | <= kValLa2 (the flat kernel, 1024 lists per pass) | the rest (hub kernel in passes)]
      h->hub_from_mid = h->dense_lds && g_spgemm.val_kernel == 2 && g_spgemm.val_mid && g_spgemm.hub_chunked;
      const int64_t la_max[2] = {g_spgemm.val_la < kValLa ? g_spgemm.val_la : kValLa,
                                 h->hub_from_mid ? (int64_t)g_spgemm.val_la2 : (g_spgemm.hub_chunked ? INT64_MAX : (int64_t)kHubLa)};
      int64_t first[2] = {0, 0};
      for (int pass = 0; pass < 2 && len > 0; ++pass) {
        KK_HIP(hipMemsetAsync(d_cnt, 0, 2 * sizeof(unsigned long long), st));
        KK_HIP(hipMemcpyAsync(d_tmp, seg + lo, sizeof(int32_t) * (size_t)len, hipMemcpyDeviceToDevice, st));
        KK_LAUNCH((spgemm_split_dense_kernel<OffT>), (unsigned)ceil_div(len, kBlock), kBlock, 0, st, len, (const int32_t*)d_tmp, rmA,
                  la_max[pass], h->dense_lds ? 0 : 1, seg + lo, d_cnt);
        KK_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost, st));
        KK_HIP(hipStreamSynchronize(st));
        first[pass] = (int64_t)h_cnt[0];
        lo += first[pass]; len -= first[pass];
      }
      h->n_dense_lds = first[0]; h->n_dense_hub_lds = first[1];
      h->n_dense_small = 0; h->n_dense_tiny = 0;
      if (first[0] > 0 && h->dense_lds && g_spgemm.val_kernel == 2) {
        // [ lightest shape | light shape | the 512-work-item shape ]
        const int64_t lim_la[2] = {kValLaTiny, kValLaSmall}, lim_cnt[2] = {g_spgemm.val_tiny_cnt, g_spgemm.val_small_cnt};
        int64_t got[2] = {0, 0}, at = 0, left = first[0];
        const int64_t* d_sz = h->d_sizes;
        for (int lvl = 0; lvl < 2 && left > 0; ++lvl) {
          if (lim_cnt[lvl] <= 0) continue;
          KK_HIP(hipMemsetAsync(d_cnt, 0, 2 * sizeof(unsigned long long), st));
          KK_HIP(hipMemcpyAsync(d_tmp, seg + at, sizeof(int32_t) * (size_t)left, hipMemcpyDeviceToDevice, st));
          KK_LAUNCH((spgemm_split_small_kernel<OffT>), (unsigned)ceil_div(left, kBlock), kBlock, 0, st, left, (const int32_t*)d_tmp, rmA, d_sz,
                    lim_la[lvl], lim_cnt[lvl], seg + at, d_cnt);
          KK_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost, st));
          KK_HIP(hipStreamSynchronize(st));
          got[lvl] = (int64_t)h_cnt[0]; at += got[lvl]; left -= got[lvl];
        }
        h->n_dense_tiny = got[0]; h->n_dense_small = got[1];
      }
      // every launch's rows largest first (the hub rows' (row, pass) items are ordered by passes further down)
      if (h->dense_lds && g_spgemm.val_kernel == 2) {
        const int64_t cuts[5] = {0, h->n_dense_tiny, h->n_dense_tiny + h->n_dense_small, first[0], first[0] + first[1]};
        for (int sgi = 0; sgi < 4; ++sgi)
          if ((rc = order_list_by_size(seg + cuts[sgi], cuts[sgi + 1] - cuts[sgi], (const int64_t*)h->d_sizes, st))) return rc;
      }
    }
    h->n_wave_quad = 0;
    {
      const int64_t nw = h->num_off.off[2] - h->num_off.off[1];
      if (nw > 0 && !dense_alg) {
        if (g_spgemm.quad_rows == 2 || (g_spgemm.quad_rows == 1 && h->max_row_nnz > 0 && h->max_row_nnz <= (int64_t)kQuadNnz)) h->n_wave_quad = nw;
        else if (g_spgemm.quad_rows == 1 && (rc = split_list_by_size<OffT>(h->d_perm + h->num_off.off[1], nw, rmA, (const int64_t*)h->d_sizes, (int64_t)kQuadNnz, &h->n_wave_quad, st))) return rc;
      }
    }
    h->numeric_bins_ready = true;
    if (h->d_emit_perm) { (void)hipFree(h->d_emit_perm); h->d_emit_perm = nullptr; h->n_emit_stored = 0; }
    if (h->d_emit_perm2) { (void)hipFree(h->d_emit_perm2); h->d_emit_perm2 = nullptr; h->n_emit_pooled = 0; }
    if (h->d_emit_perm3) { (void)hipFree(h->d_emit_perm3); h->d_emit_perm3 = nullptr; h->emit3_src = nullptr; h->emit3_n = 0; h->n_emit_sort = 0; }
    if (h->d_hub_items) { (void)hipFree(h->d_hub_items); h->d_hub_items = nullptr; h->n_hub_items = 0; }
    if (h->d_hub_multi) { (void)hipFree(h->d_hub_multi); h->d_hub_multi = nullptr; h->n_hub_multi = 0; }
  }
  const BinOffsets& off = h->num_off;
  const int sg = h->sg_log2;
  auto nb = [&](int b) { return off.off[b + 1] - off.off[b]; };
  if (h->verbose)
    KK_VERBOSE("\tkkamd spgemm numeric (%s): rows per kernel -- wave hash %lld, block hash small %lld, block hash large %lld, dense rows %lld "
           "(column blocks %lld, LDS value windows %lld, LDS hub windows %lld, HBM accumulator %lld)\n", h->algorithm == 1 ? "SPGEMM_KK_DENSE" : "SPGEMM_KK",
           (long long)nb(1), (long long)nb(2), (long long)nb(3), (long long)nb(4), (long long)h->n_dense_block, (long long)h->n_dense_lds, (long long)h->n_dense_hub_lds,
           (long long)(nb(4) - h->n_dense_block - h->n_dense_lds - h->n_dense_hub_lds));
  if (nb(1)) {
    const int64_t nq = h->n_wave_quad < nb(1) ? h->n_wave_quad : nb(1);
    const int32_t* wlist = h->d_perm + off.off[1];
    if (nq)
      KK_LAUNCH((spgemm_num_quad_kernel<OffT, VT>), (unsigned)ceil_div(nq, 4 * (kBlock / 64)), kBlock, 0, st, nq, wlist, rmA, entA, valA, rmB, entB, valB, rmC, entC, valC);
    if (nb(1) - nq)
      KK_LAUNCH((spgemm_num_wave_kernel<OffT, VT>), (unsigned)ceil_div(nb(1) - nq, kBlock / 64), kBlock, 0, st, nb(1) - nq, wlist + nq, rmA, entA, valA, rmB, entB, valB,
                rmC, entC, valC, sg);
  }
  if (nb(2)) KK_LAUNCH((spgemm_num_block_kernel<OffT, VT, kNumBlkS>), (unsigned)nb(2), kBlock, 0, st, nb(2),
                       (const int32_t*)(h->d_perm + off.off[2]), rmA, entA, valA, rmB, entB, valB, rmC, entC, valC, sg);
  if (nb(3)) KK_LAUNCH((spgemm_num_block_kernel<OffT, VT, kNumBlkL>), (unsigned)nb(3), kBlock, 0, st, nb(3),
                       (const int32_t*)(h->d_perm + off.off[3]), rmA, entA, valA, rmB, entB, valB, rmC, entC, valC, sg);
  DevBuf acc_b;
  VT* d_acc = nullptr;
  // The reference's plug-in contract fills entries(C) only while !are_entries_computed()
  // (sparse/tpls/KokkosSparse_spgemm_numeric_tpl_spec_decl.hpp:288-329): a repeated numeric call on the same handle and the same
  // C arrays (new values of A / B, same structure) keeps the entries the previous call wrote.  What is skipped is the separate
  // structure pass of the dense rows (the LDS bitmap kernel: 116 of 398 ms on R-MAT scale 20); the hash kernels of the short
  // rows emit entries and values in one pass and rewrite the same entries.
  const bool keep_entries = h->entries_valid && h->entC_ptr == (const void*)entC && h->rmC_ptr == rmC_;
  h->entries_reused = false;
  // entries(C) of dense-bin rows that have to walk their products: the rows with few products are sorted in LDS (spgemm_emit_sort_kernel),
  // the others go through the bitmap kernel.  The split of `list` is made once per handle and list.
  auto emit_walk = [&](int64_t n, const int32_t* list, int64_t win_cap) -> int {
    int64_t nsort = 0;
    h->sorted_used = 0;
    if (g_spgemm.emit_sort && h->d_flop_cls && h->algorithm == 0 && n > 0) {
      if (!h->d_emit_perm3 || h->emit3_src != list || h->emit3_n != n) {
        if (h->d_emit_perm3) { (void)hipFree(h->d_emit_perm3); h->d_emit_perm3 = nullptr; }
        DevBuf c4;
        KK_HIP(c4.alloc(2 * sizeof(unsigned long long)));
        KK_HIP(hipMemsetAsync(c4.p, 0, 2 * sizeof(unsigned long long), st));
        KK_HIP(hipMalloc((void**)&h->d_emit_perm3, sizeof(int32_t) * (size_t)n));
        int32_t* d_ep3 = h->d_emit_perm3; const int32_t* d_cls = h->d_flop_cls; unsigned long long* d_c4 = c4.as<unsigned long long>();
        KK_LAUNCH((spgemm_split_stored_kernel<int32_t>), (unsigned)ceil_div(n, kBlock), kBlock, 0, st, n, list, d_cls, d_ep3, d_c4);
        unsigned long long h_c4[2] = {0, 0};
        KK_HIP(hipMemcpyAsync(h_c4, c4.p, sizeof h_c4, hipMemcpyDeviceToHost, st));
        KK_HIP(hipStreamSynchronize(st));
        h->n_emit_sort = (int64_t)h_c4[0]; h->emit3_src = list; h->emit3_n = n;
      }
      nsort = h->n_emit_sort; list = h->d_emit_perm3;
      if (nsort) KK_LAUNCH((spgemm_emit_sort_kernel<OffT>), (unsigned)nsort, kBlock, 0, st, list, rmA, entA, rmB, entB, rmC, entC);
      h->sorted_used = nsort;
    }
    if (n - nsort) return launch_dense_cols<OffT, true>(n - nsort, list + nsort, rmA, entA, rmB, entB, (OffT*)nullptr, rmC, entC, k, h->nnzB, sg, st, nullptr, nullptr,
                                                        BitmapStore(), win_cap);
    return KKAMD_OK;
  };
  auto emit_units = [&](hipStream_t sq, int) {
    const UnitHead* d_hd = h->d_heads; const int32_t* d_ur = h->d_row_slot; const unsigned* d_uc = h->d_ucnt; const unsigned* d_co = h->d_ucoff;
    const char* d_st = (const char*)h->d_bm_store;
    const int64_t min_nnz = (h->dense_lds ? kNumLimitsSorted : kNumLimits).lim[3];
#define KK_EMIT_UNIT(NTT) KK_LAUNCH((spgemm_emit_unit_kernel<OffT, NTT>), (unsigned)h->n_heads, NTT, 0, sq, d_hd, d_ur, h->unit_nwin, h->unit_wb, k, d_uc, d_co, d_st, rmC, entC, min_nnz)
    KK_EMIT_UNIT(256);
#undef KK_EMIT_UNIT
  };
  if (nb(4)) {
    const int32_t* dperm = h->d_perm + off.off[4];
    // entries(C) of every dense row, column-sorted
    if (keep_entries) h->entries_reused = true;
    else if (h->unit_mode && h->unit_rows_kept > 0 && h->algorithm == 0) {
      // the symbolic phase's units: rows whose every unit kept its structure are written unit by unit (bitmaps, entry lists); the others walk their products
      if (!h->d_emit_perm) {
        DevBuf c2;
        KK_HIP(c2.alloc(3 * sizeof(unsigned long long)));
        KK_HIP(hipMemsetAsync(c2.p, 0, 3 * sizeof(unsigned long long), st));
        KK_HIP(hipMalloc((void**)&h->d_emit_perm, sizeof(int32_t) * (size_t)nb(4)));
        int32_t* d_ep = h->d_emit_perm; const int32_t* d_rs = h->d_row_slot; unsigned long long* d_c2 = c2.as<unsigned long long>();
        KK_LAUNCH((spgemm_split_stored_kernel<int32_t>), (unsigned)ceil_div(nb(4), kBlock), kBlock, 0, st, nb(4), dperm, d_rs, d_ep, d_c2);
        KK_LAUNCH(spgemm_count_flag_kernel, (unsigned)ceil_div(nb(4), kBlock), kBlock, 0, st, nb(4), dperm, d_rs, d_c2 + 2);
        unsigned long long h_c2[3] = {0, 0, 0};
        KK_HIP(hipMemcpyAsync(h_c2, c2.p, sizeof h_c2, hipMemcpyDeviceToHost, st));
        KK_HIP(hipStreamSynchronize(st));
        h->n_emit_stored = (int64_t)h_c2[0]; h->n_emit_pooled = (int64_t)h_c2[2];
      }
      const int64_t ns = h->n_emit_stored, nr = nb(4) - ns;
      if (ns && h->n_heads) {
        // (Measured and not kept: the units of the column-block rows first and the others' on the second stream beside the column-block value kernels --
        // the two 13 ms halves overlapped, and the value kernel beside the second took 26.3 instead of 13.5 ms: both live on the memory system.)
        emit_units(st, -1);
      }
      if (nr && (rc = emit_walk(nr, h->d_emit_perm + ns, (int64_t)g_spgemm.emit_win_bits))) return rc;
      h->bitmaps_used = ns; h->pooled_used = h->n_emit_pooled;
    }
    else if (h->d_bm_store && (h->bm_stored > 0 || h->pool_used > 0) && h->algorithm == 0) {
      // rows whose bitmap the symbolic phase kept are written from it; the others walk their products
      if (!h->d_emit_perm) {
        DevBuf c2;
        KK_HIP(c2.alloc(2 * sizeof(unsigned long long)));
        KK_HIP(hipMemsetAsync(c2.p, 0, 2 * sizeof(unsigned long long), st));
        KK_HIP(hipMalloc((void**)&h->d_emit_perm, sizeof(int32_t) * (size_t)nb(4)));
        int32_t* d_ep = h->d_emit_perm; const int32_t* d_rs = h->d_row_slot; unsigned long long* d_c2 = c2.as<unsigned long long>();
        KK_LAUNCH((spgemm_split_stored_kernel<int32_t>), (unsigned)ceil_div(nb(4), kBlock), kBlock, 0, st, nb(4), dperm, d_rs, d_ep, d_c2);
        unsigned long long h_c2[2] = {0, 0};
        KK_HIP(hipMemcpyAsync(h_c2, c2.p, sizeof h_c2, hipMemcpyDeviceToHost, st));
        KK_HIP(hipStreamSynchronize(st));
        h->n_emit_stored = (int64_t)h_c2[0];
      }
      const int64_t ns = h->n_emit_stored, nr = nb(4) - ns;
      if (ns) {
        const int32_t* d_ep = h->d_emit_perm; const int32_t* d_rs = h->d_row_slot; const kk_u64* d_st = (const kk_u64*)h->d_bm_store;
        KK_LAUNCH((spgemm_emit_bitmap_kernel<OffT>), (unsigned)ns, kDenseBlock, 0, st, d_ep, d_rs, d_st, h->bm_words, rmC, entC, g_spgemm.emit_staged);
      }
      int64_t np = 0;                                            // of the others: rows whose entry list the symbolic phase left in the pool
      const int32_t* rest = h->d_emit_perm + ns;
      if (nr && h->d_pool_off && h->pool_used > 0) {
        if (!h->d_emit_perm2) {
          DevBuf c3;
          KK_HIP(c3.alloc(2 * sizeof(unsigned long long)));
          KK_HIP(hipMemsetAsync(c3.p, 0, 2 * sizeof(unsigned long long), st));
          KK_HIP(hipMalloc((void**)&h->d_emit_perm2, sizeof(int32_t) * (size_t)nr));
          int32_t* d_ep2 = h->d_emit_perm2; const long long* d_po = h->d_pool_off; unsigned long long* d_c3 = c3.as<unsigned long long>();
          KK_LAUNCH((spgemm_split_stored_kernel<long long>), (unsigned)ceil_div(nr, kBlock), kBlock, 0, st, nr, rest, d_po, d_ep2, d_c3);
          unsigned long long h_c3[2] = {0, 0};
          KK_HIP(hipMemcpyAsync(h_c3, c3.p, sizeof h_c3, hipMemcpyDeviceToHost, st));
          KK_HIP(hipStreamSynchronize(st));
          h->n_emit_pooled = (int64_t)h_c3[0];
        }
        np = h->n_emit_pooled; rest = h->d_emit_perm2;
        if (np) {
          const long long* d_po = h->d_pool_off; const int32_t* d_pl = h->d_ent_pool;
          KK_LAUNCH((spgemm_copy_pool_kernel<OffT>), (unsigned)np, kBlock, 0, st, rest, d_po, d_pl, rmC, entC);
        }
      }
      if (nr - np && (rc = emit_walk(nr - np, rest + np, (int64_t)g_spgemm.emit_win_bits))) return rc;
      h->bitmaps_used = ns; h->pooled_used = np;
    }
    else if ((rc = emit_walk(nb(4), dperm, (int64_t)0))) return rc;
#define KK_VALS2(HH, NTT, GG, LAA, GRID, PERM, CAP)                                                                                     \
  do {                                                                                                                                  \
    KK_LAUNCH((spgemm_dense_vals2_kernel<OffT, VT, HH, NTT, GG, LAA, 0>), (unsigned)(GRID), NTT, 0, st, PERM, rmA, entA, valA, rmB, entB, valB, \
              rmC, (const int32_t*)entC, valC, (CAP) | (g_spgemm.nt ? (1 << 30) : 0), h->nnzB KK_DBG_ARG);                             \
  } while (0)
    const int64_t n_blk = h->n_dense_block;
    const int64_t n_lds = h->n_dense_lds; int64_t n_hubl = h->n_dense_hub_lds, n_hub = nb(4) - n_blk - n_lds - n_hubl;
    const bool flat_vals = g_spgemm.val_kernel == 2 && h->dense_lds;
    if (n_blk) {
      // column-block value kernel: the index of B (kept while B's arrays and the block grid are the same), the index of these rows'
      // entries(C) (once per symbolic phase: it depends on the structure only), then one workgroup per (row, block), block-major
      int wshift = 6;
      while ((1 << wshift) < g_spgemm.block_w) ++wshift;
      const int nblk = (int)ceil_div(k, (int64_t)1 << wshift);
      const int64_t nB = h->n;
      const int32_t* bperm = dperm + (nb(4) - n_blk);
      if (!h->d_bidx || h->bidx_rmB != rmB_ || h->bidx_entB != (const void*)entB || h->bidx_nB != nB || h->bidx_nblk != nblk || h->bidx_wshift != wshift) {
        if (h->d_bidx) { (void)hipFree(h->d_bidx); h->d_bidx = nullptr; }
        KK_HIP(hipMalloc((void**)&h->d_bidx, sizeof(unsigned) * (size_t)(nblk + 1) * (size_t)nB));
        unsigned* d_bx = h->d_bidx;
        KK_LAUNCH((spgemm_bidx_kernel<OffT>), (unsigned)ceil_div((int64_t)(nblk + 1) * nB, kBlock), kBlock, 0, st, nB, nblk, wshift, rmB, entB, d_bx);
        h->bidx_rmB = rmB_; h->bidx_entB = entB; h->bidx_nB = nB; h->bidx_nblk = nblk; h->bidx_wshift = wshift;
      }
      if (h->idx_nblk != nblk || h->idx_wshift != wshift) { h->cidx_ready = false; h->items_ready = false; }
      if (h->items_cap != g_spgemm.item_cap || h->items_blocks != g_spgemm.item_blocks) h->items_ready = false;
      if (!h->cidx_ready) {
        h->idx_nblk = nblk; h->idx_wshift = wshift;
        if (h->d_cidx) { (void)hipFree(h->d_cidx); h->d_cidx = nullptr; }
        KK_HIP(hipMalloc((void**)&h->d_cidx, sizeof(unsigned) * (size_t)(nblk + 1) * (size_t)n_blk));
        unsigned* d_cx = h->d_cidx;
        KK_LAUNCH((spgemm_cidx_kernel<OffT>), (unsigned)ceil_div((int64_t)(nblk + 1) * n_blk, kBlock), kBlock, 0, st, n_blk, bperm, nblk, wshift, rmC, (const int32_t*)entC, d_cx);
        h->cidx_ready = true;
      }
      const size_t smem = sizeof(VT) << wshift;
      const unsigned* d_bx = h->d_bidx; const unsigned* d_cx = h->d_cidx;
      const bool use_items = g_spgemm.items && nblk <= 4096 && wshift >= 5;
      if (use_items && !h->items_ready) {
        // the class's items, once per symbolic phase (they depend on the structure of C only): count, scan, fill, order by first block
        if (h->d_items_rank) { (void)hipFree(h->d_items_rank); h->d_items_rank = nullptr; }
        if (h->d_items_direct) { (void)hipFree(h->d_items_direct); h->d_items_direct = nullptr; }
        h->n_items_rank = 0; h->n_items_direct = 0;
        DevBuf nr_b, nd_b, hist_b, tmp_r, tmp_d;
        KK_HIP(nr_b.alloc(sizeof(unsigned) * (size_t)(n_blk + 1))); KK_HIP(nd_b.alloc(sizeof(unsigned) * (size_t)(n_blk + 1)));
        unsigned* d_nr = nr_b.as<unsigned>(); unsigned* d_nd = nd_b.as<unsigned>();
        KK_HIP(hipMemsetAsync(d_nr, 0, sizeof(unsigned) * (size_t)(n_blk + 1), st)); KK_HIP(hipMemsetAsync(d_nd, 0, sizeof(unsigned) * (size_t)(n_blk + 1), st));
        const unsigned cap_items = (unsigned)g_spgemm.item_cap;
        const unsigned bgrid = (unsigned)ceil_div(n_blk, kBlock);
        KK_LAUNCH(spgemm_items_build_kernel, bgrid, kBlock, 0, st, n_blk, nblk, d_cx, cap_items, g_spgemm.item_blocks, 0, d_nr, d_nd, (int2*)nullptr, (int2*)nullptr);
        if ((rc = exclusive_scan_inplace<unsigned>(d_nr, n_blk + 1, st))) return rc;
        if ((rc = exclusive_scan_inplace<unsigned>(d_nd, n_blk + 1, st))) return rc;
        unsigned h_tot[2] = {0, 0};
        KK_HIP(hipMemcpyAsync(&h_tot[0], d_nr + n_blk, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        KK_HIP(hipMemcpyAsync(&h_tot[1], d_nd + n_blk, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        KK_HIP(hipStreamSynchronize(st));
        KK_HIP(tmp_r.alloc(sizeof(int2) * (size_t)(h_tot[0] ? h_tot[0] : 1))); KK_HIP(tmp_d.alloc(sizeof(int2) * (size_t)(h_tot[1] ? h_tot[1] : 1)));
        DevBuf ord_r, ord_d;                       // the items ordered by first block; what the value kernels read are the heads made of them
        KK_HIP(ord_r.alloc(sizeof(int2) * (size_t)(h_tot[0] ? h_tot[0] : 1))); KK_HIP(ord_d.alloc(sizeof(int2) * (size_t)(h_tot[1] ? h_tot[1] : 1)));
        KK_HIP(hipMalloc((void**)&h->d_items_rank, sizeof(ItemHead) * (size_t)(h_tot[0] ? h_tot[0] : 1)));
        KK_HIP(hipMalloc((void**)&h->d_items_direct, sizeof(ItemHead) * (size_t)(h_tot[1] ? h_tot[1] : 1)));
        int2* t_r = tmp_r.as<int2>(); int2* t_d = tmp_d.as<int2>();
        KK_LAUNCH(spgemm_items_build_kernel, bgrid, kBlock, 0, st, n_blk, nblk, d_cx, cap_items, g_spgemm.item_blocks, 1, d_nr, d_nd, t_r, t_d);
        KK_HIP(hist_b.alloc(sizeof(unsigned) * (size_t)(nblk + 1)));
        unsigned* d_hist = hist_b.as<unsigned>();
        for (int which = 0; which < 2; ++which) {
          const int64_t n_it = h_tot[which];
          if (!n_it) continue;
          const int2* src = which == 0 ? t_r : t_d; int2* dst = which == 0 ? ord_r.as<int2>() : ord_d.as<int2>();
          KK_HIP(hipMemsetAsync(d_hist, 0, sizeof(unsigned) * (size_t)(nblk + 1), st));
          const int64_t nbk_it = ceil_div(n_it, kBlock);
          const unsigned g_it = (unsigned)(nbk_it < 2048 ? nbk_it : 2048);
          KK_LAUNCH(spgemm_items_hist_kernel, g_it, kBlock, 0, st, n_it, src, nblk, d_hist);
          KK_LAUNCH(spgemm_items_scan_kernel, 1, 64, 0, st, nblk, d_hist);
          KK_LAUNCH(spgemm_items_scatter_kernel, g_it, kBlock, 0, st, n_it, src, nblk, d_hist, dst);
          KK_LAUNCH((spgemm_items_head_kernel<OffT>), (unsigned)nbk_it, kBlock, 0, st, n_it, (const int2*)dst, bperm, nblk, d_cx, rmA, rmC,
                    which == 0 ? h->d_items_rank : h->d_items_direct);
        }
        KK_HIP(hipStreamSynchronize(st));              // the scratch buffers go out of scope
        h->n_items_rank = h_tot[0]; h->n_items_direct = h_tot[1]; h->items_ready = true; h->items_cap = g_spgemm.item_cap; h->items_blocks = g_spgemm.item_blocks;
        if (h->verbose) KK_VERBOSE("\tkkamd spgemm numeric: column-block class: %lld rows as %lld position-indexed items (<= %u entries, <= %d blocks) and %lld column-indexed blocks\n",
                                   (long long)n_blk, (long long)h->n_items_rank, cap_items, g_spgemm.item_blocks, (long long)h->n_items_direct);
      }
      if (use_items) {
        const ItemHead* d_ir = h->d_items_rank; const ItemHead* d_id = h->d_items_direct;
        if (h->n_items_direct) {
#ifndef KK_EMU
          KK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spgemm_item_vals_kernel<OffT, VT, kDenseBlock, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
#endif
          KK_LAUNCH((spgemm_item_vals_kernel<OffT, VT, kDenseBlock, false>), (unsigned)h->n_items_direct, kDenseBlock, smem, st, d_id, wshift, nB, d_bx,
                    entA, valA, rmB, entB, valB, h->nnzB, (const int32_t*)entC, valC, 1);
        }
        if (h->n_items_rank) {
          const size_t smem_r = (((size_t)g_spgemm.item_blocks << (wshift - 5)) * 8) + sizeof(VT) * (size_t)g_spgemm.item_cap;
#ifndef KK_EMU
          KK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spgemm_item_vals_kernel<OffT, VT, kValBlock, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_r));
#endif
          KK_LAUNCH((spgemm_item_vals_kernel<OffT, VT, kValBlock, true>), (unsigned)h->n_items_rank, kValBlock, smem_r, st, d_ir, wshift, nB, d_bx,
                    entA, valA, rmB, entB, valB, h->nnzB, (const int32_t*)entC, valC, g_spgemm.item_blocks);
        }
      } else {
      // blocks of 16384 columns: one workgroup of 1024 per CU around 128 KB of sums; narrower blocks: 512 work-items, two (or more) workgroups per CU
#ifndef KK_EMU
#define KK_BLK_ATTR(NTT) KK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spgemm_block_vals_kernel<OffT, VT, NTT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem))
#else
#define KK_BLK_ATTR(NTT) (void)0
#endif
#define KK_BLK(NTT)                                                                                                                          \
      do {                                                                                                                                   \
        KK_BLK_ATTR(NTT);                                                                                                                    \
        KK_LAUNCH((spgemm_block_vals_kernel<OffT, VT, NTT>), dim3((unsigned)n_blk, (unsigned)nblk), NTT, smem, st, n_blk, bperm, nblk, wshift, nB, d_bx, d_cx, \
                  rmA, entA, valA, rmB, entB, valB, h->nnzB, rmC, (const int32_t*)entC, valC);                                               \
      } while (0)
      if (wshift >= 14) KK_BLK(kDenseBlock); else KK_BLK(kValBlock);
#undef KK_BLK
#undef KK_BLK_ATTR
      }
    }
    if (flat_vals && g_spgemm.val_hub_flat) {
      // measured and not kept as the default: A rows above kValLa through the flat kernel, kValLa lists per pass (R-MAT scale 20:
      // numeric 353 -> 479 ms -- with thousands of lists a group of windows costs 8 x 512 searches per pass and a pass finds a
      // handful of products per list and window; the cached-next-column sweep of spgemm_hub_vals_kernel skips the empty lists for free)
      int cap = g_spgemm.val_cap;
      cap = cap < 64 ? 64 : (cap > kValTable / 2 ? kValTable / 2 : cap);
      if (n_hubl + n_hub) KK_VALS2(kValTable, kValBlock, 8, kValLa, n_hubl + n_hub, dperm + n_lds, cap);
      n_hubl = 0; n_hub = 0;
    }
    if (n_hub && flat_vals && h->hub_from_mid) {      // A rows above kValLa2 entries (the heaviest rows first): the cached-cursor hub kernel, kHubLa entries per pass
      int cap = g_spgemm.val_cap;
      cap = cap < 64 ? 64 : (cap > kValTable / 2 ? kValTable / 2 : cap);
      const int32_t* hperm = dperm + n_lds + n_hubl;
      // the passes of a row that has several add into values(C) with global_atomic_add_f64: only into ordinary device memory (hardware
      // floating-point atomics are not dependable on fine-grained / managed allocations); anything else keeps one workgroup per row, which
      // runs the passes one after the other without atomics
      bool plain_valc = true;
#ifndef KK_EMU
      {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, (const void*)valC) != hipSuccess) { (void)hipGetLastError(); plain_valc = false; }
        else plain_valc = attr.type == hipMemoryTypeDevice && !attr.isManaged;
      }
#endif
      const bool hub_split = g_spgemm.hub_split && plain_valc;
      if (hub_split && !h->d_hub_items) {            // (row, pass) items, once per set of bins
        DevBuf pb;
        KK_HIP(pb.alloc(sizeof(int32_t) * (size_t)n_hub));
        int32_t* d_p = pb.as<int32_t>();
        KK_LAUNCH((spgemm_hub_passes_kernel<OffT>), (unsigned)ceil_div(n_hub, kBlock), kBlock, 0, st, n_hub, hperm, rmA, d_p);
        std::vector<int32_t> h_p((size_t)n_hub), h_items, h_multi;
        KK_HIP(hipMemcpyAsync(h_p.data(), d_p, sizeof(int32_t) * (size_t)n_hub, hipMemcpyDeviceToHost, st));
        KK_HIP(hipStreamSynchronize(st));
        // the rows with the most passes first: their passes are the longest-running items
        std::vector<int32_t> ord((size_t)n_hub);
        for (int64_t i = 0; i < n_hub; ++i) ord[(size_t)i] = (int32_t)i;
        std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return h_p[(size_t)x] > h_p[(size_t)y]; });
        for (int32_t i : ord) {
          for (int32_t ps = 0; ps < h_p[(size_t)i]; ++ps) { h_items.push_back(i); h_items.push_back(ps); }
          if (h_p[(size_t)i] > 1) h_multi.push_back(i);
        }
        h->n_hub_items = (int64_t)h_items.size() / 2; h->n_hub_multi = (int64_t)h_multi.size();
        KK_HIP(hipMalloc((void**)&h->d_hub_items, sizeof(int32_t) * (h_items.size() ? h_items.size() : 1)));
        KK_HIP(hipMalloc((void**)&h->d_hub_multi, sizeof(int32_t) * (h_multi.size() ? h_multi.size() : 1)));
        if (!h_items.empty()) KK_HIP(hipMemcpyAsync(h->d_hub_items, h_items.data(), sizeof(int32_t) * h_items.size(), hipMemcpyHostToDevice, st));
        if (!h_multi.empty()) KK_HIP(hipMemcpyAsync(h->d_hub_multi, h_multi.data(), sizeof(int32_t) * h_multi.size(), hipMemcpyHostToDevice, st));
        KK_HIP(hipStreamSynchronize(st));
      }
      if (hub_split && h->d_hub_items && h->n_hub_items > 0) {
        const int32_t* d_items = h->d_hub_items; const int32_t* d_multi = h->d_hub_multi;
        if (h->n_hub_multi) KK_LAUNCH((spgemm_zero_rows_kernel<OffT, VT>), (unsigned)h->n_hub_multi, kBlock, 0, st, hperm, d_multi, rmC, valC);
        KK_LAUNCH((spgemm_hub_vals_kernel<OffT, VT>), (unsigned)h->n_hub_items, kDenseBlock, 0, st, hperm, rmA, entA, valA, rmB, entB, valB,
                  rmC, (const int32_t*)entC, valC, cap | (g_spgemm.nt ? (1 << 30) : 0), d_items);
      } else {
        KK_LAUNCH((spgemm_hub_vals_kernel<OffT, VT>), (unsigned)n_hub, kDenseBlock, 0, st, hperm, rmA, entA, valA, rmB, entB, valB,
                  rmC, (const int32_t*)entC, valC, cap | (g_spgemm.nt ? (1 << 30) : 0), (const int32_t*)nullptr);
      }
      n_hub = 0;
    }
    if (n_hubl && flat_vals && h->hub_from_mid) {    // A rows of kValLa + 1 .. kValLa2 entries: the flat kernel with 1024 lists per pass
      int cap = g_spgemm.val_cap;
      cap = cap < 64 ? 64 : (cap > kValTable / 2 ? kValTable / 2 : cap);
      KK_VALS2(kValTable, kDenseBlock, 4, kValLa2, n_hubl, dperm + n_lds, cap);
      n_hubl = 0;
    }
    if (n_hubl) {      // heaviest rows first
      int cap = g_spgemm.val_cap;
      cap = cap < 64 ? 64 : (cap > kValTable / 2 ? kValTable / 2 : cap);
      KK_LAUNCH((spgemm_hub_vals_kernel<OffT, VT>), (unsigned)n_hubl, kDenseBlock, 0, st, dperm + n_lds, rmA, entA, valA, rmB, entB, valB,
                rmC, (const int32_t*)entC, valC, cap | (g_spgemm.nt ? (1 << 30) : 0), (const int32_t*)nullptr);
    }
    if (n_lds) {
      int cap = g_spgemm.val_cap;
      if (cap < 64) cap = 64;
#define KK_VALS(HH, NTT)                                                                                              \
  do {                                                                                                                \
    if (cap > HH / 2) cap = HH / 2;                                                                                   \
    KK_LAUNCH((spgemm_dense_vals_kernel<OffT, VT, HH, NTT>), (unsigned)n_lds, NTT, 0, st, dperm, rmA, entA, valA, rmB, \
              entB, valB, rmC, (const int32_t*)entC, valC, cap KK_DBG_ARG);                                      \
  } while (0)
      if (flat_vals) {
        if (cap > kValTable / 2) cap = kValTable / 2;
        const int64_t n_tiny = h->n_dense_tiny < n_lds ? h->n_dense_tiny : n_lds;
        const int64_t n_small = h->n_dense_small < n_lds - n_tiny ? h->n_dense_small : n_lds - n_tiny;
        if (n_tiny) {                                            // the lightest shape (see spgemm_split_small_kernel)
          const int cap_t = cap > kValTableTiny / 2 ? kValTableTiny / 2 : cap;
          KK_VALS2(kValTableTiny, 128, 8, kValLaTiny, n_tiny, dperm, cap_t);
        }
        if (n_small) {                                           // the light shape for the rows with few entries
          const int cap_s = cap > kValTableSmall / 2 ? kValTableSmall / 2 : cap;
          KK_VALS2(kValTableSmall, kBlock, 8, kValLaSmall, n_small, dperm + n_tiny, cap_s);
        }
        if (n_lds - n_tiny - n_small)
          KK_VALS2(kValTable, kValBlock, 8, kValLa, n_lds - n_tiny - n_small, dperm + n_tiny + n_small, cap);
      } else switch (g_spgemm.val_shape) {
        case 1: KK_VALS(8192, 1024); break;
        case 2: KK_VALS(8192, 512); break;
        case 3: KK_VALS(2048, 256); break;
        case 4: KK_VALS(4096, 1024); break;
        default: KK_VALS(kValTable, kValBlock); break;
      }
#undef KK_VALS
#undef KK_VALS2
    }
    if (n_hub) {
      // batches of G rows, each with its own k-wide accumulator (bounded to 1/8 of free HBM), ~256K work-items in flight
      size_t free_b = 0, total_b = 0;
      KK_HIP(hipMemGetInfo(&free_b, &total_b));
      int64_t G = (int64_t)(free_b / 8) / (k * (int64_t)sizeof(VT));
      if (G < 1) return fail(KKAMD_ERR_ALLOC, "spgemm: not enough device memory for one hub-row accumulator (%lld bytes)", (long long)(k * (int64_t)sizeof(VT)));
      if (G > 1024) G = 1024;
      if (G > n_hub) G = n_hub;
      int64_t bx = 1024 / G;
      bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
      KK_HIP(acc_b.alloc((size_t)k * sizeof(VT) * (size_t)G));
      d_acc = acc_b.as<VT>();
      KK_HIP(hipMemsetAsync(d_acc, 0, (size_t)k * sizeof(VT) * (size_t)G, st));
      for (int64_t r0 = 0; r0 < n_hub; r0 += G) {
        const unsigned g = (unsigned)(n_hub - r0 < G ? n_hub - r0 : G);
        const int32_t* rows = dperm + n_lds + n_hubl + r0;
        KK_LAUNCH((spgemm_hub_acc_kernel<OffT, VT>), dim3((unsigned)bx, g), kBlock, 0, st, rows, rmA, entA, valA, rmB, entB, valB, d_acc, k, sg);
        KK_LAUNCH((spgemm_hub_extract_kernel<OffT, VT>), dim3((unsigned)bx, g), kBlock, 0, st, rows, rmC, (const int32_t*)entC, valC, d_acc, k);
      }
    }
  }
  hipError_t e = hipGetLastError();
  hipError_t e2 = hipStreamSynchronize(st);   // the reference's numeric phase fences too (impl_kkmem.hpp:1440,1467)
  if (e != hipSuccess || e2 != hipSuccess) { h->entries_valid = false; return fail(KKAMD_ERR_HIP, "spgemm numeric failed: %s", hipGetErrorString(e != hipSuccess ? e : e2)); }
  h->entries_valid = true; h->entC_ptr = entC; h->rmC_ptr = rmC_;
  if (h->d_bm_store || h->unit_mode) { const int64_t used = h->bitmaps_used; free_bitmap_store(h); h->bitmaps_used = used; }    // entries(C) are written: the bitmaps (GBs) are not needed again
  return KKAMD_OK;
}

template <class OffT> __global__ void max_diff_kernel(int64_t m, const OffT* __restrict__ rm, unsigned long long* out) {
  unsigned long long mx = 0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long l = (unsigned long long)((int64_t)rm[r + 1] - (int64_t)rm[r]);
    mx = l > mx ? l : mx;
  }
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(mx, o, 64); mx = other > mx ? other : mx; }
  if ((threadIdx.x & 63) == 0 && mx) atomicMax(out, mx);
}

int spgemm_set_default(const char* key, int value) {
  const std::string k(key ? key : "");
  if (k == "spgemm_win_bits") {
    if (value < 64 || value > (1 << 20) || (value & 63)) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_win_bits must be a multiple of 64 in [64, 2^20]");
    g_spgemm.win_bits = value;
  } else if (k == "spgemm_val_cap") {
    if (value < 64 || value > 4096) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_val_cap must be in [64, 4096]");
    g_spgemm.val_cap = value;
  } else if (k == "spgemm_force_unsorted") g_spgemm.force_unsorted = value != 0;
  else if (k == "spgemm_emit_chunked") g_spgemm.emit_chunked = value != 0;
#ifdef KK_ABLATE
  else if (k == "spgemm_debug") g_spgemm.debug = value;
#endif
  else if (k == "spgemm_val_shape") g_spgemm.val_shape = value;
  else if (k == "spgemm_val_hub_flat") g_spgemm.val_hub_flat = value != 0;
  else if (k == "spgemm_col_quads") { if (value != 0 && value != 1 && value != 4) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_col_quads: %d is not 0 or 4", value); g_spgemm.col_quads = value == 1 ? 4 : value; }
  else if (k == "spgemm_hub_chunked") g_spgemm.hub_chunked = value != 0;
  else if (k == "spgemm_val_mid") g_spgemm.val_mid = value != 0;
  else if (k == "spgemm_keep_bitmaps") g_spgemm.keep_bitmaps = value != 0;
  else if (k == "spgemm_keep_lists") g_spgemm.keep_lists = value != 0;
  else if (k == "spgemm_val_tiny_cnt") { if (value < 0) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_val_tiny_cnt: %d is negative", value); g_spgemm.val_tiny_cnt = value; }
  else if (k == "spgemm_emit_sort") g_spgemm.emit_sort = value != 0;
  else if (k == "spgemm_pool_keep") { if (value < 0 || value > 2) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_pool_keep: 0 (first product returns the store, repeat users keep it), 1 (keep) or 2 (return)"); g_spgemm.pool_keep = value; }
  else if (k == "spgemm_sort_rows") g_spgemm.sort_rows = value != 0;
  else if (k == "spgemm_store_cap_mb") { if (value < 0) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_store_cap_mb: %d is negative", value); g_spgemm.store_cap_mb = value; }
  else if (k == "spgemm_sym_units") g_spgemm.sym_units = value != 0;
  else if (k == "spgemm_unit_bits") { if (value < 6 || value > kUnitBitsMax) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_unit_bits: 6 .. %d", kUnitBitsMax); g_spgemm.unit_bits = value; }
  else if (k == "spgemm_nt") g_spgemm.nt = value != 0;
  else if (k == "spgemm_list_staged") g_spgemm.list_staged = value != 0;
  else if (k == "spgemm_block") g_spgemm.block = value != 0;
  else if (k == "spgemm_items") g_spgemm.items = value != 0;
  else if (k == "spgemm_item_blocks") { if (value < 1 || value > 16) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_item_blocks: 1 .. 16"); g_spgemm.item_blocks = value; }
  else if (k == "spgemm_item_cap") { if (value < 64 || value > 16384) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_item_cap: 64 .. 16384"); g_spgemm.item_cap = value; }
  else if (k == "spgemm_block_w") { if (value < 64 || value > 16384 || (value & (value - 1))) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_block_w: %d is not a power of two in [64, 16384]", value); g_spgemm.block_w = value; }
  else if (k == "spgemm_block_min_pct") { if (value < 0 || value > 100) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_block_min_pct: 0 .. 100"); g_spgemm.block_min_pct = value; }
  else if (k == "spgemm_block_la_pct") { if (value < 0 || value > 100) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_block_la_pct: 0 .. 100"); g_spgemm.block_la_pct = value; }
  else if (k == "spgemm_quad_rows") { if (value < 0 || value > 2) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_quad_rows: %d is not 0, 1 or 2", value); g_spgemm.quad_rows = value; }
  else if (k == "spgemm_val_small_cnt") { if (value < 0) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_val_small_cnt: %d is negative", value); g_spgemm.val_small_cnt = value; }
  else if (k == "spgemm_emit_staged") g_spgemm.emit_staged = value != 0;
  else if (k == "spgemm_sym_large") g_spgemm.sym_large = value != 0;
  else if (k == "spgemm_hub_split") g_spgemm.hub_split = value != 0;
  else if (k == "spgemm_emit_win_bits") { if (value < 0 || (value & 63)) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_emit_win_bits must be a multiple of 64"); g_spgemm.emit_win_bits = value; }
  else if (k == "spgemm_val_la2") { if (value < kValLa2) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_val_la2 must be at least %d", kValLa2); g_spgemm.val_la2 = value; }
  else if (k == "spgemm_val_kernel") { if (value != 1 && value != 2) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_val_kernel is 1 or 2"); g_spgemm.val_kernel = value; }
  else if (k == "spgemm_val_la") g_spgemm.val_la = value;
  else return fail(KKAMD_ERR_INVALID_ARG, "kkamd_set_default: unknown key '%s'", k.c_str());
  return KKAMD_OK;
}

}  // namespace kk

extern "C" {

int kkamd_spgemm_create(kkamd_spgemm_handle_t** handle) {
  if (!handle) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_create: null output pointer");
  *handle = new (std::nothrow) kkamd_spgemm_handle();
  if (*handle) { kk::BmPool& pool = kk::bm_pool(); std::lock_guard<std::mutex> g(pool.m); ++pool.live_handles; }
  return *handle ? KKAMD_OK : kk::fail(KKAMD_ERR_ALLOC, "kkamd_spgemm_create: out of host memory");
}

int kkamd_spgemm_destroy(kkamd_spgemm_handle_t* h) {
  if (!h) return KKAMD_OK;
  if (h->d_sizes) (void)hipFree(h->d_sizes);
  if (h->d_perm) (void)hipFree(h->d_perm);
  if (h->d_flop_cls) (void)hipFree(h->d_flop_cls);
  kk::free_bitmap_store(h);
  if (h->d_hub_items) (void)hipFree(h->d_hub_items);
  if (h->d_hub_multi) (void)hipFree(h->d_hub_multi);
  if (h->d_bidx) (void)hipFree(h->d_bidx);
  if (h->aux) { (void)hipStreamSynchronize(h->aux); (void)hipStreamDestroy(h->aux); }
  kk::park_small(h);
  if (h->d_scan_ws) (void)hipFree(h->d_scan_ws);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->d_cidx) (void)hipFree(h->d_cidx);
  if (h->d_items_rank) (void)hipFree(h->d_items_rank);
  if (h->d_items_direct) (void)hipFree(h->d_items_direct);
  delete h;
  // the last handle gone: the pooled store (GBs) goes back to the device unless the host asked to keep it ("spgemm_pool_keep")
  bool release = false;
  {
    kk::BmPool& pool = kk::bm_pool(); std::lock_guard<std::mutex> g(pool.m);
    const bool last = --pool.live_handles <= 0;
    if (pool.live_handles < 0) pool.live_handles = 0;
    release = last && (pool.p || kk::tmp_pool().p) && (kk::g_spgemm.pool_keep == 2 || (kk::g_spgemm.pool_keep == 0 && !pool.sticky));
    if (release) pool.released_once = true;
  }
  if (release) (void)kk::release_bitmap_pool();
  return KKAMD_OK;
}

int kkamd_spgemm_symbolic(kkamd_spgemm_handle_t* h, int64_t m, int64_t n, int64_t k, const void* d_row_mapA,
                          const int32_t* d_entriesA, const void* d_row_mapB, const int32_t* d_entriesB,
                          void* d_row_mapC, int offset_type, int64_t* c_nnz, kkamd_stream_t stream) {
  if (!h) return kk::fail(KKAMD_ERR_STATE, "KokkosSparse::spgemm_symbolic: the given KernelHandle does not have an SpGEMM handle associated with it.");
  if (m < 0 || n < 0 || k < 0) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_symbolic: negative dimension");
  if (m > INT32_MAX || n > INT32_MAX || k > INT32_MAX) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_symbolic: dimension exceeds int32 ordinals");
  if (offset_type != KKAMD_I32 && offset_type != KKAMD_I64) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_symbolic: unknown offset_type %d", offset_type);
  if (!d_row_mapC) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_symbolic: null row_map C");
  hipStream_t st = kk::to_hip(stream);
  kk::TraceRange range("KokkosSparse::spgemm_symbolic[TPL_KKAMD]");
  const size_t osz = offset_type == KKAMD_I64 ? 8 : 4;
  // idempotent: a second symbolic on the same handle returns the stored answer
  // (sparse/impl/KokkosSparse_spgemm_symbolic_spec.hpp:99)
  if (h->symbolic_called && h->m == m && h->n == n && h->k == k && h->rmA == d_row_mapA && h->rmB == d_row_mapB) {
    if (c_nnz) *c_nnz = h->c_nnz;
    return KKAMD_OK;
  }
  h->m = m; h->n = n; h->k = k; h->offset_type = offset_type; h->rmA = d_row_mapA; h->rmB = d_row_mapB;
  h->symbolic_called = false; h->numeric_called = false; h->numeric_bins_ready = false; h->entries_valid = false;
  if (h->d_bidx) { (void)hipFree(h->d_bidx); h->d_bidx = nullptr; }     // a new symbolic phase may bring another B in the same arrays
  h->cidx_ready = false; h->items_ready = false; h->n_dense_block = 0;
  kk::free_bitmap_store(h); h->bitmaps_used = 0; h->last_units = 0; h->last_unit_bitmaps = 0; h->last_unit_rows_kept = 0;
  h->c_nnz = 0; h->mults = 0; h->max_row_flops = 0; h->max_row_nnz = 0;
  // empty product: zero row_map (:100-107; the rocSPARSE wrapper memsets too)
  int64_t nnzA = 0, nnzB = 0;
  if (m > 0 && n > 0) {
    if (!d_row_mapA || !d_row_mapB) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_symbolic: null row_map");
    unsigned char buf[16];
    KK_HIP(hipMemcpyAsync(buf, (const char*)d_row_mapA + osz * (size_t)m, osz, hipMemcpyDeviceToHost, st));
    KK_HIP(hipMemcpyAsync(buf + 8, (const char*)d_row_mapB + osz * (size_t)n, osz, hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
    nnzA = offset_type == KKAMD_I64 ? *(int64_t*)buf : (int64_t) * (int32_t*)buf;
    nnzB = offset_type == KKAMD_I64 ? *(int64_t*)(buf + 8) : (int64_t) * (int32_t*)(buf + 8);
  }
  if (m == 0 || n == 0 || k == 0 || nnzA == 0 || nnzB == 0) {
    KK_HIP(hipMemsetAsync(d_row_mapC, 0, osz * (size_t)(m + 1), st));
    KK_HIP(hipStreamSynchronize(st));
    h->symbolic_called = true;
    if (c_nnz) *c_nnz = 0;
    return KKAMD_OK;
  }
  if (!d_entriesA || !d_entriesB) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_symbolic: null entries");
  h->nnzA = nnzA;
  if (h->sizes_cap < m || !h->d_sizes || !h->d_perm) {
    if (h->d_sizes) { (void)hipFree(h->d_sizes); h->d_sizes = nullptr; }
    if (h->d_perm) { (void)hipFree(h->d_perm); h->d_perm = nullptr; }
    h->sizes_cap = 0;
    KK_HIP(hipMalloc((void**)&h->d_sizes, sizeof(int64_t) * (size_t)m));
    KK_HIP(hipMalloc((void**)&h->d_perm, sizeof(int32_t) * (size_t)m));
    h->sizes_cap = m;
  }
  int64_t total = 0;
  int rc = offset_type == KKAMD_I64
               ? kk::symbolic_typed<int64_t>(h, m, n, k, d_row_mapA, d_entriesA, d_row_mapB, d_entriesB, d_row_mapC, nnzB, &total, st)
               : kk::symbolic_typed<int32_t>(h, m, n, k, d_row_mapA, d_entriesA, d_row_mapB, d_entriesB, d_row_mapC, nnzB, &total, st);
  if (rc) return rc;
  if (offset_type == KKAMD_I32 && total < 0)
    return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_symbolic: nnz(C) overflows 32-bit offsets; use 64-bit offsets");
  h->c_nnz = total; h->symbolic_called = true;
  if (c_nnz) *c_nnz = total;
  return KKAMD_OK;
}

int kkamd_spgemm_numeric(kkamd_spgemm_handle_t* h, int64_t m, int64_t n, int64_t k, const void* d_row_mapA,
                         const int32_t* d_entriesA, const void* d_valuesA, const void* d_row_mapB,
                         const int32_t* d_entriesB, const void* d_valuesB, const void* d_row_mapC,
                         int32_t* d_entriesC, void* d_valuesC, int offset_type, int value_type,
                         kkamd_stream_t stream) {
  if (!h) return kk::fail(KKAMD_ERR_STATE, "KokkosSparse::spgemm_numeric: the given KernelHandle does not have an SpGEMM handle associated with it.");
  if (!h->symbolic_called)
    return kk::fail(KKAMD_ERR_STATE, "KokkosSparse::spgemm_numeric: must first call spgemm_symbolic with the same handle.");
  if (h->m != m || h->n != n || h->k != k || h->offset_type != offset_type)
    return kk::fail(KKAMD_ERR_STATE, "KokkosSparse::spgemm_numeric: dimensions/offset type differ from the symbolic call on this handle.");
  if (value_type != KKAMD_F32 && value_type != KKAMD_F64) return kk::fail(KKAMD_ERR_UNSUPPORTED, "kkamd_spgemm_numeric: unsupported value_type %d", value_type);
  h->numeric_called = true;
  if (h->c_nnz == 0) return KKAMD_OK;
  if (!d_valuesA || !d_valuesB || !d_entriesC || !d_valuesC || !d_row_mapC || !d_entriesA || !d_entriesB)
    return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_numeric: null pointer");
  hipStream_t st = kk::to_hip(stream);
  kk::TraceRange range(value_type == KKAMD_F64 ? "KokkosSparse::spgemm_numeric[TPL_KKAMD,double]" : "KokkosSparse::spgemm_numeric[TPL_KKAMD,float]");
  (void)n;
  if (offset_type == KKAMD_I64) {
    return value_type == KKAMD_F64
               ? kk::numeric_typed<int64_t, double>(h, m, k, d_row_mapA, d_entriesA, d_valuesA, d_row_mapB, d_entriesB, d_valuesB, d_row_mapC, d_entriesC, d_valuesC, st)
               : kk::numeric_typed<int64_t, float>(h, m, k, d_row_mapA, d_entriesA, d_valuesA, d_row_mapB, d_entriesB, d_valuesB, d_row_mapC, d_entriesC, d_valuesC, st);
  }
  return value_type == KKAMD_F64
             ? kk::numeric_typed<int32_t, double>(h, m, k, d_row_mapA, d_entriesA, d_valuesA, d_row_mapB, d_entriesB, d_valuesB, d_row_mapC, d_entriesC, d_valuesC, st)
             : kk::numeric_typed<int32_t, float>(h, m, k, d_row_mapA, d_entriesA, d_valuesA, d_row_mapB, d_entriesB, d_valuesB, d_row_mapC, d_entriesC, d_valuesC, st);
}

int kkamd_spgemm_set(kkamd_spgemm_handle_t* h, const char* key, double value) {
  if (!h || !key) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_set: null argument");
  const std::string k(key);
  if (k == "algorithm") {
    // SPGEMMAlgorithm (sparse/src/KokkosSparse_spgemm_handle.hpp:44-93): 0 KK, 1 KK_DENSE, 2 KK_MEMORY, 3 KK_LP, 4 DEFAULT, 5 DEBUG,
    // 6 SERIAL, 7 KK_SPEED, 8 KK_MEMSPEED
    const int a = (int)value;
    if (a < 0 || a > 8) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_set: unknown SPGEMMAlgorithm %d", a);
    // KK_MEMORY / KK_SPEED / KK_MEMSPEED / KK_LP are variants of the hash algorithm.  SPGEMM_DEBUG / SPGEMM_SERIAL are the
    // reference's host-sequential algorithms (spgemm_debug_symbolic / _numeric copy the device arrays to the host); the public
    // spgemm_numeric sorts every algorithm's rows afterwards (sparse/impl/KokkosSparse_spgemm_numeric_spec.hpp:138-140), so
    // their C is the C of every other algorithm: they run the device hash algorithm here, and say so under verbose.
    const int alg = (a == 1) ? 1 : 0;
    if (alg != h->algorithm) { h->numeric_bins_ready = false; h->entries_valid = false; }
    h->algorithm = alg; h->requested_algorithm = a;
    if (h->verbose && (a == 5 || a == 6)) printf("kkamd spgemm: %s requested: host-sequential in the reference, runs the LDS hash algorithm on the device here\n", a == 5 ? "SPGEMM_DEBUG" : "SPGEMM_SERIAL");
  } else if (k == "accumulator") {                               // SPGEMMAccumulator: 0 default, 1 dense, 2 sparse
    const int a = (int)value;
    if (a < 0 || a > 2) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_set: unknown SPGEMMAccumulator %d", a);
    if ((a == 1) != (h->algorithm == 1)) { h->numeric_bins_ready = false; h->entries_valid = false; }
    h->algorithm = a == 1 ? 1 : 0;
  } else if (k == "compression") {
    if (value != 0 && value != 1 && value != 2) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_set: compression is 0 (off), 1 (keep if it pays) or 2 (always)");
    h->compression = (int)value;
  } else if (k == "compression_cut_off") {
    if (!(value > 0.0 && value <= 1.0)) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_set: compression cut-off must be in (0, 1]");
    h->compression_cutoff = value;
  } else if (k == "verbose") {
    h->verbose = value < 0 ? 0 : (int)value;          // 2: also host-side stage times of the symbolic phase (adds stream synchronisations)
  } else if (k == "entries_computed") {
    // the reference's SPGEMMHandle::are_entries_computed() as the caller sees it: 0 = entries(C) must be written again by the next
    // numeric call (the caller re-allocated or overwrote them); 1 = leave the decision to the handle (it keeps them only when it
    // wrote them itself into the same arrays)
    if (value == 0) h->entries_valid = false;
  } else if (k == "sort_option" || k == "team_work_size" || k == "shmem_size" || k == "suggested_team_size" || k == "suggested_vector_size" ||
             k == "dynamic_scheduling" || k == "min_hash_size_scale" || k == "first_level_hash_cut_off" || k == "mkl_sort_option" ||
             k == "mkl_keep_output" || k == "mkl_convert_to_1base" || k == "multi_color_scale" || k == "read_write_cost_calc" ||
             k == "compression_steps" || k == "max_col_dense_acc") {
    // Hints of the reference (sparse/src/KokkosKernels_Handle.hpp:380-465, KokkosSparse_spgemm_handle.hpp:295-317,684): they size Kokkos
    // team launches, the two-level hash tables and the MKL path of the reference's kernels.  The reference's own TPL paths
    // (rocSPARSE, cuSPARSE) take and ignore them, and its driver and unit tests set them before every spgemm
    // (perf_test/sparse/KokkosSparse_spgemm.cpp:311-317,378-399, unit_test/Test_Sparse_spgemm.hpp:91-92): accepted, recorded
    // (kkamd_spgemm_get_hint), reported under verbose, without effect -- LDS tables and launch shapes follow the row bins, and rows of C
    // always leave sorted whatever sort_option says.
    h->hints[k] = value;
    if (h->verbose) printf("kkamd spgemm: hint %s = %g recorded (no effect on gfx950)\n", key, value);
  } else {
    return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_set: unknown key '%s'", key);
  }
  return KKAMD_OK;
}

int kkamd_spgemm_get_hint(kkamd_spgemm_handle_t* h, const char* key, double* value) {
  if (!h || !key || !value) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_get_hint: null argument");
  const auto it = h->hints.find(key);
  if (it == h->hints.end()) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_get_hint: '%s' was never set", key);
  *value = it->second;
  return KKAMD_OK;
}

int kkamd_spgemm_get(kkamd_spgemm_handle_t* h, int what, int64_t* value) {
  if (!h || !value) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_get: null pointer");
  switch (what) {
    case 0: *value = h->c_nnz; break;
    case 1: *value = h->mults; break;
    case 2: *value = h->max_row_flops; break;
    case 3: *value = h->max_row_nnz; break;
    case 4: *value = h->symbolic_called; break;
    case 5: *value = h->numeric_called; break;
    case 6: *value = h->compressed ? 1 : 0; break;
    case 7: *value = h->compressed_mults; break;
    case 8: *value = h->algorithm; break;
    case 9: *value = h->requested_algorithm; break;
    case 10: *value = (int64_t)h->hints.size(); break;
    case 11: *value = h->entries_reused ? 1 : 0; break;
    case 12: *value = h->bitmaps_used; break;
    case 13: *value = h->bm_stored; break;
    case 14: *value = h->pooled_used; break;
    case 15: *value = h->sorted_used; break;
    case 16: *value = h->n_dense_block; break;
    case 22: { kk::BmPool& pool = kk::bm_pool(); std::lock_guard<std::mutex> g(pool.m); *value = (int64_t)pool.bytes; } break;        // bytes the process-wide store holds now
    case 23: { kk::BmPool& pool = kk::bm_pool(); std::lock_guard<std::mutex> g(pool.m); *value = (int64_t)pool.high_water; } break;   // ... and the most it has held
    case 19: *value = h->last_units; break;           // units (row, window) of the last symbolic phase's dense class (0: the class was empty or went row by row)
    case 20: *value = h->last_unit_bitmaps; break;    // ... of which kept their bitmap for the numeric phase
    case 21: *value = h->last_unit_rows_kept; break;  // rows of the class whose every unit kept its structure
    case 17: *value = h->n_items_rank; break;        // ... of which as position-indexed items / column-indexed blocks
    case 18: *value = h->n_items_direct; break;       // rows of the last numeric call's bins that take the column-block value kernel
    default: return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_get: unknown query %d", what);
  }
  return KKAMD_OK;
}

// ---- row-partitioned SpGEMM over the GPUs of one node (the path shards by rows of A; B is replicated) -----------
struct kkamd_dist_spgemm {
  int world = 1, rank = 0;
  std::vector<int64_t> offsets;
  kkamd_spgemm_handle_t* h = nullptr;
};

int kkamd_dist_spgemm_partition(int64_t m, const void* d_row_mapA, const int32_t* d_entriesA, const void* d_row_mapB, int offset_type, int world,
                                int64_t* row_offsets, int64_t* mults_per_rank, kkamd_stream_t stream) {
  if (m < 0 || world < 1 || !row_offsets) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spgemm_partition: bad argument");
  if (offset_type != KKAMD_I32 && offset_type != KKAMD_I64) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spgemm_partition: unknown offset_type %d", offset_type);
  if (m > 0 && (!d_row_mapA || !d_row_mapB)) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spgemm_partition: null row_map");
  hipStream_t st = kk::to_hip(stream);
  std::vector<int64_t> flops((size_t)m);
  if (m > 0) {
    kk::DevBuf f_b, s_b, l_b;
    KK_HIP(f_b.alloc(sizeof(int64_t) * (size_t)m)); KK_HIP(s_b.alloc(3 * sizeof(unsigned long long))); KK_HIP(l_b.alloc(sizeof(int32_t) * (size_t)m));
    KK_HIP(hipMemsetAsync(s_b.p, 0, 3 * sizeof(unsigned long long), st));
    int64_t* d_f = f_b.as<int64_t>(); unsigned long long* d_s = s_b.as<unsigned long long>(); int32_t* d_l = l_b.as<int32_t>();
    const int64_t nbk = kk::ceil_div(m * 8, kk::kBlock);
    // rows above kFlopsLong entries: listed by the first kernel, a workgroup each; at most nnz(A) / kFlopsLong of them
    int64_t nnzA_h = 0;
    {
      unsigned char buf[8] = {0};
      const size_t osz = offset_type == KKAMD_I64 ? 8 : 4;
      KK_HIP(hipMemcpyAsync(buf, (const char*)d_row_mapA + osz * (size_t)m, osz, hipMemcpyDeviceToHost, st));
      KK_HIP(hipStreamSynchronize(st));
      nnzA_h = offset_type == KKAMD_I64 ? *(int64_t*)buf : (int64_t) * (int32_t*)buf;
    }
    const int64_t gl_cap = nnzA_h / kk::kFlopsLong + 1;
    const unsigned gl = (unsigned)(gl_cap < m ? gl_cap : m);
    if (offset_type == KKAMD_I64) {
      KK_LAUNCH((kk::spgemm_flops_kernel<int64_t>), (unsigned)(nbk < 4096 ? nbk : 4096), kk::kBlock, 0, st, m, (const int64_t*)d_row_mapA, d_entriesA, (const int64_t*)d_row_mapB, d_f, d_s, d_l, d_s + 2);
      KK_LAUNCH((kk::spgemm_flops_long_kernel<int64_t>), gl, kk::kBlock, 0, st, (const int32_t*)d_l, (const unsigned long long*)(d_s + 2), (const int64_t*)d_row_mapA, d_entriesA, (const int64_t*)d_row_mapB, d_f, d_s);
    } else {
      KK_LAUNCH((kk::spgemm_flops_kernel<int32_t>), (unsigned)(nbk < 4096 ? nbk : 4096), kk::kBlock, 0, st, m, (const int32_t*)d_row_mapA, d_entriesA, (const int32_t*)d_row_mapB, d_f, d_s, d_l, d_s + 2);
      KK_LAUNCH((kk::spgemm_flops_long_kernel<int32_t>), gl, kk::kBlock, 0, st, (const int32_t*)d_l, (const unsigned long long*)(d_s + 2), (const int32_t*)d_row_mapA, d_entriesA, (const int32_t*)d_row_mapB, d_f, d_s);
    }
    KK_HIP(hipMemcpyAsync(flops.data(), d_f, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
  }
  // contiguous slabs of near-equal multiplications: cut where the running sum passes r / world of the total
  long double total = 0; for (int64_t v : flops) total += (long double)v;
  row_offsets[0] = 0;
  long double run = 0; int64_t row = 0;
  for (int r = 1; r < world; ++r) {
    const long double target = total * (long double)r / (long double)world;
    while (row < m && run + (long double)flops[(size_t)row] <= target) { run += (long double)flops[(size_t)row]; ++row; }
    row_offsets[r] = row;
  }
  row_offsets[world] = m;
  if (mults_per_rank)
    for (int r = 0; r < world; ++r) { int64_t sum = 0; for (int64_t i = row_offsets[r]; i < row_offsets[r + 1]; ++i) sum += flops[(size_t)i]; mults_per_rank[r] = sum; }
  return KKAMD_OK;
}

int kkamd_dist_spgemm_create(kkamd_dist_spgemm_t** out, int world, int rank, const int64_t* row_offsets) {
  if (!out) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spgemm_create: null output pointer");
  *out = nullptr;
  if (!row_offsets || world < 1 || rank < 0 || rank >= world || row_offsets[0] != 0) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spgemm_create: bad partition");
  for (int r = 0; r < world; ++r) if (row_offsets[r + 1] < row_offsets[r]) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spgemm_create: row offsets must ascend");
  kkamd_dist_spgemm* op = new (std::nothrow) kkamd_dist_spgemm();
  if (!op) return kk::fail(KKAMD_ERR_ALLOC, "kkamd_dist_spgemm_create: out of host memory");
  op->world = world; op->rank = rank; op->offsets.assign(row_offsets, row_offsets + world + 1);
  const int rc = kkamd_spgemm_create(&op->h);
  if (rc) { delete op; return rc; }
  *out = op;
  return KKAMD_OK;
}
int kkamd_dist_spgemm_destroy(kkamd_dist_spgemm_t* op) {
  if (!op) return KKAMD_OK;
  (void)kkamd_spgemm_destroy(op->h);
  delete op;
  return KKAMD_OK;
}
kkamd_spgemm_handle_t* kkamd_dist_spgemm_handle(kkamd_dist_spgemm_t* op) { return op ? op->h : nullptr; }

static int dist_spgemm_rows(const kkamd_dist_spgemm_t* op, int64_t m_local, const char* who) {
  if (!op) return kk::fail(KKAMD_ERR_INVALID_ARG, "%s: null operator", who);
  const int64_t want = op->offsets[(size_t)op->rank + 1] - op->offsets[(size_t)op->rank];
  if (m_local != want) return kk::fail(KKAMD_ERR_INVALID_ARG, "%s: the slab of A has %lld rows, the partition gives rank %d %lld", who, (long long)m_local, op->rank, (long long)want);
  return KKAMD_OK;
}
int kkamd_dist_spgemm_symbolic(kkamd_dist_spgemm_t* op, int64_t m_local, int64_t n, int64_t k, const void* d_row_mapA_local, const int32_t* d_entriesA_local,
                               const void* d_row_mapB, const int32_t* d_entriesB, void* d_row_mapC_local, int offset_type, int64_t* c_nnz_local, kkamd_stream_t stream) {
  const int rc = dist_spgemm_rows(op, m_local, "kkamd_dist_spgemm_symbolic");
  if (rc) return rc;
  return kkamd_spgemm_symbolic(op->h, m_local, n, k, d_row_mapA_local, d_entriesA_local, d_row_mapB, d_entriesB, d_row_mapC_local, offset_type, c_nnz_local, stream);
}
int kkamd_dist_spgemm_numeric(kkamd_dist_spgemm_t* op, int64_t m_local, int64_t n, int64_t k, const void* d_row_mapA_local, const int32_t* d_entriesA_local,
                              const void* d_valuesA_local, const void* d_row_mapB, const int32_t* d_entriesB, const void* d_valuesB, const void* d_row_mapC_local,
                              int32_t* d_entriesC_local, void* d_valuesC_local, int offset_type, int value_type, kkamd_stream_t stream) {
  const int rc = dist_spgemm_rows(op, m_local, "kkamd_dist_spgemm_numeric");
  if (rc) return rc;
  return kkamd_spgemm_numeric(op->h, m_local, n, k, d_row_mapA_local, d_entriesA_local, d_valuesA_local, d_row_mapB, d_entriesB, d_valuesB, d_row_mapC_local,
                              d_entriesC_local, d_valuesC_local, offset_type, value_type, stream);
}
int kkamd_dist_spgemm_query(const kkamd_dist_spgemm_t* op, const char* key, int64_t* value) {
  if (!op || !key || !value) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spgemm_query: null argument");
  const std::string k(key);
  if (k == "row0") *value = op->offsets[(size_t)op->rank];
  else if (k == "rows_local") *value = op->offsets[(size_t)op->rank + 1] - op->offsets[(size_t)op->rank];
  else if (k == "rows_global") *value = op->offsets[(size_t)op->world];
  else if (k == "c_nnz_local") *value = op->h->c_nnz;
  else if (k == "mults_local") *value = op->h->mults;
  else return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_dist_spgemm_query: unknown key '%s'", key);
  return KKAMD_OK;
}

}  // extern "C"
