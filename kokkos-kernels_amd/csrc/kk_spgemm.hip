// kk_spgemm.hip -- C = A*B over CSR for gfx950: symbolic (row_map of C) + numeric (entries, values).
//
// Reference structure being replaced (sparse/impl/KokkosSparse_spgemm_impl_def.hpp:25-136): row-flop
// estimate (K8) -> [optional bit-compression of B] -> hash symbolic with a 16 KB LDS first level and a
// pooled global second level, one Kokkos thread per row (K10) -> prefix sum (K11) -> hash numeric in three
// team shapes (K12-K14) -> a separate per-row sort pass over C (K17).
//
// gfx950-native structure here.  A CU has 160 KB of LDS, so whole per-row hash tables live in LDS and
// no second level / memory pool exists:
//   1. spgemm_flops_kernel      upper bound per row, total multiplications, max.
//   2. rows are BINNED by that bound (symbolic) / by their exact nnz (numeric) and each bin gets the
//      launch shape that fits it:
//         wave  per row, 512-slot   LDS table  (4 rows per workgroup)        small rows
//         block per row, 4096-slot  LDS table                                medium rows
//         block per row, 32768-slot key table (symbolic) / 8192-slot key+value table (numeric)
//         block per row, DENSE bitmap (+ dense accumulator) in HBM            hub rows (R-MAT)
//      Open addressing, linear probing, hash (col*107) & mask, empty = -1 -- the same function as the
//      reference's linear-probe kernels (sparse/impl/KokkosSparse_spgemm_impl_kkmem.hpp:17,679).
//   3. exclusive scan of the counts -> row_map C, nnz(C) returned to the host.
//   4. numeric accumulates with LDS atomics (ds_cmpst / ds_add_f64), then orders each row INSIDE the
//      same kernel (rank-by-counting for wave rows, bitonic network for block rows, in-order bitmap walk
//      for dense rows), so C leaves the kernel column-sorted and the reference's extra sort pass over
//      C (numeric_spec.hpp:138-140) disappears.
// Symbolic results are exact (bit-identical row_map / entries to the SPGEMM_DEBUG oracle after its
// sort); numeric sums are order-dependent (atomics) and compared at 1e-6 relative.
#include "kk_common.h"
#include "kk_scan.h"
#include <climits>
#include <new>
#include <string>

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
constexpr int kSymWaveTable = 2048;   // symbolic keys only: 8 KB per wave
constexpr int kWaveTable = 512;       // numeric keys + values
constexpr int kSymBlkS = 4096,  kSymBlkL = 32768;
constexpr int kNumBlkS = 4096,  kNumBlkL = 8192;
constexpr int kDenseBlock = 1024;     // dense-row column kernel: 16 waves around one LDS bitmap
constexpr int kValBlock   = 512;      // dense-row value kernel
constexpr int kValTable   = 4096;     // value-window hash table (16 KB keys + 32 KB fp64 sums in LDS: 3 workgroups per CU)
constexpr int kValCap     = kValTable / 2;   // C entries per value window
constexpr int kValLa      = 512;      // A entries of a row whose B cursors live in LDS (10 KB)
constexpr int kValLong    = 128;      // B rows at least this long are streamed by a whole wave

struct SpgemmTuning {
  int win_bits       = 1 << 20;   // columns per LDS bitmap window (128 KB); rows wider than this take several passes
  int val_cap        = kValCap;   // C entries per value window
  int force_unsorted = 0;         // test hook: treat B as unsorted (dense rows accumulate in HBM)
  int debug          = 0;         // bench-only ablation bits for the dense-row kernels
};
static SpgemmTuning g_spgemm;

struct BinLimits { int64_t lim[kNumBins - 1]; };   // size <= lim[b] -> bin b  (lim[0] = 0)
static const BinLimits kSymLimits = {{0, (kSymWaveTable * 2) / 3, kSymBlkS / 2, kSymBlkL / 2}};
static const BinLimits kNumLimits = {{0, kWaveTable / 2, kNumBlkS / 2, (kNumBlkL * 2) / 3}};
// B sorted: everything above the small block table goes to the column + windowed value kernels (no 8192-slot bitonic sort)
static const BinLimits kNumLimitsSorted = {{0, kWaveTable / 2, kNumBlkS / 2, kNumBlkS / 2}};

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
                                                              unsigned long long* __restrict__ stats /*[0]=total,[1]=max*/) {
  __shared__ unsigned long long s_sum, s_max;
  if (threadIdx.x == 0) { s_sum = 0; s_max = 0; }
  __syncthreads();
  const int lane = threadIdx.x & 7;
  long long sum = 0, mx = 0;      // per work-item partials over the grid-stride loop: two global atomics per workgroup
  const int64_t stride = (int64_t)gridDim.x * (kBlock / 8);
  for (int64_t r0 = (int64_t)blockIdx.x * (kBlock / 8); r0 < m; r0 += stride) {     // workgroup-uniform trip count
    const int64_t row = r0 + threadIdx.x / 8;
    long long f = 0;
    if (row < m)
      for (int64_t a = (int64_t)rmA[row] + lane; a < (int64_t)rmA[row + 1]; a += 8) {
        const int32_t c = entA[a];
        f += (long long)rmB[c + 1] - (long long)rmB[c];
      }
    f = group_sum(f, 8);
    if (row < m && lane == 0) { flops[row] = f; sum += f; mx = f > mx ? f : mx; }
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

// are the rows of a CRS graph column-sorted (non-strict)?  8 lanes per row; *unsorted is set to 1 otherwise.
template <class OffT>
__global__ __launch_bounds__(kBlock) void rows_sorted_kernel(int64_t n, const OffT* __restrict__ rm,
                                                             const int32_t* __restrict__ ent, int* __restrict__ unsorted) {
  const int lane = threadIdx.x & 7;
  bool bad = false;
  for (int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / 8; row < n; row += (int64_t)gridDim.x * (kBlock / 8)) {
    const int64_t e = (int64_t)rm[row + 1];
    for (int64_t j = (int64_t)rm[row] + lane; j + 1 < e; j += 8) bad |= ent[j] > ent[j + 1];
  }
  if (bad) *unsorted = 1;
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
__device__ __forceinline__ bool hash_insert_key(int* tab, int mask, int key) {   // true if the key is new
  int h = (int)(((unsigned)key * (unsigned)kHashMul) & (unsigned)mask);
  while (true) {
    const int old = atomicCAS(&tab[h], -1, key);
    if (old == -1) return true;
    if (old == key) return false;
    h = (h + 1) & mask;
  }
}
template <class VT> __device__ __forceinline__ void hash_accumulate(int* keys, VT* vals, int mask, int key, VT v) {
  int h = (int)(((unsigned)key * (unsigned)kHashMul) & (unsigned)mask);
  while (true) {
    const int old = atomicCAS(&keys[h], -1, key);
    if (old == -1 || old == key) { KK_ATOMIC_FADD(&vals[h], v); return; }
    h = (h + 1) & mask;
  }
}

// Visit every product column of A(row,:)*B with `nthreads` cooperating work-items (id tid): sub-groups of 2^s lanes
// share an A entry and stride over that B row.  s is at least sg_log2 (the matrix-wide hint: average B row length) and
// grows for rows of A with few entries so that the sub-groups (nthreads >> s of them) just cover the row -- a row with
// 13 entries handled by 1024 work-items runs 16 sub-groups of 64 lanes instead of leaving 115 of 128 idle.
// Each lane issues kProdUnroll independent B loads per step (a step is latency-bound otherwise): f(a, j, column).
constexpr int kProdUnroll = 4;
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

// ------------------------------------------------------------------------------------------------
// 3. symbolic kernels
template <class OffT>
__global__ __launch_bounds__(kBlock) void spgemm_sym_wave_kernel(int64_t nbin, const int32_t* __restrict__ perm,
                                                                 const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                 const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                                                 OffT* __restrict__ counts, int sg_log2) {
  constexpr int H = kSymWaveTable;
  __shared__ int tab[kBlock / 64][H];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int64_t idx = (int64_t)blockIdx.x * (kBlock / 64) + w;
  for (int i = lane; i < H; i += 64) tab[w][i] = -1;
  __syncthreads();
  int cnt = 0;
  int64_t row = -1;
  if (idx < nbin) {
    row = perm[idx];
    int* mytab = tab[w];
    for_each_product<OffT>(row, rmA, entA, rmB, entB, lane, 64, sg_log2,
                           [&](int64_t, int64_t, int c) { cnt += hash_insert_key(mytab, H - 1, c) ? 1 : 0; });
  }
  cnt = group_sum(cnt, 64);
  if (idx < nbin && lane == 0) counts[row] = (OffT)cnt;
}

template <class OffT, int H, int NT>
__global__ __launch_bounds__(NT) void spgemm_sym_block_kernel(int64_t nbin, const int32_t* __restrict__ perm,
                                                                  const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                  const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                                                  OffT* __restrict__ counts, int sg_log2) {
  __shared__ int tab[H];
  __shared__ int s_count;
  const int t = threadIdx.x;
  const int64_t row = perm[blockIdx.x];
  for (int i = t; i < H; i += NT) tab[i] = -1;
  if (t == 0) s_count = 0;
  __syncthreads();
  int cnt = 0;
  for_each_product<OffT>(row, rmA, entA, rmB, entB, t, NT, sg_log2,
                         [&](int64_t, int64_t, int c) { cnt += hash_insert_key(tab, H - 1, c) ? 1 : 0; });
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
template <class OffT, bool EMIT>
__global__ __launch_bounds__(kDenseBlock) void spgemm_dense_cols_kernel(const int32_t* __restrict__ perm,
                                                                        const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                        const OffT* __restrict__ rmB, const int32_t* __restrict__ entB,
                                                                        OffT* __restrict__ counts, const OffT* __restrict__ rmC,
                                                                        int32_t* __restrict__ entC, int64_t k, int win_bits,
                                                                        int sg_log2, int debug) {
  KK_DYN_SMEM(kk_u64, bm);
  __shared__ int s_min, s_max;
  __shared__ int s_wave[kDenseBlock / 64];
  const int t       = threadIdx.x;
  const int64_t row = perm[blockIdx.x];
  int64_t total     = 0;
  for (int64_t c0 = 0; c0 < k; c0 += win_bits) {
    const int nbits  = (int)((k - c0 < (int64_t)win_bits) ? k - c0 : (int64_t)win_bits);
    const int nwords = (nbits + 63) >> 6;
    for (int i = t; i < nwords; i += kDenseBlock) bm[i] = 0ull;
    if (t == 0) { s_min = INT_MAX; s_max = -1; }
    __syncthreads();
    int cmin = INT_MAX, cmax = -1;
    if (!(debug & 2)) for_each_product<OffT>(row, rmA, entA, rmB, entB, t, kDenseBlock, sg_log2, [&](int64_t, int64_t, int cb) {
      const int64_t c64 = (int64_t)cb - c0;
      if (c64 >= 0 && c64 < nbits) {
        const int c = (int)c64;
        atomicOr(&bm[c >> 6], 1ull << (c & 63));
        cmin = c < cmin ? c : cmin; cmax = c > cmax ? c : cmax;
      }
    });
    if (cmax >= 0) { atomicMin(&s_min, cmin); atomicMax(&s_max, cmax); }
    __syncthreads();
    if (s_max >= 0 && !(debug & 4)) {
      const int w_lo = s_min >> 6, w_hi = s_max >> 6;
      for (int wb = w_lo; wb <= w_hi; wb += kDenseBlock) {
        const int wd = wb + t;
        kk_u64 v     = (wd <= w_hi) ? bm[wd] : 0ull;
        int tot;
        const int excl = block_exclusive_scan_n<int, kDenseBlock>(__popcll(v), &tot, s_wave);
        if (EMIT && !(debug & 1)) {
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
    __syncthreads();
  }
  if (!EMIT && t == 0) counts[row] = (OffT)total;
}

// ------------------------------------------------------------------------------------------------
// 4. numeric kernels
template <class OffT, class VT>
__global__ __launch_bounds__(kBlock) void spgemm_num_wave_kernel(int64_t nbin, const int32_t* __restrict__ perm,
                                                                 const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                 const VT* __restrict__ valA, const OffT* __restrict__ rmB,
                                                                 const int32_t* __restrict__ entB, const VT* __restrict__ valB,
                                                                 const OffT* __restrict__ rmC, int32_t* __restrict__ entC,
                                                                 VT* __restrict__ valC, int sg_log2) {
  constexpr int H = kWaveTable;
  __shared__ int keys[kBlock / 64][H];
  __shared__ VT vals[kBlock / 64][H];
  __shared__ int ckey[kBlock / 64][H / 2];
  __shared__ unsigned short cslot[kBlock / 64][H / 2];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int64_t idx = (int64_t)blockIdx.x * (kBlock / 64) + w;
  for (int i = lane; i < H; i += 64) { keys[w][i] = -1; vals[w][i] = VT(0); }
  __syncthreads();
  if (idx < nbin) {
    const int64_t row = perm[idx];
    int* mk = keys[w]; VT* mv = vals[w];
    for_each_product_v<OffT, VT>(row, rmA, entA, rmB, entB, valB, lane, 64, sg_log2,
                                 [&](int64_t a, int c, VT bv) { hash_accumulate<VT>(mk, mv, H - 1, c, valA[a] * bv); });
  }
  __syncthreads();
  // compact the occupied slots (ballot + prefix), then rank-by-counting over the compact list only: keys are unique,
  // so rank = number of smaller keys.  ceil(n/64) * n compares per wave instead of 8 * 512 over the whole table.
  int n = 0;
  for (int s0 = 0; s0 < H; s0 += 64) {
    const int key        = keys[w][s0 + lane];
    const kk_u64 occ     = __ballot(key >= 0);
    if (key >= 0) {
      const int pos = n + __popcll(occ & ((1ull << lane) - 1ull));
      ckey[w][pos]  = key;
      cslot[w][pos] = (unsigned short)(s0 + lane);
    }
    n += __popcll(occ);
  }
  KK_WAVE_SYNC();
  if (idx < nbin) {
    const int64_t row  = perm[idx];
    const int64_t base = (int64_t)rmC[row];
    for (int i = lane; i < n; i += 64) {
      const int key = ckey[w][i];
      int rank = 0;
      for (int q = 0; q < n; ++q) rank += (ckey[w][q] < key) ? 1 : 0;
      entC[base + rank] = key;
      valC[base + rank] = vals[w][cslot[w][i]];
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
  const int t = threadIdx.x;
  const int64_t row = perm[blockIdx.x];
  for (int i = t; i < H; i += kBlock) { keys[i] = -1; vals[i] = VT(0); }
  __syncthreads();
  for_each_product_v<OffT, VT>(row, rmA, entA, rmB, entB, valB, t, kBlock, sg_log2,
                               [&](int64_t a, int c, VT bv) { hash_accumulate<VT>(keys, vals, H - 1, c, valA[a] * bv); });
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

// dense rows: bitmap + dense accumulator per workgroup in HBM.  Each lane owns one contiguous chunk of the touched
// bitmap range: pass 1 counts its set bits, one workgroup scan turns the counts into output offsets, pass 2 walks
// the chunk again and emits (column, value) pairs in ascending order -- sorted output without a sort.  Bitmap and
// accumulator are read with L2-served loads (the atomics that built them were performed in L2) and cleared with
// plain stores; the closing barrier (s_waitcnt vmcnt(0) + s_barrier) orders those stores before the next row.
template <class VT> __device__ __forceinline__ VT load_l2(const VT* p);
template <> __device__ __forceinline__ double load_l2<double>(const double* p) {
  return __longlong_as_double((long long)KK_LOAD_L2(reinterpret_cast<const kk_u64*>(p)));
}
template <> __device__ __forceinline__ float load_l2<float>(const float* p) {
  return __int_as_float((int)KK_LOAD_L2(reinterpret_cast<const unsigned*>(p)));
}
template <class OffT, class VT>
__global__ __launch_bounds__(kBlock) void spgemm_num_dense_kernel(int64_t nbin, const int32_t* __restrict__ perm,
                                                                  const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                  const VT* __restrict__ valA, const OffT* __restrict__ rmB,
                                                                  const int32_t* __restrict__ entB, const VT* __restrict__ valB,
                                                                  const OffT* __restrict__ rmC, int32_t* __restrict__ entC,
                                                                  VT* __restrict__ valC, kk_u64* __restrict__ bitmaps,
                                                                  VT* __restrict__ accs, int64_t words, int64_t k, int sg_log2) {
  __shared__ int s_min, s_max;
  __shared__ int s_wave[kBlock / 64];
  const int t = threadIdx.x;
  kk_u64* bm  = bitmaps + (int64_t)blockIdx.x * words;
  VT* acc     = accs + (int64_t)blockIdx.x * k;
  for (int64_t ri = blockIdx.x; ri < nbin; ri += gridDim.x) {
    const int64_t row = perm[ri];
    if (t == 0) { s_min = INT_MAX; s_max = -1; }
    __syncthreads();
    int cmin = INT_MAX, cmax = -1;
    for_each_product_v<OffT, VT>(row, rmA, entA, rmB, entB, valB, t, kBlock, sg_log2, [&](int64_t a, int c, VT bv) {
      atomicOr(&bm[c >> 6], 1ull << (c & 63));
      KK_ATOMIC_FADD(&acc[c], valA[a] * bv);
      cmin = c < cmin ? c : cmin; cmax = c > cmax ? c : cmax;
    });
    if (cmax >= 0) { atomicMin(&s_min, cmin); atomicMax(&s_max, cmax); }
    __syncthreads();
    int64_t a = 0, z = 0;
    int cnt = 0;
    if (s_max >= 0) {
      const int64_t w_lo = s_min >> 6, nw = (s_max >> 6) - w_lo + 1, per = (nw + kBlock - 1) / kBlock;
      a = w_lo + t * per; z = (a + per < w_lo + nw) ? a + per : w_lo + nw;
      for (int64_t wd = a; wd < z; ++wd) cnt += __popcll(KK_LOAD_L2(&bm[wd]));
    }
    int tot;
    const int excl = block_exclusive_scan<int>(cnt, &tot, s_wave);
    int64_t pos    = (int64_t)rmC[row] + excl;
    for (int64_t wd = a; wd < z; ++wd) {
      kk_u64 v = KK_LOAD_L2(&bm[wd]);
      if (!v) continue;
      bm[wd] = 0ull;
      while (v) {
        const int bit = __ffsll(v) - 1;
        const int c   = (int)(wd * 64 + bit);
        entC[pos]     = c;
        valC[pos]     = load_l2<VT>(&acc[c]);
        acc[c]        = VT(0);
        ++pos;
        v &= v - 1;
      }
    }
    __syncthreads();
  }
}

// dense rows, values (B rows column-sorted): entries(C) of the row are already in place and sorted
// (spgemm_dense_cols_kernel<EMIT>), so the row is cut into windows of `cap` consecutive C entries.  A window's columns
// are hashed into an LDS table with zeroed sums.  Sub-groups of 8..64 lanes (fewer A entries -> wider groups) each own
// one A entry at a time and stream the part of that B row whose columns fall inside the window (coalesced, sg entries
// per step), probe the table (the column is known to be present) and accumulate with ds_add; a sub-group that runs
// out of in-window entries moves to its next A entry, so a wave keeps up to 8 independent B streams in flight.  The
// resume point of every A entry (position in B, entries left, A value) lives in LDS for the first kValLa entries of
// the row and in an int32 HBM cursor (indexed like entries(A)) beyond that, so every B entry is read exactly once.
// Finally the sums are looked up in C order and leave coalesced.
template <class OffT, class VT>
__global__ __launch_bounds__(kValBlock) void spgemm_dense_vals_kernel(const int32_t* __restrict__ perm,
                                                                      const OffT* __restrict__ rmA, const int32_t* __restrict__ entA,
                                                                      const VT* __restrict__ valA, const OffT* __restrict__ rmB,
                                                                      const int32_t* __restrict__ entB, const VT* __restrict__ valB,
                                                                      const OffT* __restrict__ rmC, const int32_t* __restrict__ entC,
                                                                      VT* __restrict__ valC, int32_t* __restrict__ cursors, int cap) {
  constexpr int H = kValTable;
  __shared__ int hk[H];
  __shared__ VT hv[H];
  __shared__ long long s_cur[kValLa];
  __shared__ int s_rem[kValLa];
  __shared__ VT s_av[kValLa];
  __shared__ unsigned char s_long[kValLa];
  __shared__ int s_whi;
  constexpr int UL = 4, US = 2;       // independent B loads per lane and step (long / short B rows)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t row = perm[blockIdx.x];
  const int64_t a0 = (int64_t)rmA[row], la = (int64_t)rmA[row + 1] - a0;
  const int64_t la_c = la < kValLa ? la : kValLa;
  const int64_t base = (int64_t)rmC[row], cnt = (int64_t)rmC[row + 1] - base;
  int sg_log2 = 3;
  while ((int64_t)(kValBlock >> (sg_log2 + 1)) >= la && sg_log2 < 6) ++sg_log2;
  const int sg = 1 << sg_log2, sub = t >> sg_log2, nsub = kValBlock >> sg_log2, sl = t & (sg - 1);
  const int sg_shift   = lane & ~(sg - 1);
  const kk_u64 sg_mask = sg == 64 ? ~0ull : ((1ull << sg) - 1ull);
  for (int64_t a = t; a < la_c; a += kValBlock) {
    const int32_t kc = entA[a0 + a];
    const int64_t b0 = (int64_t)rmB[kc];
    const int len    = (int)((int64_t)rmB[kc + 1] - b0);
    s_cur[a] = b0; s_rem[a] = len; s_av[a] = valA[a0 + a]; s_long[a] = len >= kValLong ? 1 : 0;
  }
  auto accumulate = [&](int c, VT v) {
    int hh = (int)(((unsigned)c * (unsigned)kHashMul) & (unsigned)(H - 1));
    int probes = 0;
    while (hk[hh] != c && probes < H) { hh = (hh + 1) & (H - 1); ++probes; }
    if (probes < H) KK_ATOMIC_FADD(&hv[hh], v);
  };
  for (int64_t done = 0; done < cnt; done += cap) {
    const int n = (int)(cnt - done < (int64_t)cap ? cnt - done : (int64_t)cap);
    for (int i = t; i < H; i += kValBlock) { hk[i] = -1; hv[i] = VT(0); }
    __syncthreads();
    for (int i = t; i < n; i += kValBlock) {
      const int key = entC[base + done + i];
      (void)hash_insert_key(hk, H - 1, key);
      if (i == n - 1) s_whi = key;
    }
    __syncthreads();
    const int whi    = s_whi;
    const bool first = done == 0, last = done + n >= cnt;
    // long B rows: one wave per A entry, UL * 64 consecutive B entries per step
    for (int64_t a = wave; a < la_c; a += kValBlock / 64) {
      if (!s_long[a]) continue;
      int64_t p = s_cur[a];
      int rem   = s_rem[a];
      const VT av = s_av[a];
      while (true) {
        int c[UL];
        KK_UNROLL
        for (int u = 0; u < UL; ++u) { const int idx = u * 64 + lane; c[u] = idx < rem ? entB[p + idx] : INT_MAX; }
        int nin = 0;
        KK_UNROLL
        for (int u = 0; u < UL; ++u) {
          const bool in = c[u] <= whi;
          if (in) accumulate(c[u], av * valB[p + u * 64 + lane]);
          nin += __popcll(__ballot(in));
        }
        p += nin; rem -= nin;
        if (nin < UL * 64) break;
      }
      if (!last && lane == 0) { s_cur[a] = p; s_rem[a] = rem; }
    }
    // short B rows (and every A entry beyond the LDS cursor cache): persistent sub-groups
    int64_t a = sub, p = 0, b0 = 0;
    int rem = 0;
    VT av   = VT(0);
    bool have = false;
    auto fetch = [&]() {
      while (a < la_c && s_long[a]) a += nsub;
      have = a < la;
      if (!have) return;
      if (a < kValLa) { p = s_cur[a]; rem = s_rem[a]; av = s_av[a]; }
      else {
        const int32_t kc = entA[a0 + a];
        b0               = (int64_t)rmB[kc];
        const int off    = first ? 0 : cursors[a0 + a];
        p = b0 + off; rem = (int)((int64_t)rmB[kc + 1] - b0) - off; av = valA[a0 + a];
      }
    };
    fetch();
    while (true) {
      int c[US];
      KK_UNROLL
      for (int u = 0; u < US; ++u) { const int idx = u * sg + sl; c[u] = (have && idx < rem) ? entB[p + idx] : INT_MAX; }
      int nin = 0;
      KK_UNROLL
      for (int u = 0; u < US; ++u) {
        const bool in = c[u] <= whi;
        if (in) accumulate(c[u], av * valB[p + u * sg + sl]);
        nin += __popcll((__ballot(in) >> sg_shift) & sg_mask);
      }
      if (have) {
        p += nin; rem -= nin;
        if (nin < US * sg) {       // this A entry has nothing more inside the window: park it, take the next one
          if (!last && sl == 0) {
            if (a < kValLa) { s_cur[a] = p; s_rem[a] = rem; }
            else cursors[a0 + a] = (int32_t)(p - b0);
          }
          a += nsub;
          fetch();
        }
      }
      if (__ballot(have) == 0ull) break;
    }
    __syncthreads();
    for (int i = t; i < n; i += kValBlock) {
      const int key = entC[base + done + i];
      int hh = (int)(((unsigned)key * (unsigned)kHashMul) & (unsigned)(H - 1));
      while (hk[hh] != key) hh = (hh + 1) & (H - 1);
      valC[base + done + i] = hv[hh];
    }
    __syncthreads();
  }
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
  int64_t nnzA = 0;
  bool b_sorted = false;           // rows of B column-sorted: dense rows may use the windowed LDS value kernel
  bool dense_lds = false;          // decided when the numeric bins are made
};

namespace kk {

static int pick_sg_log2(int64_t nnzB, int64_t n) {
  const int64_t avg = n > 0 ? nnzB / n : 1;
  int l = 0;
  while ((1 << l) < 64 && (1 << (l + 1)) <= avg) ++l;    // largest power of two <= average B row length
  return l;
}

static int make_bins(int64_t m, const int64_t* d_sizes, int64_t cap, const BinLimits& L, int32_t* d_perm,
                     BinOffsets* off, hipStream_t st) {
  unsigned long long* d_cnt = nullptr;
  KK_HIP(hipMalloc((void**)&d_cnt, sizeof(unsigned long long) * 2 * kNumBins));
  KK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long) * 2 * kNumBins, st));
  const unsigned grid = (unsigned)ceil_div(m, kBlock);
  KK_LAUNCH(spgemm_bin_count_kernel, grid, kBlock, 0, st, m, d_sizes, cap, L, d_cnt);
  unsigned long long h_cnt[kNumBins];
  KK_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  off->off[0] = 0;
  for (int b = 0; b < kNumBins; ++b) off->off[b + 1] = off->off[b] + (int64_t)h_cnt[b];
  KK_LAUNCH(spgemm_bin_scatter_kernel, grid, kBlock, 0, st, m, d_sizes, cap, L, *off, d_cnt + kNumBins, d_perm);
  hipError_t e = hipGetLastError();
  hipError_t e2 = hipStreamSynchronize(st);
  (void)hipFree(d_cnt);
  if (e != hipSuccess || e2 != hipSuccess) return fail(KKAMD_ERR_HIP, "spgemm binning failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  return KKAMD_OK;
}

// number of dense-row workgroups and their workspace, bounded to 1/4 of free HBM (and 2048 workgroups = 8 per CU)
static int dense_geometry(int64_t nrows_dense, int64_t bytes_per_wg, int* nwg) {
  size_t free_b = 0, total_b = 0;
  KK_HIP(hipMemGetInfo(&free_b, &total_b));
  int64_t cap = (int64_t)(free_b / 4) / (bytes_per_wg > 0 ? bytes_per_wg : 1);
  if (cap < 1) return fail(KKAMD_ERR_ALLOC, "spgemm: not enough device memory for one dense accumulator (%lld bytes)", (long long)bytes_per_wg);
  int64_t g = nrows_dense < 2048 ? nrows_dense : 2048;
  if (g > cap) g = cap;
  *nwg = (int)g;
  return KKAMD_OK;
}

// one workgroup per dense row; dynamic LDS = the bitmap window
template <class OffT, bool EMIT>
static int launch_dense_cols(int64_t nrows, const int32_t* perm, const OffT* rmA, const int32_t* entA, const OffT* rmB,
                             const int32_t* entB, OffT* counts, const OffT* rmC, int32_t* entC, int64_t k, int sg, hipStream_t st) {
  int64_t win = g_spgemm.win_bits;
  if (win > k) win = ceil_div(k, 64) * 64;
  const size_t smem = (size_t)(win / 8);
#ifndef KK_EMU
  KK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spgemm_dense_cols_kernel<OffT, EMIT>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
#endif
  KK_LAUNCH((spgemm_dense_cols_kernel<OffT, EMIT>), (unsigned)nrows, kDenseBlock, smem, st, perm, rmA, entA, rmB, entB, counts,
            rmC, entC, k, (int)win, sg, g_spgemm.debug);
  return KKAMD_OK;
}

template <class OffT>
static int symbolic_typed(kkamd_spgemm_handle* h, int64_t m, int64_t n, int64_t k, const void* rmA_, const int32_t* entA,
                          const void* rmB_, const int32_t* entB, void* rmC_, int64_t nnzB, int64_t* c_nnz, hipStream_t st) {
  const OffT* rmA = (const OffT*)rmA_;
  const OffT* rmB = (const OffT*)rmB_;
  OffT* rmC       = (OffT*)rmC_;
  KK_HIP(hipMemsetAsync(rmC, 0, sizeof(OffT) * (size_t)(m + 1), st));
  unsigned long long* d_stats = nullptr;
  KK_HIP(hipMalloc((void**)&d_stats, 2 * sizeof(unsigned long long)));
  KK_HIP(hipMemsetAsync(d_stats, 0, 2 * sizeof(unsigned long long), st));
  {
    const int64_t nbk = ceil_div(m * 8, kBlock);
    KK_LAUNCH((spgemm_flops_kernel<OffT>), (unsigned)(nbk < 4096 ? nbk : 4096), kBlock, 0, st, m, rmA, entA, rmB, h->d_sizes, d_stats);
  }
  unsigned long long h_stats[2] = {0, 0};
  KK_HIP(hipMemcpyAsync(h_stats, d_stats, sizeof h_stats, hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  KK_HIP(hipFree(d_stats));
  h->mults = (int64_t)h_stats[0]; h->max_row_flops = (int64_t)h_stats[1];
  h->sg_log2 = pick_sg_log2(nnzB, n);

  BinOffsets off;
  int rc = make_bins(m, h->d_sizes, k, kSymLimits, h->d_perm, &off, st);   // a C row cannot exceed k columns
  if (rc) return rc;
  const int sg = h->sg_log2;
  auto nb = [&](int b) { return off.off[b + 1] - off.off[b]; };
  if (nb(1)) KK_LAUNCH((spgemm_sym_wave_kernel<OffT>), (unsigned)ceil_div(nb(1), kBlock / 64), kBlock, 0, st, nb(1),
                       (const int32_t*)(h->d_perm + off.off[1]), rmA, entA, rmB, entB, rmC, sg);
  if (nb(2)) KK_LAUNCH((spgemm_sym_block_kernel<OffT, kSymBlkS, kBlock>), (unsigned)nb(2), kBlock, 0, st, nb(2),
                       (const int32_t*)(h->d_perm + off.off[2]), rmA, entA, rmB, entB, rmC, sg);
  if (nb(3)) KK_LAUNCH((spgemm_sym_block_kernel<OffT, kSymBlkL, kDenseBlock>), (unsigned)nb(3), kDenseBlock, 0, st, nb(3),
                       (const int32_t*)(h->d_perm + off.off[3]), rmA, entA, rmB, entB, rmC, sg);
  if (nb(4)) {
    if ((rc = launch_dense_cols<OffT, false>(nb(4), h->d_perm + off.off[4], rmA, entA, rmB, entB, rmC, (const OffT*)nullptr,
                                             (int32_t*)nullptr, k, sg, st))) return rc;
  }
  // sortedness of B decides how the numeric phase handles dense rows
  {
    int* d_flag = nullptr; int h_flag = 0;
    KK_HIP(hipMalloc((void**)&d_flag, sizeof(int)));
    KK_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), st));
    const int64_t nbk = ceil_div(n * 8, kBlock);
    KK_LAUNCH((rows_sorted_kernel<OffT>), (unsigned)(nbk < 8192 ? nbk : 8192), kBlock, 0, st, n, rmB, entB, d_flag);
    KK_HIP(hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    KK_HIP(hipStreamSynchronize(st));
    KK_HIP(hipFree(d_flag));
    h->b_sorted = h_flag == 0;
  }
  hipError_t e = hipGetLastError();
  rc = (e == hipSuccess) ? exclusive_scan_inplace<OffT>(rmC, m + 1, st) : fail(KKAMD_ERR_HIP, "spgemm symbolic launch failed: %s", hipGetErrorString(e));
  OffT total = 0;
  if (rc == KKAMD_OK) {
    hipError_t e1 = hipMemcpyAsync(&total, rmC + m, sizeof(OffT), hipMemcpyDeviceToHost, st);
    hipError_t e2 = hipStreamSynchronize(st);
    if (e1 != hipSuccess || e2 != hipSuccess) rc = fail(KKAMD_ERR_HIP, "spgemm symbolic failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
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
    h->dense_lds = h->b_sorted && !g_spgemm.force_unsorted;
    if ((rc = make_bins(m, h->d_sizes, INT64_MAX, h->dense_lds ? kNumLimitsSorted : kNumLimits, h->d_perm, &h->num_off, st))) return rc;
    h->numeric_bins_ready = true;
  }
  const BinOffsets& off = h->num_off;
  const int sg = h->sg_log2;
  auto nb = [&](int b) { return off.off[b + 1] - off.off[b]; };
  if (nb(1)) KK_LAUNCH((spgemm_num_wave_kernel<OffT, VT>), (unsigned)ceil_div(nb(1), kBlock / 64), kBlock, 0, st, nb(1),
                       (const int32_t*)(h->d_perm + off.off[1]), rmA, entA, valA, rmB, entB, valB, rmC, entC, valC, sg);
  if (nb(2)) KK_LAUNCH((spgemm_num_block_kernel<OffT, VT, kNumBlkS>), (unsigned)nb(2), kBlock, 0, st, nb(2),
                       (const int32_t*)(h->d_perm + off.off[2]), rmA, entA, valA, rmB, entB, valB, rmC, entC, valC, sg);
  if (nb(3)) KK_LAUNCH((spgemm_num_block_kernel<OffT, VT, kNumBlkL>), (unsigned)nb(3), kBlock, 0, st, nb(3),
                       (const int32_t*)(h->d_perm + off.off[3]), rmA, entA, valA, rmB, entB, valB, rmC, entC, valC, sg);
  kk_u64* d_bm = nullptr; VT* d_acc = nullptr; int32_t* d_cur = nullptr;
  if (nb(4) && h->dense_lds) {
    const int32_t* dperm = h->d_perm + off.off[4];
    if ((rc = launch_dense_cols<OffT, true>(nb(4), dperm, rmA, entA, rmB, entB, (OffT*)nullptr, rmC, entC, k, sg, st))) return rc;
    KK_HIP(hipMalloc((void**)&d_cur, sizeof(int32_t) * (size_t)(h->nnzA > 0 ? h->nnzA : 1)));
    int cap = g_spgemm.val_cap;
    if (cap < 64) cap = 64;
    if (cap > kValCap) cap = kValCap;
    KK_LAUNCH((spgemm_dense_vals_kernel<OffT, VT>), (unsigned)nb(4), kValBlock, 0, st, dperm, rmA, entA, valA, rmB, entB, valB,
              rmC, (const int32_t*)entC, valC, d_cur, cap);
  } else if (nb(4)) {
    const int64_t words = ceil_div(k, 64);
    int nwg = 1;
    if ((rc = dense_geometry(nb(4), words * 8 + k * (int64_t)sizeof(VT), &nwg))) return rc;
    KK_HIP(hipMalloc((void**)&d_bm, (size_t)words * 8 * (size_t)nwg));
    KK_HIP(hipMalloc((void**)&d_acc, (size_t)k * sizeof(VT) * (size_t)nwg));
    KK_HIP(hipMemsetAsync(d_bm, 0, (size_t)words * 8 * (size_t)nwg, st));
    KK_HIP(hipMemsetAsync(d_acc, 0, (size_t)k * sizeof(VT) * (size_t)nwg, st));
    KK_LAUNCH((spgemm_num_dense_kernel<OffT, VT>), (unsigned)nwg, kBlock, 0, st, nb(4), (const int32_t*)(h->d_perm + off.off[4]),
              rmA, entA, valA, rmB, entB, valB, rmC, entC, valC, d_bm, d_acc, words, k, sg);
  }
  hipError_t e = hipGetLastError();
  hipError_t e2 = hipStreamSynchronize(st);   // the reference's numeric phase fences too (impl_kkmem.hpp:1440,1467)
  if (d_bm) (void)hipFree(d_bm);
  if (d_acc) (void)hipFree(d_acc);
  if (d_cur) (void)hipFree(d_cur);
  if (e != hipSuccess || e2 != hipSuccess) return fail(KKAMD_ERR_HIP, "spgemm numeric failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
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
    if (value < 64 || value > kValCap) return fail(KKAMD_ERR_INVALID_ARG, "spgemm_val_cap must be in [64, %d]", kValCap);
    g_spgemm.val_cap = value;
  } else if (k == "spgemm_force_unsorted") g_spgemm.force_unsorted = value != 0;
  else if (k == "spgemm_debug") g_spgemm.debug = value;
  else return fail(KKAMD_ERR_INVALID_ARG, "kkamd_set_default: unknown key '%s'", k.c_str());
  return KKAMD_OK;
}

}  // namespace kk

extern "C" {

int kkamd_spgemm_create(kkamd_spgemm_handle_t** handle) {
  if (!handle) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_create: null output pointer");
  *handle = new (std::nothrow) kkamd_spgemm_handle();
  return *handle ? KKAMD_OK : kk::fail(KKAMD_ERR_ALLOC, "kkamd_spgemm_create: out of host memory");
}

int kkamd_spgemm_destroy(kkamd_spgemm_handle_t* h) {
  if (!h) return KKAMD_OK;
  if (h->d_sizes) (void)hipFree(h->d_sizes);
  if (h->d_perm) (void)hipFree(h->d_perm);
  delete h;
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
  const size_t osz = offset_type == KKAMD_I64 ? 8 : 4;
  // idempotent: a second symbolic on the same handle returns the stored answer
  // (sparse/impl/KokkosSparse_spgemm_symbolic_spec.hpp:99)
  if (h->symbolic_called && h->m == m && h->n == n && h->k == k && h->rmA == d_row_mapA && h->rmB == d_row_mapB) {
    if (c_nnz) *c_nnz = h->c_nnz;
    return KKAMD_OK;
  }
  h->m = m; h->n = n; h->k = k; h->offset_type = offset_type; h->rmA = d_row_mapA; h->rmB = d_row_mapB;
  h->numeric_called = false; h->numeric_bins_ready = false;
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
  if (h->d_sizes) { (void)hipFree(h->d_sizes); h->d_sizes = nullptr; }
  if (h->d_perm) { (void)hipFree(h->d_perm); h->d_perm = nullptr; }
  KK_HIP(hipMalloc((void**)&h->d_sizes, sizeof(int64_t) * (size_t)m));
  KK_HIP(hipMalloc((void**)&h->d_perm, sizeof(int32_t) * (size_t)m));
  int64_t total = 0;
  int rc = offset_type == KKAMD_I64
               ? kk::symbolic_typed<int64_t>(h, m, n, k, d_row_mapA, d_entriesA, d_row_mapB, d_entriesB, d_row_mapC, nnzB, &total, st)
               : kk::symbolic_typed<int32_t>(h, m, n, k, d_row_mapA, d_entriesA, d_row_mapB, d_entriesB, d_row_mapC, nnzB, &total, st);
  if (rc) return rc;
  if (offset_type == KKAMD_I32 && total < 0)
    return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_symbolic: nnz(C) overflows 32-bit offsets; use 64-bit offsets");
  // max nnz in a C row (the reference's set_max_result_nnz, impl_symbolic.hpp:1501-1505)
  unsigned long long* d_mx = nullptr; unsigned long long h_mx = 0;
  KK_HIP(hipMalloc((void**)&d_mx, sizeof(unsigned long long)));
  KK_HIP(hipMemsetAsync(d_mx, 0, sizeof(unsigned long long), st));
  const int64_t nbk = kk::ceil_div(m, kk::kBlock);
  if (offset_type == KKAMD_I64) { KK_LAUNCH((kk::max_diff_kernel<int64_t>), (unsigned)(nbk < 4096 ? nbk : 4096), kk::kBlock, 0, st, m, (const int64_t*)d_row_mapC, d_mx); }
  else { KK_LAUNCH((kk::max_diff_kernel<int32_t>), (unsigned)(nbk < 4096 ? nbk : 4096), kk::kBlock, 0, st, m, (const int32_t*)d_row_mapC, d_mx); }
  KK_HIP(hipMemcpyAsync(&h_mx, d_mx, sizeof h_mx, hipMemcpyDeviceToHost, st));
  KK_HIP(hipStreamSynchronize(st));
  KK_HIP(hipFree(d_mx));
  h->max_row_nnz = (int64_t)h_mx;
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

int kkamd_spgemm_get(kkamd_spgemm_handle_t* h, int what, int64_t* value) {
  if (!h || !value) return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_get: null pointer");
  switch (what) {
    case 0: *value = h->c_nnz; break;
    case 1: *value = h->mults; break;
    case 2: *value = h->max_row_flops; break;
    case 3: *value = h->max_row_nnz; break;
    case 4: *value = h->symbolic_called; break;
    case 5: *value = h->numeric_called; break;
    default: return kk::fail(KKAMD_ERR_INVALID_ARG, "kkamd_spgemm_get: unknown query %d", what);
  }
  return KKAMD_OK;
}

}  // extern "C"
