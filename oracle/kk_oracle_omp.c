/* kk_oracle_omp.c -- CPU BASELINE kernels (test/bench infrastructure, see kk_oracle.h).
 *
 * OpenMP restatements of the reference's host (Kokkos::OpenMP) SpMV path, used
 * only by bench.py's cpu_baseline leg ("kind": "port") and by tests that check
 * them against the Serial oracle.  The reference's own OpenMP build cannot be
 * produced here (no Kokkos core), hence a port.
 */
#include "kk_oracle.h"
#include <omp.h>

int kko_omp_max_threads(void) { return omp_get_max_threads(); }
int kko_omp_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }

/* parallel first-touch copy: what Kokkos::View allocation + deep_copy do on the OpenMP backend (pages end up
 * spread over the NUMA nodes of the threads that touch them), instead of one thread faulting everything in. */
int kko_first_touch_copy(void* dst, const void* src, int64_t bytes) {
  char* d = (char*)dst; const char* s = (const char*)src;
  const int64_t chunk = 1 << 20;
  const int64_t n = (bytes + chunk - 1) / chunk;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < n; ++c) {
    const int64_t o = c * chunk, l = (o + chunk <= bytes) ? chunk : bytes - o;
    for (int64_t i = 0; i < l; ++i) d[o + i] = s[o + i];
  }
  return 0;
}

/* Kokkos::RangePolicy's automatic chunk size (Kokkos core, un-vendored; restated
 * from its published RangePolicy::set_auto_chunk_size: grow a power of two until
 * chunk*100*concurrency covers the range, falling back to a 40x rule capped at
 * 128 for small ranges).                                                        */
static int64_t kko_auto_chunk(int64_t n, int64_t conc) {
  int64_t c = 1;
  while (c * 100 * conc < n) c *= 2;
  if (c < 128) {
    c = 1;
    while ((c * 40 * conc < n) && (c < 128)) c *= 2;
  }
  return c;
}

/* SPMV_Functor::operator()(iRow): sparse/impl/KokkosSparse_spmv_impl.hpp:110-132
 * (sum over the row, sum *= alpha, y = sum or beta*y + sum) launched as
 * RangePolicy<Dynamic> when nnz > 1e7 else RangePolicy<Static>: :323-333.       */
#define KKO_DEFINE_SPMV_OMP(NAME, OT)                                                                           \
  int NAME(int64_t nrows, const OT* row_map, const int32_t* entries, const double* values, double alpha,        \
           const double* x, double beta, double* y) {                                                           \
    const int64_t nnz   = nrows > 0 ? (int64_t)row_map[nrows] : 0;                                              \
    const int dobeta    = (beta != 0.0);                                                                        \
    const int64_t chunk = kko_auto_chunk(nrows, omp_get_max_threads());                                         \
    if (nnz > 10000000) {                                                                                       \
      _Pragma("omp parallel for schedule(dynamic, chunk)") for (int64_t i = 0; i < nrows; ++i) {                \
        double sum = 0.0;                                                                                       \
        for (int64_t j = row_map[i]; j < (int64_t)row_map[i + 1]; ++j) sum += values[j] * x[entries[j]];        \
        sum *= alpha;                                                                                           \
        y[i] = dobeta ? beta * y[i] + sum : sum;                                                                \
      }                                                                                                         \
    } else {                                                                                                    \
      _Pragma("omp parallel for schedule(static)") for (int64_t i = 0; i < nrows; ++i) {                        \
        double sum = 0.0;                                                                                       \
        for (int64_t j = row_map[i]; j < (int64_t)row_map[i + 1]; ++j) sum += values[j] * x[entries[j]];        \
        sum *= alpha;                                                                                           \
        y[i] = dobeta ? beta * y[i] + sum : sum;                                                                \
      }                                                                                                         \
    }                                                                                                           \
    return 0;                                                                                                   \
  }
KKO_DEFINE_SPMV_OMP(kko_spmv_omp, int64_t)
KKO_DEFINE_SPMV_OMP(kko_spmv_omp_i32, int32_t)

/* SPMV_MV_LayoutLeft_Functor::operator()(iRow) on a host space: strips of 16
 * columns (sparse/impl/KokkosSparse_spmv_impl.hpp:883-890 with strip_mine<16>
 * :745-792), remainder columns in one narrower strip; RangePolicy schedule rule
 * as for rank-1 (spmv_alpha_beta_mv_no_transpose, host version :1008-1055).     */
int kko_spmv_mv_omp_i32(int64_t nrows, int64_t nvec, const int32_t* row_map, const int32_t* entries,
                        const double* values, double alpha, const double* X, int64_t xs0, int64_t xs1, double beta,
                        double* Y, int64_t ys0, int64_t ys1) {
  const int64_t chunk = kko_auto_chunk(nrows, omp_get_max_threads());
#pragma omp parallel for schedule(dynamic, chunk)
  for (int64_t i = 0; i < nrows; ++i) {
    for (int64_t kk = 0; kk < nvec; kk += 16) {
      const int u = (int)((nvec - kk) < 16 ? (nvec - kk) : 16);
      double sum[16];
      for (int k = 0; k < u; ++k) sum[k] = 0.0;
      for (int64_t j = row_map[i]; j < (int64_t)row_map[i + 1]; ++j) {
        const double av  = (alpha == 1.0) ? values[j] : alpha * values[j];
        const double* xr = X + (int64_t)entries[j] * xs0 + kk * xs1;
        for (int k = 0; k < u; ++k) sum[k] += av * xr[k * xs1];
      }
      double* yr = Y + i * ys0 + kk * ys1;
      for (int k = 0; k < u; ++k) yr[k * ys1] = (beta == 0.0) ? sum[k] : beta * yr[k * ys1] + sum[k];
    }
  }
  return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * SPGEMM_KK on the host (Kokkos::OpenMP): the KKMEM kernels with a per-thread linear-probing hash accumulator,
 * sparse/impl/KokkosSparse_spgemm_impl_kkmem.hpp:196-272 (MultiCoreTag4: hash = (column * HASHSCALAR) & (size - 1),
 * HASHSCALAR = 107 (:17); first-touch order kept in used_indices, table reset through that list), selected on the host
 * when k is large (:1259-1300; k < 250001 would take the dense accumulator of impl_speed.hpp instead), and the symbolic
 * counterpart without values (impl_symbolic.hpp:527-700, uncompressed).  Rows are handed out dynamically
 * (Kokkos::Schedule<Kokkos::Dynamic> team policy, :1440-1467) in chunks of team_work_size = 16
 * (KokkosKernels_Handle.hpp:323).  The library sorts C afterwards (numeric_spec.hpp:138-140): callers apply kko_sort_crs.
 * phase 0: symbolic -- fills row_mapC (m + 1, exclusive scan done here) and returns nnz(C);
 * phase 1: numeric  -- fills entriesC / valuesC in first-touch order.                                                  */
#include <stdlib.h>
#include <string.h>
static int64_t kko_next_pow2(int64_t v) { int64_t p = 1; while (p < v) p <<= 1; return p; }

int64_t kko_spgemm_kkmem_omp(int phase, int32_t m, int32_t n, int32_t k, const int64_t* row_mapA, const int32_t* entriesA,
                             const double* valuesA, const int64_t* row_mapB, const int32_t* entriesB, const double* valuesB,
                             int64_t* row_mapC, int32_t* entriesC, double* valuesC) {
  (void)n;
  /* table size: power of two >= 2 * the largest row it must hold (numeric: max nnz of a C row; symbolic: max row flops, <= k) */
  int64_t max_need = 1;
  if (phase == 0) {
#pragma omp parallel for schedule(static) reduction(max : max_need)
    for (int32_t i = 0; i < m; ++i) {
      int64_t f = 0;
      for (int64_t a = row_mapA[i]; a < row_mapA[i + 1]; ++a) { const int32_t r = entriesA[a]; f += row_mapB[r + 1] - row_mapB[r]; }
      if (f > max_need) max_need = f;
    }
    if (max_need > k) max_need = k;
  } else {
#pragma omp parallel for schedule(static) reduction(max : max_need)
    for (int32_t i = 0; i < m; ++i) { const int64_t l = row_mapC[i + 1] - row_mapC[i]; if (l > max_need) max_need = l; }
  }
  const int64_t hsize = kko_next_pow2(2 * max_need), hmask = hsize - 1;
  int fail = 0;
#pragma omp parallel
  {
    int32_t* hash_ids   = (int32_t*)malloc(sizeof(int32_t) * (size_t)hsize);
    double* hash_values = phase ? (double*)malloc(sizeof(double) * (size_t)hsize) : NULL;
    int32_t* used       = (int32_t*)malloc(sizeof(int32_t) * (size_t)hsize);
    if (!hash_ids || !used || (phase && !hash_values)) {
#pragma omp atomic write
      fail = 1;
    } else {
      for (int64_t i = 0; i < hsize; ++i) hash_ids[i] = -1;
#pragma omp for schedule(dynamic, 16)
      for (int32_t row = 0; row < m; ++row) {
        int32_t used_count = 0;
        for (int64_t a = row_mapA[row]; a < row_mapA[row + 1]; ++a) {
          const int32_t rowB = entriesA[a];
          const double valA  = phase ? valuesA[a] : 0.0;
          for (int64_t j = row_mapB[rowB]; j < row_mapB[rowB + 1]; ++j) {
            const int32_t c = entriesB[j];
            int64_t hash    = ((int64_t)c * 107) & hmask;
            while (1) {
              if (hash_ids[hash] == -1) {
                used[used_count++] = (int32_t)hash; hash_ids[hash] = c;
                if (phase) hash_values[hash] = valuesB[j] * valA;
                break;
              } else if (hash_ids[hash] == c) {
                if (phase) hash_values[hash] += valuesB[j] * valA;
                break;
              }
              hash = (hash + 1) & hmask;
            }
          }
        }
        if (phase) {
          int64_t pos = row_mapC[row];
          for (int32_t i = 0; i < used_count; ++i) { entriesC[pos] = hash_ids[used[i]]; valuesC[pos++] = hash_values[used[i]]; hash_ids[used[i]] = -1; }
        } else {
          row_mapC[row + 1] = used_count;
          for (int32_t i = 0; i < used_count; ++i) hash_ids[used[i]] = -1;
        }
      }
    }
    free(hash_ids); free(hash_values); free(used);
  }
  if (fail) return -1;
  if (phase == 0) {
    row_mapC[0] = 0;
    for (int32_t i = 0; i < m; ++i) row_mapC[i + 1] += row_mapC[i];      /* kk_exclusive_parallel_prefix_sum */
  }
  return row_mapC[m];
}

/* sort_crs_matrix on the host cores: rows in parallel, like the reference (sparse/src/KokkosSparse_SortCrs.hpp:43-120 sorts every
 * row inside one parallel_for) -- the baseline's SpGEMM ends with this pass (impl/KokkosSparse_spgemm_numeric_spec.hpp:138-140). */
#include "kk_oracle.h"
int kko_sort_crs_omp(int64_t nrows, const int64_t* row_map, int32_t* entries, double* values) {
  int64_t maxlen = 0;
#pragma omp parallel for schedule(static) reduction(max : maxlen)
  for (int64_t i = 0; i < nrows; ++i) { const int64_t l = row_map[i + 1] - row_map[i]; if (l > maxlen) maxlen = l; }
  int fail = 0;
#pragma omp parallel
  {
    int32_t* te = (int32_t*)malloc(sizeof(int32_t) * (size_t)(maxlen + 1));
    double* tv  = (double*)malloc(sizeof(double) * (size_t)(maxlen + 1));
    if (!te || !tv) {
#pragma omp atomic write
      fail = 1;
    } else {
#pragma omp for schedule(dynamic, 16)
      for (int64_t i = 0; i < nrows; ++i)
        kko_sort_row(row_map[i + 1] - row_map[i], entries + row_map[i], values ? values + row_map[i] : 0, te, tv);
    }
    free(te); free(tv);
  }
  return fail ? -1 : 0;
}
