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
