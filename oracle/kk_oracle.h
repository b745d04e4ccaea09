/* kk_oracle.h -- CPU ORACLE for the KokkosSparse spmv / spgemm hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call it, and only as the checker / the CPU baseline.  The product path
 * (kokkos-kernels_amd/, libkkamd.so) never links or loads this code.
 *
 * Every function is a plain-C restatement of a specific reference function
 * (kokkos/kokkos-kernels v4.7.00); the file:line it follows is cited at each
 * definition in kk_oracle.c.  The reference itself cannot be built here
 * (Kokkos core is neither installed nor vendored), so the oracle is pinned by
 *   - the reference's own known-answer tests (test_github_issue_101, the
 *     matrixIssue402 fixture, the wiki SpMV example), and
 *   - golden matrices produced by interpreting the reference's generator
 *     source (tests/golden/make_structured_golden.py),
 * see tests/test_oracle_*.py.
 *
 * Conventions: row_map is int64_t (covers the reference's int and size_t
 * offset ETI combos), entries are int32_t (the reference's default ordinal),
 * scalars are double unless a function says otherwise.
 */
#ifndef KK_ORACLE_H
#define KK_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- SpMV -------------------------------------------------------------- */
/* Kokkos::Serial path. mode: 'N','C','T','H'.  Returns 0, or -1 on bad mode. */
int kko_spmv_serial(char mode, int64_t nrows, int64_t ncols, const int64_t* row_map, const int32_t* entries,
                    const double* values, double alpha, const double* x, double beta, double* y);
/* same, float matrix values with double vectors (test_github_issue_101 mixed case) */
int kko_spmv_serial_f32a(char mode, int64_t nrows, int64_t ncols, const int64_t* row_map, const int32_t* entries,
                         const float* values, double alpha, const double* x, double beta, double* y);
/* all-float variant */
int kko_spmv_serial_f32(char mode, int64_t nrows, int64_t ncols, const int64_t* row_map, const int32_t* entries,
                        const float* values, float alpha, const float* x, float beta, float* y);
/* the unit test's own oracle (sequential_spmv) */
int kko_spmv_sequential(char mode, int64_t nrows, int64_t ncols, const int64_t* row_map, const int32_t* entries,
                        const double* values, double alpha, const double* x, double beta, double* y);
/* KokkosSparse::Experimental::spmv_struct (host path): ndim 1..3, structure = {ni[,nj[,nk]]}, stencil_type 1 FD / 2 FE */
int kko_spmv_struct(char mode, int stencil_type, int ndim, const int64_t* structure, int64_t nrows, int64_t ncols,
                    const int64_t* row_map, const int32_t* entries, const double* values, double alpha,
                    const double* x, double beta, double* y);
/* rank-2: X is ncols x nvec, Y is nrows x nvec; element (i,k) at i*xs0 + k*xs1. */
int kko_spmv_mv_serial(char mode, int64_t nrows, int64_t ncols, int64_t nvec, const int64_t* row_map,
                       const int32_t* entries, const double* values, double alpha, const double* X, int64_t xs0,
                       int64_t xs1, double beta, double* Y, int64_t ys0, int64_t ys1);

/* ---- SpGEMM (SPGEMM_DEBUG / SPGEMM_SERIAL) ------------------------------ */
/* fills row_mapC[0..m]; returns nnz(C) (or -1 on allocation failure). */
int64_t kko_spgemm_symbolic(int32_t m, int32_t n, int32_t k, const int64_t* row_mapA, const int32_t* entriesA,
                            const int64_t* row_mapB, const int32_t* entriesB, int64_t* row_mapC);
/* fills entriesC/valuesC in first-touch order (as the reference does). */
int kko_spgemm_numeric(int32_t m, int32_t n, int32_t k, const int64_t* row_mapA, const int32_t* entriesA,
                       const double* valuesA, const int64_t* row_mapB, const int32_t* entriesB, const double* valuesB,
                       const int64_t* row_mapC, int32_t* entriesC, double* valuesC);
/* sort_crs_matrix: per-row ascending columns, values permuted alongside (stable). */
int kko_sort_crs(int64_t nrows, const int64_t* row_map, int32_t* entries, double* values);
/* sort_and_merge_matrix: sorts in place, merged row_map always, merged entries/values when out_entries != NULL */
int64_t kko_sort_and_merge(int64_t nrows, const int64_t* row_map, int32_t* entries, double* values, int64_t* out_row_map,
                           int32_t* out_entries, double* out_values);
/* sum over rows of sum over A(i,:) of nnz(B(k,:)) -- the reference's original_overall_flops / 2 */
int64_t kko_spgemm_mults(int32_t m, const int64_t* row_mapA, const int32_t* entriesA, const int64_t* row_mapB,
                         int64_t* max_row_flops);
/* explicit transpose (used by the issue-402 test to form A^T) */
int kko_transpose(int32_t nrows, int32_t ncols, const int64_t* row_map, const int32_t* entries, const double* values,
                  int64_t* t_row_map, int32_t* t_entries, double* t_values);

/* ---- generators ---------------------------------------------------------- */
/* stencil: 0 = FD (5-pt / 7-pt), 1 = FE (9-pt / 27-pt). */
int64_t kko_laplace2d_nnz(int stencil, int64_t nx, int64_t ny);
int64_t kko_laplace3d_nnz(int stencil, int64_t nx, int64_t ny, int64_t nz);
int kko_gen_laplace1d(int64_t nx, int leftBC, int rightBC, int64_t* row_map, int32_t* entries, double* values);
/* bc[4] = {left,right,bottom,top}; each 0 or 1. */
int kko_gen_laplace2d(int stencil, int64_t nx, int64_t ny, const int* bc, int64_t* row_map, int32_t* entries,
                      double* values);
/* all six BCs = 1 (the only 3-D case the hot-path configs use). */
int kko_gen_laplace3d(int stencil, int64_t nx, int64_t ny, int64_t nz, int64_t* row_map, int32_t* entries,
                      double* values);
/* R-MAT edge list (ours; the reference has no generator -- SURVEY F9). Writes nedges (row<<32|col) keys. */
int kko_gen_rmat_keys(int scale, int64_t nedges, uint64_t seed, uint64_t* keys);
double kko_hash_value_1_50(uint64_t key, uint64_t seed);

/* ---- CPU baselines (OpenMP restatements of the reference's host kernels) -- */
int kko_omp_max_threads(void);
int kko_omp_set_threads(int n);
int kko_first_touch_copy(void* dst, const void* src, int64_t bytes);
int kko_spmv_omp(int64_t nrows, const int64_t* row_map, const int32_t* entries, const double* values, double alpha,
                 const double* x, double beta, double* y);
int kko_spmv_omp_i32(int64_t nrows, const int32_t* row_map, const int32_t* entries, const double* values, double alpha,
                     const double* x, double beta, double* y);
int64_t kko_spgemm_kkmem_omp(int phase, int32_t m, int32_t n, int32_t k, const int64_t* row_mapA, const int32_t* entriesA,
                             const double* valuesA, const int64_t* row_mapB, const int32_t* entriesB, const double* valuesB,
                             int64_t* row_mapC, int32_t* entriesC, double* valuesC);
int kko_spmv_mv_omp_i32(int64_t nrows, int64_t nvec, const int32_t* row_map, const int32_t* entries,
                        const double* values, double alpha, const double* X, int64_t xs0, int64_t xs1, double beta,
                        double* Y, int64_t ys0, int64_t ys1);

#ifdef __cplusplus
}
#endif
/* merge sort of one row's (column, value) pairs; te / tv: scratch of the row's length (used by kko_sort_crs and its OpenMP form) */
void kko_sort_row(int64_t len, int32_t* e, double* v, int32_t* te, double* tv);
#endif
