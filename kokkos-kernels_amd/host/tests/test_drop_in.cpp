// Host-side test of the drop-in headers (compiled with hipcc, run on the GPU by tests/test_gpu_host_api.py).
// The cases mirror the reference's own unit tests for this path:
//   test_github_issue_101            sparse/unit_test/Test_Sparse_spmv.hpp:823-961 (exact known answer)
//   test_spmv_all_interfaces_light   :964-1055 (space/handle/neither x rank-1/rank-2 on 111 x 99)
//   spgemm view/matrix/no-reuse APIs sparse/unit_test/Test_Sparse_spgemm.hpp:243-252,444-481
//   matrix files: the 6x6 fixtures of sparse/unit_test/Test_Sparse_IOUtils.hpp:39-54 written as general / symmetric /
//   hermitian / skew-symmetric .mtx and read back (:129-163), plus .bin / .crs / .mtx round trips
//   spmv_struct (2-D 5-pt, every overload) sparse/unit_test/Test_Sparse_spmv.hpp:263-296,652-711
// Expected values come from the test's own sequential loops, as in the reference's tests.
#include <cmath>
#include <cstdio>
#include <limits>
#include <random>
#include "KokkosSparse_spmv.hpp"
#include "KokkosSparse_dist_spmv.hpp"
#include "KokkosSparse_spgemm.hpp"
#include "KokkosSparse_IOUtils.hpp"
#include "KokkosSparse_SortCrs.hpp"

using device = Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>;
static int failures = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

template <class scalar, class size_type>
KokkosSparse::CrsMatrix<scalar, int, device, void, size_type> make_random(int m, int n, int per_row, std::vector<size_type>& rm,
                                                                          std::vector<int>& ent, std::vector<scalar>& val, unsigned seed) {
  std::mt19937 g(seed); rm.assign(m + 1, 0); ent.clear(); val.clear();
  for (int i = 0; i < m; ++i) {
    const int len = n ? (int)(g() % (2 * per_row + 1)) : 0;
    std::vector<int> cols;
    for (int j = 0; j < len; ++j) cols.push_back((int)(g() % n));
    std::sort(cols.begin(), cols.end()); cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
    for (int c : cols) { ent.push_back(c); val.push_back((scalar)(1 + (g() % 4900) / 100.0)); }
    rm[i + 1] = (size_type)ent.size();
  }
  using M = KokkosSparse::CrsMatrix<scalar, int, device, void, size_type>;
  typename M::row_map_type::non_const_type d_rm("rm", m + 1);
  typename M::index_type d_ent("ent", ent.size());
  typename M::values_type d_val("val", val.size());
  Kokkos::deep_copy(d_rm, Kokkos::View<size_type*, Kokkos::HostSpace>(rm.data(), rm.size()));
  if (!ent.empty()) {
    Kokkos::deep_copy(d_ent, Kokkos::View<int*, Kokkos::HostSpace>(ent.data(), ent.size()));
    Kokkos::deep_copy(d_val, Kokkos::View<scalar*, Kokkos::HostSpace>(val.data(), val.size()));
  }
  return M("A", m, n, ent.size(), d_val, d_rm, d_ent);
}

void test_github_issue_101() {
  using graph_type = KokkosSparse::CrsMatrix<double, int, device>::StaticCrsGraphType;
  const float EPS_f = std::numeric_limits<float>::epsilon();
  const double expected = 1.0 + (double)EPS_f / 2.0;
  Kokkos::View<int*, device> colInds("colInds", 2);
  Kokkos::View<int*, device> rowOffsets("rowOffsets", 2);
  int h_c[2] = {0, 1}, h_r[2] = {0, 2};
  Kokkos::deep_copy(colInds, Kokkos::View<int*, Kokkos::HostSpace>(h_c, 2));
  Kokkos::deep_copy(rowOffsets, Kokkos::View<int*, Kokkos::HostSpace>(h_r, 2));
  graph_type G(colInds, rowOffsets);
  Kokkos::View<double*, device> x("x", 2), y("y", 1);
  Kokkos::deep_copy(x, 1.0);
  auto y_h = Kokkos::create_mirror_view(y);
  {
    KokkosSparse::CrsMatrix<double, int, device> A_d("A_d", G, 2);
    double h_v[2] = {1.0, (double)EPS_f / 2.0};
    Kokkos::deep_copy(A_d.values, Kokkos::View<double*, Kokkos::HostSpace>(h_v, 2));
    Kokkos::deep_copy(y, 0.0);
    KokkosSparse::spmv("N", 1.0, A_d, x, 0.0, y);
    Kokkos::deep_copy(y_h, y);
    EXPECT(y_h(0) == expected);
    for (int nv = 1; nv <= 22; ++nv) {
      Kokkos::View<double**, Kokkos::LayoutLeft, device> X("X", 2, nv), Y("Y", 1, nv);
      Kokkos::deep_copy(X, 1.0);
      KokkosSparse::spmv("N", 1.0, A_d, X, 0.0, Y);
      auto Y_h = Kokkos::create_mirror_view(Y); Kokkos::deep_copy(Y_h, Y);
      for (int j = 0; j < nv; ++j) EXPECT(Y_h(0, j) == expected);
    }
  }
  {  // float matrix, double vectors
    KokkosSparse::CrsMatrix<float, int, device> A_f("A_f", KokkosSparse::CrsMatrix<float, int, device>::StaticCrsGraphType(colInds, rowOffsets), 2);
    float h_v[2] = {1.0f, EPS_f / 2.0f};
    Kokkos::deep_copy(A_f.values, Kokkos::View<float*, Kokkos::HostSpace>(h_v, 2));
    Kokkos::deep_copy(y, 0.0);
    KokkosSparse::spmv("N", 1.0, A_f, x, 0.0, y);
    Kokkos::deep_copy(y_h, y);
    EXPECT(y_h(0) == expected);
  }
}

template <class layout>
void test_all_interfaces() {
  using M  = KokkosSparse::CrsMatrix<double, int, device, void, int>;
  using V1 = Kokkos::View<double*, device>;
  using V2 = Kokkos::View<double**, layout, device>;
  std::vector<int> rm, ent; std::vector<double> val;
  const int m = 111, n = 99, nv = 7;
  M A = make_random<double, int>(m, n, 10, rm, ent, val, 11);
  std::vector<double> hx(n * nv), hy(m * nv, 0.0);
  std::mt19937 g(3);
  for (auto& v : hx) v = (g() % 1000) / 1000.0;
  auto idx = [&](int i, int j, int rows) { return std::is_same<layout, Kokkos::LayoutLeft>::value ? i + (size_t)j * rows : (size_t)i * nv + j; };
  for (int i = 0; i < m; ++i) for (int j = rm[i]; j < rm[i + 1]; ++j) for (int c = 0; c < nv; ++c)
    hy[idx(i, c, m)] += 2.0 * val[j] * hx[idx(ent[j], c, n)];
  V2 X("X", n, nv), Y("Y", m, nv);
  Kokkos::deep_copy(X, Kokkos::View<double**, layout, Kokkos::HostSpace>(hx.data(), n, nv));
  V1 x1("x1", n), y1("y1", m);
  std::vector<double> hx1(n), hy1(m, 0.0);
  for (int i = 0; i < n; ++i) hx1[i] = hx[idx(i, 0, n)];
  for (int i = 0; i < m; ++i) for (int j = rm[i]; j < rm[i + 1]; ++j) hy1[i] += 2.0 * val[j] * hx1[ent[j]];
  Kokkos::deep_copy(x1, Kokkos::View<double*, Kokkos::HostSpace>(hx1.data(), n));
  auto check1 = [&]() { auto h = Kokkos::create_mirror_view(y1); Kokkos::deep_copy(h, y1); double e = 0; for (int i = 0; i < m; ++i) e = std::max(e, std::fabs(h(i) - hy1[i])); EXPECT(e < 1e-12); Kokkos::deep_copy(y1, -7.0); };
  auto check2 = [&]() { auto h = Kokkos::create_mirror_view(Y); Kokkos::deep_copy(h, Y); double e = 0; for (int i = 0; i < m; ++i) for (int c = 0; c < nv; ++c) e = std::max(e, std::fabs(h(i, c) - hy[idx(i, c, m)])); EXPECT(e < 1e-12); Kokkos::deep_copy(Y, -7.0); };
  Kokkos::HIP space;
  KokkosSparse::SPMVHandle<device, M, V1, V1> h1(KokkosSparse::SPMV_DEFAULT);
  KokkosSparse::SPMVHandle<device, M, V2, V2> h2(KokkosSparse::SPMV_MERGE_PATH);
  KokkosSparse::spmv(space, &h1, "N", 2.0, A, x1, 0.0, y1); space.fence(); check1();
  KokkosSparse::spmv(&h1, "N", 2.0, A, x1, 0.0, y1); Kokkos::fence(); check1();
  KokkosSparse::spmv(space, "N", 2.0, A, x1, 0.0, y1); space.fence(); check1();
  KokkosSparse::spmv("N", 2.0, A, x1, 0.0, y1); Kokkos::fence(); check1();
  KokkosSparse::spmv(space, &h2, "N", 2.0, A, X, 0.0, Y); space.fence(); check2();
  KokkosSparse::spmv(&h2, "N", 2.0, A, X, 0.0, Y); Kokkos::fence(); check2();
  KokkosSparse::spmv(space, "N", 2.0, A, X, 0.0, Y); space.fence(); check2();
  KokkosSparse::spmv("N", 2.0, A, X, 0.0, Y); Kokkos::fence(); check2();
  // a handle whose FIRST call is mode T gets its plan, and with it the cached transpose; the reference's
  // public expert members are assignable (sparse/src/KokkosSparse_spmv_handle.hpp:243-252)
  {
    KokkosSparse::SPMVHandle<device, M, V1, V1> ht(KokkosSparse::SPMV_DEFAULT);
    ht.team_size = 64; ht.vector_length = 8; ht.rows_per_thread = 4; ht.force_static_schedule = true; ht.force_dynamic_schedule = false;
    ht.set_knob("explicit_transpose_min_knnz", 0);
    V1 xt("xt", m), yt("yt", n);
    std::vector<double> hxt(m), hyt(n, 0.0);
    for (int i = 0; i < m; ++i) hxt[i] = (g() % 1000) / 1000.0;
    for (int i = 0; i < m; ++i) for (int j = rm[i]; j < rm[i + 1]; ++j) hyt[ent[j]] += 2.0 * val[j] * hxt[i];
    Kokkos::deep_copy(xt, Kokkos::View<double*, Kokkos::HostSpace>(hxt.data(), m));
    for (int rep = 0; rep < 2; ++rep) {
      Kokkos::deep_copy(yt, -7.0);
      KokkosSparse::spmv(space, &ht, "T", 2.0, A, xt, 0.0, yt); space.fence();
      auto h = Kokkos::create_mirror_view(yt); Kokkos::deep_copy(h, yt);
      double e = 0; for (int i = 0; i < n; ++i) e = std::max(e, std::fabs(h(i) - hyt[i]));
      EXPECT(e < 1e-12);
    }
    EXPECT(ht.plan != nullptr);
    int64_t cached = 0;
    if (ht.plan) KokkosSparse::Impl::kkamd_check(kkamd_spmv_plan_query(ht.plan, "transpose_cached", &cached));
    EXPECT(cached == 1);
    // A.values rewritten in place under the live handle (the reference reads them at every call): the cached transpose follows --
    // exactly by default, and by notification (handle.values_changed()) when the caller asks for "values_tracking" 1
    auto values_host = Kokkos::create_mirror_view(A.values);
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) ht.set_knob("values_tracking", 1);
      for (int i = 0; i < (int)val.size(); ++i) { val[i] = (pass ? -1.0 : 3.0) * val[i] + 0.125; values_host(i) = val[i]; }
      Kokkos::deep_copy(A.values, values_host);
      if (pass == 1) ht.values_changed();
      std::fill(hyt.begin(), hyt.end(), 0.0);
      for (int i = 0; i < m; ++i) for (int j = rm[i]; j < rm[i + 1]; ++j) hyt[ent[j]] += 2.0 * val[j] * hxt[i];
      Kokkos::deep_copy(yt, -7.0);
      KokkosSparse::spmv(space, &ht, "T", 2.0, A, xt, 0.0, yt); space.fence();
      auto h = Kokkos::create_mirror_view(yt); Kokkos::deep_copy(h, yt);
      double e = 0; for (int i = 0; i < n; ++i) e = std::max(e, std::fabs(h(i) - hyt[i]));
      EXPECT(e < 1e-11);
    }
    // rank 2, mode T, through the same kind of handle (the mode-N dispatch on the cached transpose)
    KokkosSparse::SPMVHandle<device, M, V2, V2> ht2(KokkosSparse::SPMV_DEFAULT);
    ht2.set_knob("explicit_transpose_min_knnz", 0);
    V2 Xt("Xt", m, nv), Yt("Yt", n, nv);
    std::vector<double> hXt((size_t)m * nv), hYt((size_t)n * nv, 0.0);
    for (auto& v : hXt) v = (g() % 1000) / 1000.0;
    for (int i = 0; i < m; ++i) for (int j = rm[i]; j < rm[i + 1]; ++j) for (int c = 0; c < nv; ++c) hYt[idx(ent[j], c, n)] += 2.0 * val[j] * hXt[idx(i, c, m)];
    Kokkos::deep_copy(Xt, Kokkos::View<double**, layout, Kokkos::HostSpace>(hXt.data(), m, nv));
    Kokkos::deep_copy(Yt, -7.0);
    KokkosSparse::spmv(space, &ht2, "T", 2.0, A, Xt, 0.0, Yt); space.fence();
    auto h2t = Kokkos::create_mirror_view(Yt); Kokkos::deep_copy(h2t, Yt);
    double e2 = 0; for (int i = 0; i < n; ++i) for (int c = 0; c < nv; ++c) e2 = std::max(e2, std::fabs(h2t(i, c) - hYt[idx(i, c, n)]));
    EXPECT(e2 < 1e-11);
    if (ht2.plan) KokkosSparse::Impl::kkamd_check(kkamd_spmv_plan_query(ht2.plan, "transpose_cached", &cached));
    EXPECT(cached == 1);
  }
  // error behaviour: dimension mismatch and BSR-only algorithm on a CrsMatrix
  bool threw = false;
  try { V1 bad("bad", n + 1); KokkosSparse::spmv("N", 1.0, A, bad, 0.0, y1); } catch (const std::runtime_error& e) { threw = std::string(e.what()).find("Dimensions do not match") != std::string::npos; }
  EXPECT(threw);
  threw = false;
  try { KokkosSparse::SPMVHandle<device, M, V1, V1> hb(KokkosSparse::SPMV_BSR_TC); } catch (const std::invalid_argument&) { threw = true; }
  EXPECT(threw);
  threw = false;
  try { KokkosSparse::spmv("X", 1.0, A, x1, 0.0, y1); } catch (const std::runtime_error&) { threw = true; }
  EXPECT(threw);
}

template <class size_type>
void test_spgemm() {
  using M  = KokkosSparse::CrsMatrix<double, int, device, void, size_type>;
  using KH = KokkosKernels::Experimental::KokkosKernelsHandle<size_type, int, double, Kokkos::HIP, Kokkos::HIPSpace, Kokkos::HIPSpace>;
  std::vector<size_type> rmA, rmB; std::vector<int> eA, eB; std::vector<double> vA, vB;
  const int m = 300, n = 250, k = 200;
  M A = make_random<double, size_type>(m, n, 8, rmA, eA, vA, 5), B = make_random<double, size_type>(n, k, 6, rmB, eB, vB, 6);
  // host Gustavson with a dense accumulator, rows emitted sorted
  std::vector<size_type> rmC(m + 1, 0); std::vector<int> eC; std::vector<double> vC;
  std::vector<double> acc(k, 0.0); std::vector<char> flag(k, 0);
  for (int i = 0; i < m; ++i) {
    std::vector<int> cols;
    for (size_type a = rmA[i]; a < rmA[i + 1]; ++a) for (size_type b = rmB[eA[a]]; b < rmB[eA[a] + 1]; ++b) {
      if (!flag[eB[b]]) { flag[eB[b]] = 1; cols.push_back(eB[b]); }
      acc[eB[b]] += vB[b] * vA[a];
    }
    std::sort(cols.begin(), cols.end());
    for (int c : cols) { eC.push_back(c); vC.push_back(acc[c]); acc[c] = 0; flag[c] = 0; }
    rmC[i + 1] = (size_type)eC.size();
  }
  auto compare = [&](const M& C) {
    EXPECT((size_t)C.nnz() == eC.size());
    auto h_rm = Kokkos::create_mirror_view(C.graph.row_map); Kokkos::deep_copy(h_rm, C.graph.row_map);
    auto h_e  = Kokkos::create_mirror_view(C.graph.entries); Kokkos::deep_copy(h_e, C.graph.entries);
    auto h_v  = Kokkos::create_mirror_view(C.values);        Kokkos::deep_copy(h_v, C.values);
    bool ok = true;
    for (int i = 0; i <= m; ++i) ok = ok && (h_rm(i) == rmC[i]);
    for (size_t j = 0; j < eC.size() && ok; ++j) ok = ok && (h_e(j) == eC[j]) && (std::fabs(h_v(j) - vC[j]) / (std::fabs(h_v(j)) + std::fabs(vC[j])) < 1e-7);
    EXPECT(ok);
  };
  KH kh;
  bool threw = false;
  M C;
  try { KokkosSparse::spgemm_symbolic(kh, A, false, B, false, C); } catch (const std::invalid_argument&) { threw = true; }
  EXPECT(threw);                                   // no SpGEMM sub-handle yet
  kh.create_spgemm_handle(KokkosSparse::SPGEMM_KK);
  KokkosSparse::spgemm_symbolic(kh, A, false, B, false, C);
  EXPECT((size_t)kh.get_spgemm_handle()->get_c_nnz() == eC.size());
  EXPECT(kh.get_spgemm_handle()->is_symbolic_called() && !kh.get_spgemm_handle()->is_numeric_called());
  KokkosSparse::spgemm_numeric(kh, A, false, B, false, C);
  compare(C);
  KokkosSparse::spgemm_numeric(kh, A, false, B, false, C);   // numeric reuse
  compare(C);
  kh.destroy_spgemm_handle();
  M C2 = KokkosSparse::spgemm<M>(A, false, B, false);        // no-reuse interface
  compare(C2);
  threw = false;
  try { KokkosSparse::spgemm<M>(A, false, A, false); } catch (const std::invalid_argument&) { threw = true; }
  EXPECT(threw);
  // numeric before symbolic on a fresh handle
  KH kh2; kh2.create_spgemm_handle();
  threw = false;
  try { KokkosSparse::spgemm_numeric(kh2, A, false, B, false, C); } catch (const std::invalid_argument&) { threw = true; }
  EXPECT(threw);
  // handle options act or throw (sparse/src/KokkosSparse_spgemm_handle.hpp:427-501, KokkosKernels_Handle.hpp:380-465)
  KH kh3; kh3.create_spgemm_handle(KokkosSparse::SPGEMM_KK_DENSE);          // dense-accumulator numeric
  kh3.get_spgemm_handle()->set_compression(true); kh3.get_spgemm_handle()->set_compression_cut_off(1.0);
  M C3;
  KokkosSparse::spgemm_symbolic(kh3, A, false, B, false, C3);
  EXPECT(kh3.get_spgemm_handle()->is_compressed());          // B is sorted here, the cut-off of 1.0 keeps whatever compression gives
  KokkosSparse::spgemm_numeric(kh3, A, false, B, false, C3);
  compare(C3);
  kh3.get_spgemm_handle()->set_accumulator_type(KokkosSparse::SPGEMM_ACC_SPARSE);
  kh3.get_spgemm_handle()->set_compression(false);
  KokkosSparse::spgemm_numeric(kh3, A, false, B, false, C3);
  compare(C3);
  // The reference driver's own call sequence, unchanged (perf_test/sparse/KokkosSparse_spgemm.cpp:310-317,345-352,377-399) with its
  // default parameters (perf_test/sparse/KokkosSparse_spgemm.cpp parameters: chunk 16, shmem 16128, team / vector -1): hints are
  // accepted and remembered, SPGEMM_SERIAL (its check_output path) gives the same C
  {
    const int chunk_size = 16, shmemsize = 16128, team_size = -1, vector_size = -1, use_dynamic_scheduling = 1, verbose = 0;
    KH drv;
    drv.set_team_work_size(chunk_size);
    drv.set_shmem_size(shmemsize);
    drv.set_suggested_team_size(team_size);
    drv.set_suggested_vector_size(vector_size);
    if (use_dynamic_scheduling) drv.set_dynamic_scheduling(true);
    if (verbose) drv.set_verbose(true);
    EXPECT(drv.get_set_team_work_size() == 16 && drv.get_shmem_size() == 16128 && drv.is_dynamic_scheduling());
    EXPECT(drv.get_team_work_size(256, 0, 0) == 16);
    KH seq;
    seq.set_team_work_size(chunk_size); seq.set_shmem_size(shmemsize); seq.set_suggested_team_size(team_size);
    seq.create_spgemm_handle(KokkosSparse::SPGEMM_SERIAL);
    if (use_dynamic_scheduling) seq.set_dynamic_scheduling(true);
    M Cref;
    KokkosSparse::spgemm_symbolic(seq, A, false, B, false, Cref);
    KokkosSparse::spgemm_numeric(seq, A, false, B, false, Cref);
    compare(Cref);
    for (int algorithm : {(int)KokkosSparse::SPGEMM_KK, (int)KokkosSparse::SPGEMM_KK_MEMORY, (int)KokkosSparse::SPGEMM_DEBUG}) {
      drv.create_spgemm_handle(KokkosSparse::SPGEMMAlgorithm(algorithm));
      drv.get_spgemm_handle()->mkl_keep_output = true;
      drv.get_spgemm_handle()->set_mkl_sort_option(7);
      drv.get_spgemm_handle()->mkl_convert_to_1base = true;
      drv.get_spgemm_handle()->MaxColDenseAcc = 250000;
      drv.get_spgemm_handle()->set_read_write_cost_calc(false);
      drv.get_spgemm_handle()->set_compression_steps(true);
      drv.get_spgemm_handle()->set_min_hash_size_scale(1);
      drv.get_spgemm_handle()->set_first_level_hash_cut_off(0.5);
      drv.get_spgemm_handle()->set_compression_cut_off(0.85);
      EXPECT(drv.get_spgemm_handle()->get_min_hash_size_scale() == 1 && drv.get_spgemm_handle()->get_first_level_hash_cut_off() == 0.5);
      EXPECT(drv.get_spgemm_handle()->get_algorithm_type() == KokkosSparse::SPGEMMAlgorithm(algorithm));
      // view-level calls with the driver's own allocation order (:393-414)
      typedef typename M::values_type::non_const_type scalar_view_t;
      typedef typename M::row_map_type::non_const_type lno_view_t;
      typedef typename M::index_type::non_const_type lno_nnz_view_t;
      lno_view_t row_mapC("non_const_lnow_row", m + 1);
      lno_nnz_view_t entriesC("entriesC (empty)", 0);
      scalar_view_t valuesC("valuesC (empty)", 0);
      KokkosSparse::spgemm_symbolic(&drv, m, n, k, A.graph.row_map, A.graph.entries, false, B.graph.row_map, B.graph.entries, false, row_mapC);
      Kokkos::HIP().fence();
      size_type c_nnz_size = drv.get_spgemm_handle()->get_c_nnz();
      EXPECT((size_t)c_nnz_size == (size_t)Cref.nnz());
      if (c_nnz_size) {
        entriesC = lno_nnz_view_t(Kokkos::view_alloc(Kokkos::WithoutInitializing, "entriesC"), c_nnz_size);
        valuesC  = scalar_view_t(Kokkos::view_alloc(Kokkos::WithoutInitializing, "valuesC"), c_nnz_size);
      }
      KokkosSparse::spgemm_numeric(&drv, m, n, k, A.graph.row_map, A.graph.entries, A.values, false, B.graph.row_map, B.graph.entries, B.values,
                                   false, row_mapC, entriesC, valuesC);
      Kokkos::HIP().fence();
      M Cd("CrsMatrixC", m, k, valuesC.extent(0), valuesC, row_mapC, entriesC);
      compare(Cd);
    }
    // the unit test's sequence (sparse/unit_test/Test_Sparse_spgemm.hpp:90-99)
    KH ut; ut.set_team_work_size(16); ut.set_dynamic_scheduling(true);
    ut.create_spgemm_handle(KokkosSparse::SPGEMM_KK);
    M Cu;
    KokkosSparse::spgemm_symbolic(ut, A, false, B, false, Cu);
    KokkosSparse::spgemm_numeric(ut, A, false, B, false, Cu);
    compare(Cu);
  }
  // an unknown option still fails loudly
  auto throws = [&](auto&& fn) { bool t = false; try { fn(); } catch (const std::runtime_error&) { t = true; } return t; };
  EXPECT(throws([&] { kh3.get_spgemm_handle()->set("no_such_option", 1.0); }));
  kh3.get_spgemm_handle()->set_sort_option(0);
  kh3.get_spgemm_handle()->set_sort_option(1);
}

// 5-pt stencil on an ni x nj grid: interior rows hold -1 -1 4 -1 -1 in ascending column order, boundary rows the
// identity (like the reference generator with every BC = 1); expected values from a sequential CRS loop.
void test_spmv_struct() {
  using M = KokkosSparse::CrsMatrix<double, int, device, void, int>;
  const int ni = 37, nj = 11, n = ni * nj;
  std::vector<int> rm(n + 1, 0), ent; std::vector<double> val;
  for (int j = 0; j < nj; ++j)
    for (int i = 0; i < ni; ++i) {
      const int r = j * ni + i;
      if (i == 0 || j == 0 || i == ni - 1 || j == nj - 1) { ent.push_back(r); val.push_back(1.0); }
      else {
        const int c[5] = {r - ni, r - 1, r, r + 1, r + ni};
        const double v[5] = {-1.0, -1.5, 4.25, -0.5, -1.25};
        for (int q = 0; q < 5; ++q) { ent.push_back(c[q]); val.push_back(v[q]); }
      }
      rm[r + 1] = (int)ent.size();
    }
  typename M::row_map_type::non_const_type d_rm("rm", n + 1);
  typename M::index_type d_ent("ent", ent.size());
  typename M::values_type d_val("val", val.size());
  Kokkos::deep_copy(d_rm, Kokkos::View<int*, Kokkos::HostSpace>(rm.data(), rm.size()));
  Kokkos::deep_copy(d_ent, Kokkos::View<int*, Kokkos::HostSpace>(ent.data(), ent.size()));
  Kokkos::deep_copy(d_val, Kokkos::View<double*, Kokkos::HostSpace>(val.data(), val.size()));
  M A("A", n, n, ent.size(), d_val, d_rm, d_ent);
  Kokkos::View<int*, Kokkos::HostSpace> structure("structure", 2);
  structure(0) = ni; structure(1) = nj;
  std::mt19937 g(13718);
  std::vector<double> hx(n), hy(n), expect(n);
  for (int i = 0; i < n; ++i) { hx[i] = (g() % 1000) / 1000.0; hy[i] = (g() % 1000) / 1000.0; }
  const double alpha = 1.5, beta = -0.5;
  for (int r = 0; r < n; ++r) {
    double sum = 0;
    for (int q = rm[r]; q < rm[r + 1]; ++q) sum += val[q] * hx[ent[q]];
    expect[r] = beta * hy[r] + alpha * sum;
  }
  auto check = [&](const double* got) {
    bool ok = true;
    for (int r = 0; r < n; ++r) ok = ok && std::fabs(got[r] - expect[r]) <= 1e-13 * 20;
    EXPECT(ok);
  };
  Kokkos::View<double*, device> x("x", n), y("y", n);
  auto y_h = Kokkos::create_mirror_view(y);
  Kokkos::deep_copy(x, Kokkos::View<double*, Kokkos::HostSpace>(hx.data(), n));
  auto reset = [&]() { Kokkos::deep_copy(y, Kokkos::View<double*, Kokkos::HostSpace>(hy.data(), n)); };
  reset(); KokkosSparse::Experimental::spmv_struct("N", 1, structure, alpha, A, x, beta, y);
  Kokkos::deep_copy(y_h, y); check(y_h.data());
  reset(); KokkosSparse::Experimental::spmv_struct(Kokkos::HIP(), "N", 1, structure, alpha, A, x, beta, y);
  Kokkos::deep_copy(y_h, y); check(y_h.data());
  reset(); KokkosSparse::Experimental::spmv_struct("N", 1, structure, alpha, A, x, beta, y, KokkosSparse::RANK_ONE());
  Kokkos::deep_copy(y_h, y); check(y_h.data());
  // rank-2 with a single column takes the structured path as well (:803-826)
  Kokkos::View<double**, Kokkos::LayoutLeft, device> X("X", n, 1), Y("Y", n, 1);
  auto Y_h = Kokkos::create_mirror_view(Y);
  Kokkos::deep_copy(Kokkos::subview(X, Kokkos::ALL(), 0), Kokkos::View<double*, Kokkos::HostSpace>(hx.data(), n));
  Kokkos::deep_copy(Kokkos::subview(Y, Kokkos::ALL(), 0), Kokkos::View<double*, Kokkos::HostSpace>(hy.data(), n));
  KokkosSparse::Experimental::spmv_struct("N", 1, structure, alpha, A, X, beta, Y);
  Kokkos::deep_copy(Y_h, Y); check(Y_h.data());
  // dimension check
  bool threw = false;
  Kokkos::View<double*, device> xs("xs", n - 1);
  try { KokkosSparse::Experimental::spmv_struct("N", 1, structure, alpha, A, xs, beta, y); } catch (const std::runtime_error&) { threw = true; }
  EXPECT(threw);
}

void test_ioutils() {
  using M = KokkosSparse::CrsMatrix<double, int, device, void, int>;
  const double sym[6][6]  = {{11, 12, 13, 14, 15, 16}, {12, 2, 0, 0, 0, 0}, {13, 0, 0, 0, 0, 0}, {14, 0, 0, 4, 0, 0}, {15, 0, 0, 0, 5, 0}, {16, 0, 0, 0, 0, 6}};
  const double asym[6][6] = {{1, 0, 0, 9, 0, 0}, {0, 2, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 8}, {0, 0, 0, 4, 0, 0}, {0, 7, 0, 0, 5, 0}, {0, 0, 0, 0, 0, 6}};
  auto check = [&](const char* kind, const double (*D)[6], bool lower_only, double upper_sign) {
    const std::string file = std::string("kkamd_test_io_") + kind + ".mtx";
    {
      std::ofstream out(file);
      size_t n = 0;
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) if (D[r][c] != 0 && (!lower_only || c <= r)) ++n;
      out << "%%MatrixMarket matrix coordinate real " << kind << "\n% written by test_drop_in.cpp\n6 6 " << n << "\n";
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) if (D[r][c] != 0 && (!lower_only || c <= r)) out << r + 1 << " " << c + 1 << " " << D[r][c] << "\n";
    }
    M A = KokkosSparse::Impl::read_kokkos_crst_matrix<M>(file.c_str());
    auto rm = Kokkos::create_mirror_view(A.graph.row_map); Kokkos::deep_copy(rm, A.graph.row_map);
    auto en = Kokkos::create_mirror_view(A.graph.entries); Kokkos::deep_copy(en, A.graph.entries);
    auto va = Kokkos::create_mirror_view(A.values);        Kokkos::deep_copy(va, A.values);
    bool ok = A.numRows() == 6 && A.numCols() == 6;
    int pos = 0;
    for (int r = 0; r < 6 && ok; ++r) {
      ok = ok && rm(r) == pos;
      for (int c = 0; c < 6; ++c) {
        const double e = (c > r && lower_only) ? upper_sign * D[c][r] : D[r][c];
        if (e != 0) { ok = ok && pos < (int)A.nnz() && en(pos) == c && va(pos) == e; ++pos; }
      }
    }
    EXPECT(ok && rm(6) == pos);
    // every format round-trips
    for (const char* ext : {".mtx", ".bin", ".crs"}) {
      const std::string f2 = std::string("kkamd_test_io_rt") + ext;
      KokkosSparse::Impl::write_kokkos_crst_matrix(A, f2.c_str());
      M B = KokkosSparse::Impl::read_kokkos_crst_matrix<M>(f2.c_str());
      auto rm2 = Kokkos::create_mirror_view(B.graph.row_map); Kokkos::deep_copy(rm2, B.graph.row_map);
      auto en2 = Kokkos::create_mirror_view(B.graph.entries); Kokkos::deep_copy(en2, B.graph.entries);
      auto va2 = Kokkos::create_mirror_view(B.values);        Kokkos::deep_copy(va2, B.values);
      bool same = B.numRows() == 6 && B.numCols() == 6 && B.nnz() == A.nnz();
      for (int i = 0; i <= 6 && same; ++i) same = rm2(i) == rm(i);
      for (size_t j = 0; j < (size_t)A.nnz() && same; ++j) same = en2(j) == en(j) && va2(j) == va(j);
      EXPECT(same);
      std::remove(f2.c_str());
    }
    std::remove(file.c_str());
  };
  check("general", asym, false, 1.0);
  check("symmetric", sym, true, 1.0);
  check("hermitian", sym, true, 1.0);
  check("skew-symmetric", sym, true, -1.0);
  bool threw = false;
  try { KokkosSparse::Impl::read_kokkos_crst_matrix<M>("no_such_file.mtx"); } catch (const std::runtime_error&) { threw = true; }
  EXPECT(threw);
}

// sort_crs_matrix / sort_and_merge_matrix / transpose_matrix on an unsorted matrix with duplicate columns
// (sparse/unit_test/Test_Sparse_SortCrs.hpp style: compare with a host std::sort / merge of the same rows)
void test_sort_merge_transpose() {
  using M = KokkosSparse::CrsMatrix<double, int, device, void, int>;
  const int m = 120, n = 90;
  std::mt19937 g(77);
  std::vector<int> rm(m + 1, 0), ent; std::vector<double> val;
  for (int i = 0; i < m; ++i) {
    const int len = (int)(g() % 25);
    for (int j = 0; j < len; ++j) { ent.push_back((int)(g() % n)); val.push_back((double)(1 + g() % 9)); }
    rm[i + 1] = (int)ent.size();
  }
  auto upload = [&]() {
    typename M::row_map_type::non_const_type d_rm("rm", m + 1);
    typename M::index_type d_ent("ent", ent.size());
    typename M::values_type d_val("val", val.size());
    Kokkos::deep_copy(d_rm, Kokkos::View<int*, Kokkos::HostSpace>(rm.data(), rm.size()));
    Kokkos::deep_copy(d_ent, Kokkos::View<int*, Kokkos::HostSpace>(ent.data(), ent.size()));
    Kokkos::deep_copy(d_val, Kokkos::View<double*, Kokkos::HostSpace>(val.data(), val.size()));
    return M("A", m, n, ent.size(), d_val, d_rm, d_ent);
  };
  // expected: stable sort per row, then merged
  std::vector<int> s_ent(ent), mrm(m + 1, 0), ment; std::vector<double> s_val(val), mval;
  for (int i = 0; i < m; ++i) {
    std::vector<std::pair<int, int>> key;
    for (int j = rm[i]; j < rm[i + 1]; ++j) key.emplace_back(ent[j], j);
    std::stable_sort(key.begin(), key.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    for (size_t q = 0; q < key.size(); ++q) { s_ent[rm[i] + q] = key[q].first; s_val[rm[i] + q] = val[key[q].second]; }
    for (int j = rm[i]; j < rm[i + 1]; ++j) {
      if (j > rm[i] && s_ent[j] == s_ent[j - 1]) mval.back() += s_val[j];
      else { ment.push_back(s_ent[j]); mval.push_back(s_val[j]); }
    }
    mrm[i + 1] = (int)ment.size();
  }
  M A = upload();
  KokkosSparse::sort_crs_matrix(A);
  {
    auto e = Kokkos::create_mirror_view(A.graph.entries); Kokkos::deep_copy(e, A.graph.entries);
    auto v = Kokkos::create_mirror_view(A.values);        Kokkos::deep_copy(v, A.values);
    bool ok = true;
    for (size_t j = 0; j < ent.size(); ++j) ok = ok && e(j) == s_ent[j] && v(j) == s_val[j];
    EXPECT(ok);
  }
  M B = upload();
  M C = KokkosSparse::sort_and_merge_matrix(B);
  {
    auto r = Kokkos::create_mirror_view(C.graph.row_map); Kokkos::deep_copy(r, C.graph.row_map);
    auto e = Kokkos::create_mirror_view(C.graph.entries); Kokkos::deep_copy(e, C.graph.entries);
    auto v = Kokkos::create_mirror_view(C.values);        Kokkos::deep_copy(v, C.values);
    bool ok = (size_t)C.nnz() == ment.size();
    for (int i = 0; i <= m && ok; ++i) ok = r(i) == mrm[i];
    for (size_t j = 0; j < ment.size() && ok; ++j) ok = e(j) == ment[j] && v(j) == mval[j];
    EXPECT(ok);
  }
  M T = KokkosSparse::Impl::transpose_matrix(C);
  M TT = KokkosSparse::Impl::transpose_matrix(T);        // (C^T)^T == C for a sorted, merged matrix
  {
    auto r = Kokkos::create_mirror_view(TT.graph.row_map); Kokkos::deep_copy(r, TT.graph.row_map);
    auto e = Kokkos::create_mirror_view(TT.graph.entries); Kokkos::deep_copy(e, TT.graph.entries);
    auto v = Kokkos::create_mirror_view(TT.values);        Kokkos::deep_copy(v, TT.values);
    bool ok = T.numRows() == n && T.numCols() == m && (size_t)TT.nnz() == ment.size();
    for (int i = 0; i <= m && ok; ++i) ok = r(i) == mrm[i];
    for (size_t j = 0; j < ment.size() && ok; ++j) ok = e(j) == ment[j] && v(j) == mval[j];
    EXPECT(ok);
  }
}

// the multi-GPU operator with one rank (no communication): same y as the plain call, x kept in the operator's window
void test_dist_single_rank() {
  using M  = KokkosSparse::CrsMatrix<double, int, device, void, int>;
  using V1 = Kokkos::View<double*, device>;
  std::vector<int> rm, ent; std::vector<double> val;
  const int n = 5000;
  M A = make_random<double, int>(n, n, 9, rm, ent, val, 21);
  V1 x("x", n), y("y", n), y2("y2", n);
  std::vector<double> hx(n); std::mt19937 g(4); for (auto& v : hx) v = (g() % 1000) / 500.0 - 1.0;
  Kokkos::deep_copy(x, Kokkos::View<double*, Kokkos::HostSpace>(hx.data(), n));
  Kokkos::HIP space;
  KokkosSparse::Experimental::DistributedSpMV<M> op(space, A, {0, (int64_t)n}, 0, nullptr);
  EXPECT(op.query("exchange") == 0 && op.query("parts") == 1);
  op.apply(space, 2.0, x, 0.0, y); space.fence();
  KokkosSparse::spmv("N", 2.0, A, x, 0.0, y2); Kokkos::fence();
  auto h1 = Kokkos::create_mirror_view(y); Kokkos::deep_copy(h1, y);
  auto h2 = Kokkos::create_mirror_view(y2); Kokkos::deep_copy(h2, y2);
  bool ok = true;
  for (int i = 0; i < n; ++i) ok = ok && std::fabs(h1(i) - h2(i)) <= 1e-12;
  EXPECT(ok);
  V1 xl(op.x_local(), n);                                  // unmanaged view of the operator's own x window
  Kokkos::deep_copy(xl, x);
  op.apply(space, 2.0, xl, 0.0, y); space.fence();
  Kokkos::deep_copy(h1, y);
  for (int i = 0; i < n; ++i) ok = ok && std::fabs(h1(i) - h2(i)) <= 1e-12;
  EXPECT(ok);
}

int main() {
  Kokkos::initialize();
  test_dist_single_rank();
  test_sort_merge_transpose();
  test_ioutils();
  test_spmv_struct();
  test_github_issue_101();
  test_all_interfaces<Kokkos::LayoutLeft>();
  test_all_interfaces<Kokkos::LayoutRight>();
  test_spgemm<int>();
  test_spgemm<size_t>();
  Kokkos::finalize();
  std::printf(failures ? "drop-in API tests: %d FAILED\n" : "drop-in API tests: all passed\n", failures);
  return failures ? 1 : 0;
}
