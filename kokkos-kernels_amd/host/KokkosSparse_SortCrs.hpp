// KokkosSparse::sort_crs_matrix / sort_crs_graph / sort_and_merge_matrix and KokkosSparse::Impl::transpose_matrix with the
// reference's names and argument order (sparse/src/KokkosSparse_SortCrs.hpp:43-120,210-300,304-400;
// sparse/src/KokkosSparse_Utils.hpp:338-400), over the C ABI (kkamd_sort_crs, kkamd_sort_and_merge, kkamd_transpose).
// The SortAlgorithm argument of the reference is accepted and ignored: there is one device algorithm here (stable,
// LDS bitonic segments + merge-path passes, kk_util.hip).
#pragma once
#include "KokkosSparse_CrsMatrix.hpp"
#include "kkamd_status.hpp"

namespace KokkosSparse {

enum class SortAlgorithm { DEFAULT, RADIX, SHELL, BULK_SORT };

template <class execution_space, class rowmap_t, class entries_t, class values_t>
void sort_crs_matrix(const execution_space& exec, const rowmap_t& rowmap, const entries_t& entries, const values_t& values,
                     typename entries_t::non_const_value_type /*numCols*/ = 0, SortAlgorithm /*option*/ = SortAlgorithm::DEFAULT) {
  static_assert(!std::is_const<typename entries_t::value_type>::value, "sort_crs_matrix: entries_t must not be const-valued");
  static_assert(!std::is_const<typename values_t::value_type>::value, "sort_crs_matrix: value_t must not be const-valued");
  const int64_t nrows = rowmap.extent(0) ? (int64_t)rowmap.extent(0) - 1 : 0;
  if (nrows == 0 || entries.extent(0) == 0) return;
  Impl::kkamd_check(kkamd_sort_crs(nrows, rowmap.data(), (int32_t*)entries.data(), (void*)values.data(),
                                   Impl::kkamd_offset<typename rowmap_t::non_const_value_type>::value,
                                   Impl::kkamd_scalar<typename values_t::non_const_value_type>::value,
                                   reinterpret_cast<kkamd_stream_t>(exec.hip_stream())));
}
template <class crsMat_t>
void sort_crs_matrix(const typename crsMat_t::execution_space& exec, const crsMat_t& A, SortAlgorithm option = SortAlgorithm::DEFAULT) {
  sort_crs_matrix(exec, A.graph.row_map, A.graph.entries, A.values, A.numCols(), option);
}
template <class crsMat_t>
void sort_crs_matrix(const crsMat_t& A, SortAlgorithm option = SortAlgorithm::DEFAULT) {
  sort_crs_matrix(typename crsMat_t::execution_space(), A, option);
}

template <class execution_space, class rowmap_t, class entries_t>
void sort_crs_graph(const execution_space& exec, const rowmap_t& rowmap, const entries_t& entries,
                    typename entries_t::non_const_value_type /*numCols*/ = 0, SortAlgorithm /*option*/ = SortAlgorithm::DEFAULT) {
  const int64_t nrows = rowmap.extent(0) ? (int64_t)rowmap.extent(0) - 1 : 0;
  if (nrows == 0 || entries.extent(0) == 0) return;
  Impl::kkamd_check(kkamd_sort_crs(nrows, rowmap.data(), (int32_t*)entries.data(), nullptr,
                                   Impl::kkamd_offset<typename rowmap_t::non_const_value_type>::value, KKAMD_F64,
                                   reinterpret_cast<kkamd_stream_t>(exec.hip_stream())));
}
template <class crsGraph_t>
void sort_crs_graph(const crsGraph_t& G, SortAlgorithm option = SortAlgorithm::DEFAULT) {
  sort_crs_graph(typename crsGraph_t::execution_space(), G.row_map, G.entries, 0, option);
}

// sorts the input views (like the reference) and returns the merged arrays; when nothing merges the inputs are returned
template <class execution_space, class rowmap_t, class entries_t, class values_t>
void sort_and_merge_matrix(const execution_space& exec, const typename rowmap_t::const_type& rowmap_in, const entries_t& entries_in,
                           const values_t& values_in, rowmap_t& rowmap_out, entries_t& entries_out, values_t& values_out,
                           typename entries_t::non_const_value_type /*numCols*/ = 0, SortAlgorithm /*option*/ = SortAlgorithm::DEFAULT) {
  using nc_rowmap_t = typename rowmap_t::non_const_type;
  const int64_t nrows = rowmap_in.extent(0) ? (int64_t)rowmap_in.extent(0) - 1 : 0;
  if (nrows == 0) {
    rowmap_out = nc_rowmap_t("SortedMerged rowmap", rowmap_in.extent(0)); entries_out = entries_t(); values_out = values_t();
    return;
  }
  constexpr int ot = Impl::kkamd_offset<typename rowmap_t::non_const_value_type>::value;
  constexpr int vt = Impl::kkamd_scalar<typename values_t::non_const_value_type>::value;
  kkamd_stream_t st = reinterpret_cast<kkamd_stream_t>(exec.hip_stream());
  nc_rowmap_t rm_out(Kokkos::view_alloc(Kokkos::WithoutInitializing, "SortedMerged rowmap"), (size_t)nrows + 1);
  int64_t nnz_out = 0;
  Impl::kkamd_check(kkamd_sort_and_merge(nrows, rowmap_in.data(), (int32_t*)entries_in.data(), (void*)values_in.data(), ot, vt,
                                         (void*)rm_out.data(), nullptr, nullptr, &nnz_out, st));
  if ((size_t)nnz_out == (size_t)entries_in.extent(0)) {        // nothing to merge (:343-352)
    Kokkos::deep_copy(rm_out, rowmap_in);
    rowmap_out = rm_out; entries_out = entries_in; values_out = values_in;
    return;
  }
  entries_t e_out(Kokkos::view_alloc(Kokkos::WithoutInitializing, "SortedMerged entries"), (size_t)nnz_out);
  values_t v_out(Kokkos::view_alloc(Kokkos::WithoutInitializing, "SortedMerged values"), (size_t)nnz_out);
  Impl::kkamd_check(kkamd_sort_and_merge(nrows, rowmap_in.data(), (int32_t*)entries_in.data(), (void*)values_in.data(), ot, vt,
                                         (void*)rm_out.data(), (int32_t*)e_out.data(), (void*)v_out.data(), &nnz_out, st));
  rowmap_out = rm_out; entries_out = e_out; values_out = v_out;
}
template <class crsMat_t>
crsMat_t sort_and_merge_matrix(const typename crsMat_t::execution_space& exec, const crsMat_t& A, SortAlgorithm option = SortAlgorithm::DEFAULT) {
  using rowmap_t  = typename crsMat_t::row_map_type::non_const_type;
  using entries_t = typename crsMat_t::index_type::non_const_type;
  using values_t  = typename crsMat_t::values_type::non_const_type;
  rowmap_t rowmap_out; entries_t entries_out; values_t values_out;
  sort_and_merge_matrix<typename crsMat_t::execution_space, rowmap_t, entries_t, values_t>(exec, A.graph.row_map, A.graph.entries, A.values, rowmap_out,
                                                                                          entries_out, values_out, A.numCols(), option);
  return crsMat_t("SortedMerged", A.numRows(), A.numCols(), values_out.extent(0), values_out, rowmap_out, entries_out);
}
template <class crsMat_t>
crsMat_t sort_and_merge_matrix(const crsMat_t& A, SortAlgorithm option = SortAlgorithm::DEFAULT) {
  return sort_and_merge_matrix(typename crsMat_t::execution_space(), A, option);
}

namespace Impl {
// rows of the result are column-sorted (the reference leaves their order to its atomics)
template <class crsMat_t>
crsMat_t transpose_matrix(const crsMat_t& A) {
  using rowmap_t  = typename crsMat_t::row_map_type::non_const_type;
  using entries_t = typename crsMat_t::index_type::non_const_type;
  using values_t  = typename crsMat_t::values_type::non_const_type;
  rowmap_t t_rm("Transpose rowmap", (size_t)A.numCols() + 1);
  entries_t t_ent(Kokkos::view_alloc(Kokkos::WithoutInitializing, "Transpose entries"), A.nnz());
  values_t t_val(Kokkos::view_alloc(Kokkos::WithoutInitializing, "Transpose values"), A.nnz());
  kkamd_check(kkamd_transpose(A.numRows(), A.numCols(), (int64_t)A.nnz(), A.graph.row_map.data(), (const int32_t*)A.graph.entries.data(),
                              A.values.data(), kkamd_offset<typename rowmap_t::value_type>::value,
                              kkamd_scalar<typename values_t::value_type>::value, (void*)t_rm.data(), (int32_t*)t_ent.data(),
                              (void*)t_val.data(), nullptr));
  return crsMat_t("Transpose", A.numCols(), A.numRows(), A.nnz(), t_val, t_rm, t_ent);
}
}  // namespace Impl
}  // namespace KokkosSparse
