// KokkosSparse::CrsMatrix -- three-array CSR as the hot path sees it
// (reference: sparse/src/KokkosSparse_CrsMatrix.hpp:317-388,526-591: graph.row_map, graph.entries,
// values, numCols_; signed ordinal required, :320).  Constructors mirrored: the seven-argument
// (label, nrows, ncols, nnz, values, row_map, entries) form the tests use, (label, ncols, values, graph),
// (label, graph, ncols) and the converting copy used to canonicalise to const / Unmanaged.
#pragma once
#include "KokkosSparse_StaticCrsGraph.hpp"

namespace KokkosSparse {

template <class ScalarType, class OrdinalType, class Device, class MemoryTraits = void,
          class SizeType = KokkosKernels::default_size_type>
class CrsMatrix {
  static_assert(std::is_signed<std::remove_const_t<OrdinalType>>::value, "CrsMatrix requires that OrdinalType is a signed integer type.");
  using traits_ = std::conditional_t<std::is_void<MemoryTraits>::value, Kokkos::MemoryTraits<0>, MemoryTraits>;
 public:
  using value_type             = ScalarType;
  using ordinal_type           = OrdinalType;
  using size_type              = SizeType;
  using non_const_value_type   = std::remove_const_t<ScalarType>;
  using non_const_ordinal_type = std::remove_const_t<OrdinalType>;
  using non_const_size_type    = std::remove_const_t<SizeType>;
  using const_value_type       = std::add_const_t<ScalarType>;
  using const_ordinal_type     = std::add_const_t<OrdinalType>;
  using const_size_type        = std::add_const_t<SizeType>;
  using device_type            = typename Kokkos::Impl::space_of<Device>::type;
  using execution_space        = typename device_type::execution_space;
  using memory_space           = typename device_type::memory_space;
  using memory_traits          = MemoryTraits;
  using StaticCrsGraphType     = StaticCrsGraph<OrdinalType, Kokkos::LayoutLeft, device_type, MemoryTraits, non_const_size_type>;
  using staticcrsgraph_type    = StaticCrsGraphType;
  using index_type             = typename StaticCrsGraphType::entries_type;
  using row_map_type           = typename StaticCrsGraphType::row_map_type;
  using values_type            = Kokkos::View<ScalarType*, Kokkos::LayoutLeft, device_type, traits_>;
  using HostMirror             = CrsMatrix<ScalarType, OrdinalType, Kokkos::Device<Kokkos::Serial, Kokkos::HostSpace>, MemoryTraits, SizeType>;

  StaticCrsGraphType graph;
  values_type values;

  CrsMatrix() : numCols_(0) {}
  template <class V, class R, class E>
  CrsMatrix(const std::string&, OrdinalType nrows, OrdinalType ncols, size_t annz, const V& vals, const R& rowmap, const E& cols)
      : graph(cols, rowmap), values(vals), numCols_(ncols) {
    if ((size_t)(nrows ? nrows + 1 : 0) != rowmap.extent(0) && !(nrows == 0 && rowmap.extent(0) <= 1))
      KokkosKernels::Impl::throw_runtime_exception("CrsMatrix: row_map has the wrong extent for the number of rows");
    if (annz != cols.extent(0)) KokkosKernels::Impl::throw_runtime_exception("CrsMatrix: nnz does not match entries.extent(0)");
    numRows_ = nrows;
  }
  template <class V> CrsMatrix(const std::string&, OrdinalType ncols, const V& vals, const StaticCrsGraphType& g)
      : graph(g), values(vals), numCols_(ncols) { numRows_ = (OrdinalType)g.numRows(); }
  CrsMatrix(const std::string& label, const StaticCrsGraphType& g, OrdinalType ncols)
      : graph(g), values(label, g.entries.extent(0)), numCols_(ncols) { numRows_ = (OrdinalType)g.numRows(); }
  template <class S2, class O2, class D2, class M2, class Z2>
  CrsMatrix(const CrsMatrix<S2, O2, D2, M2, Z2>& o) : graph(o.graph), values(o.values), numCols_(o.numCols()), numRows_(o.numRows()) {}
  template <class S2, class O2, class D2, class M2, class Z2>
  CrsMatrix(const std::string&, const CrsMatrix<S2, O2, D2, M2, Z2>& o) : CrsMatrix(o) {}

  non_const_ordinal_type numRows() const { return numRows_; }
  non_const_ordinal_type numCols() const { return numCols_; }
  non_const_size_type nnz() const { return (non_const_size_type)graph.entries.extent(0); }

 private:
  non_const_ordinal_type numCols_ = 0, numRows_ = 0;
};

template <class T> struct is_crs_matrix : std::false_type {};
template <class... P> struct is_crs_matrix<CrsMatrix<P...>> : std::true_type {};
template <class... P> struct is_crs_matrix<const CrsMatrix<P...>> : std::true_type {};
template <class T> inline constexpr bool is_crs_matrix_v = is_crs_matrix<T>::value;

}  // namespace KokkosSparse
