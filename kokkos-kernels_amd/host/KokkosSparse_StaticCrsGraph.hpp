// KokkosSparse::StaticCrsGraph -- the CSR graph container of the drop-in surface
// (reference: sparse/src/KokkosSparse_StaticCrsGraph.hpp:254-353: `row_map` of numRows+1 offsets and
// `entries` of nnz ordinals; this header models exactly those two members and the typedefs the SpMV /
// SpGEMM entry points use).
#pragma once
#include "Kokkos_Shim.hpp"

namespace KokkosSparse {

template <class OrdinalType, class LayoutOrDevice, class DeviceOrVoid = void, class MemoryTraits = void,
          class SizeType = KokkosKernels::default_size_type>
class StaticCrsGraph {
  using dev_ = std::conditional_t<std::is_void<DeviceOrVoid>::value, LayoutOrDevice, DeviceOrVoid>;
  using traits_ = std::conditional_t<std::is_void<MemoryTraits>::value, Kokkos::MemoryTraits<0>, MemoryTraits>;
 public:
  using data_type       = OrdinalType;
  using size_type       = SizeType;
  using device_type     = typename Kokkos::Impl::space_of<dev_>::type;
  using execution_space = typename device_type::execution_space;
  using memory_space    = typename device_type::memory_space;
  using array_layout    = Kokkos::LayoutLeft;
  using entries_type    = Kokkos::View<OrdinalType*, array_layout, device_type, traits_>;
  using row_map_type    = Kokkos::View<const SizeType*, array_layout, device_type, traits_>;
  using row_block_type  = Kokkos::View<SizeType*, array_layout, device_type, traits_>;

  entries_type entries;
  row_map_type row_map;
  row_block_type row_block_offsets;   // optional in the reference; unused here

  StaticCrsGraph() = default;
  template <class E, class R> StaticCrsGraph(const E& entries_, const R& row_map_) : entries(entries_), row_map(row_map_) {}
  template <class O2, class L2, class D2, class M2, class S2>
  StaticCrsGraph(const StaticCrsGraph<O2, L2, D2, M2, S2>& o) : entries(o.entries), row_map(o.row_map) {}

  size_type numRows() const { return row_map.extent(0) ? (size_type)(row_map.extent(0) - 1) : (size_type)0; }
};

}  // namespace KokkosSparse
