// TEST INFRASTRUCTURE -- stand-in for Kokkos containers' Kokkos_StaticCrsGraph.hpp (see Kokkos_Core.hpp in this directory)
#pragma once
#include <Kokkos_Core.hpp>
namespace Kokkos {
template <class DataType, class Arg1Type, class Arg2Type = void, class Arg3Type = void, class SizeType = size_t>
class StaticCrsGraph {};
template <class G> struct GraphRowViewConst {};
template <class G, class I> G create_staticcrsgraph(const std::string&, const I&) { return G(); }
}  // namespace Kokkos
