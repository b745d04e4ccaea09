// TEST INFRASTRUCTURE -- see Kokkos_Core.hpp in this directory
#pragma once
#include <Kokkos_Core.hpp>
