#pragma once
#include <Kokkos_Core.hpp>
