// status -> exception mapping of the C ABI (include/kkamd.h): the reference reports every failure on this
// path as a C++ exception -- std::runtime_error for dimension / vendor-status problems
// (common/src/KokkosKernels_Error.hpp:26, sparse/src/KokkosSparse_Utils_rocsparse.hpp:34-83),
// std::invalid_argument for handle misuse (sparse/src/KokkosSparse_spgemm_symbolic.hpp:148-160).
#pragma once
#include <stdexcept>
#include <string>
#include "../../include/kkamd.h"

namespace KokkosSparse { namespace Impl {
inline void kkamd_check(int status) {
  if (status == KKAMD_OK) return;
  const std::string msg = kkamd_last_error();
  if (status == KKAMD_ERR_STATE) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}
template <class T> struct kkamd_scalar;
template <> struct kkamd_scalar<double> { static constexpr int value = KKAMD_F64; };
template <> struct kkamd_scalar<float> { static constexpr int value = KKAMD_F32; };
template <class T> struct kkamd_offset;
template <> struct kkamd_offset<int> { static constexpr int value = KKAMD_I32; };
template <> struct kkamd_offset<long> { static constexpr int value = KKAMD_I64; };
template <> struct kkamd_offset<long long> { static constexpr int value = KKAMD_I64; };
template <> struct kkamd_offset<unsigned long> { static constexpr int value = KKAMD_I64; };   // size_t (values < 2^63)
template <> struct kkamd_offset<unsigned long long> { static constexpr int value = KKAMD_I64; };
}}  // namespace KokkosSparse::Impl
