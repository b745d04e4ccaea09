// TEST INFRASTRUCTURE -- stand-in for common/src/Kokkos_ArithTraits.hpp (see Kokkos_Core.hpp in this directory)
#pragma once
// the reference's own file of this name is found first by quote-includes from common/src: its include guard is taken here so
// that it is skipped (it needs Kokkos_NumericTraits / MathematicalFunctions / Complex / QuadPrecisionMath of the real core)
#define KOKKOSKERNELS_KOKKOS_ARITHTRAITS_HPP
#include <Kokkos_Core.hpp>
namespace Kokkos {
template <class T> struct ArithTraits {
  using val_type = T; using mag_type = T; using magnitudeType = T;
  static constexpr bool is_specialized = true, is_signed = std::is_signed<T>::value, is_integer = std::is_integral<T>::value, is_exact = is_integer, is_complex = false;
  static constexpr bool has_infinity = !is_integer;
  static std::string name() { return "scalar"; }
  static T zero() { return T(0); }
  static T one() { return T(1); }
  static T min() { return std::numeric_limits<T>::lowest(); }
  static T max() { return std::numeric_limits<T>::max(); }
  static T infinity() { return std::numeric_limits<T>::infinity(); }
  static T nan() { return std::numeric_limits<T>::quiet_NaN(); }
  static T epsilon() { return std::numeric_limits<T>::epsilon(); }
  static mag_type abs(const T& x) { return x < T(0) ? -x : x; }
  static T conj(const T& x) { return x; }
  static T real(const T& x) { return x; }
  static T imag(const T&) { return T(0); }
  static T sqrt(const T& x) { return x; }
  static T pow(const T& x, const T&) { return x; }
  static bool isNan(const T&) { return false; }
  static bool isInf(const T&) { return false; }
};
template <> inline std::string ArithTraits<double>::name() { return "double"; }
template <> inline std::string ArithTraits<float>::name() { return "float"; }
namespace Details { template <class T> using ArithTraits = ::Kokkos::ArithTraits<T>; }
}  // namespace Kokkos
