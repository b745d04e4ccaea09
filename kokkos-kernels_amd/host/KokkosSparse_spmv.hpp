// KokkosSparse::spmv -- the public SpMV entry points of the drop-in surface, rank-1 and rank-2, CRS.
// Reference: sparse/src/KokkosSparse_spmv.hpp:75-375 (overload taking space + handle) and :400-474 (the three
// convenience overloads).  What is kept: template signature shape, runtime dimension checks and their
// message (:126-142), the alpha == 0 / empty-matrix shortcut semantics (:145-154, implemented inside the
// library), rank-2 with one contiguous column -> rank-1 (:203-217), handle-less calls = a throw-away
// SPMV_FAST_SETUP handle (docs/source/API/sparse/spmv.rst).  What replaces Impl::SPMV<...>::spmv: one call
// across the C ABI (include/kkamd.h) with the Views unwrapped to raw device pointers, exactly what the
// rocSPARSE specialisation does (sparse/tpls/KokkosSparse_spmv_tpl_spec_decl.hpp:303-338).
#pragma once
#include <sstream>
#include "KokkosSparse_CrsMatrix.hpp"
#include "KokkosSparse_spmv_handle.hpp"

namespace KokkosSparse {

namespace {
constexpr const char* NoTranspose        = "N";
constexpr const char* Transpose          = "T";
constexpr const char* Conjugate          = "C";
constexpr const char* ConjugateTranspose = "H";
}  // namespace

namespace Impl {
template <class AMatrix> kkamd_crs_t make_crs_desc(const AMatrix& A) {
  kkamd_crs_t d;
  d.num_rows = A.numRows(); d.num_cols = A.numCols(); d.nnz = (int64_t)A.nnz();
  d.d_row_map = A.graph.row_map.data(); d.d_entries = A.graph.entries.data(); d.d_values = A.values.data();
  d.offset_type = kkamd_offset<typename AMatrix::non_const_size_type>::value;
  d.value_type  = kkamd_scalar<typename AMatrix::non_const_value_type>::value;
  static_assert(std::is_same<typename AMatrix::non_const_ordinal_type, int>::value, "kkamd: ordinals must be int32");
  return d;
}
}  // namespace Impl

template <class ExecutionSpace, class Handle, class AlphaType, class AMatrix, class XVector, class BetaType, class YVector>
void spmv(const ExecutionSpace& space, Handle* handle, const char mode[], const AlphaType& alpha, const AMatrix& A,
          const XVector& x, const BetaType& beta, const YVector& y) {
  static_assert(is_crs_matrix_v<AMatrix>, "KokkosSparse::spmv: AMatrix must be a CrsMatrix");
  static_assert(Kokkos::is_view<XVector>::value, "KokkosSparse::spmv: XVector must be a Kokkos::View.");
  static_assert(Kokkos::is_view<YVector>::value, "KokkosSparse::spmv: YVector must be a Kokkos::View.");
  static_assert(XVector::rank() == YVector::rank(), "KokkosSparse::spmv: Vector ranks do not match.");
  static_assert(XVector::rank() == 1 || XVector::rank() == 2, "KokkosSparse::spmv: Both Vector inputs must have rank 1 or 2");
  static_assert(!std::is_const<typename YVector::value_type>::value, "KokkosSparse::spmv: Output Vector must be non-const.");
  static_assert(std::is_same<typename XVector::non_const_value_type, typename YVector::non_const_value_type>::value,
                "kkamd: x and y must have the same scalar type");

  const size_t m = A.numRows(), n = A.numCols();
  if ((mode[0] == NoTranspose[0]) || (mode[0] == Conjugate[0])) {
    if ((x.extent(1) != y.extent(1)) || (n != x.extent(0)) || (m != y.extent(0))) {
      std::ostringstream os;
      os << "KokkosSparse::spmv: Dimensions do not match: "
         << ", A: " << m << " x " << n << ", x: " << x.extent(0) << " x " << x.extent(1) << ", y: " << y.extent(0)
         << " x " << y.extent(1);
      KokkosKernels::Impl::throw_runtime_exception(os.str());
    }
  } else if ((mode[0] == Transpose[0]) || (mode[0] == ConjugateTranspose[0])) {
    if ((x.extent(1) != y.extent(1)) || (m != x.extent(0)) || (n != y.extent(0))) {
      std::ostringstream os;
      os << "KokkosSparse::spmv: Dimensions do not match (transpose): "
         << ", A: " << A.numRows() << " x " << A.numCols() << ", x: " << x.extent(0) << " x " << x.extent(1)
         << ", y: " << y.extent(0) << " x " << y.extent(1);
      KokkosKernels::Impl::throw_runtime_exception(os.str());
    }
  }
  kkamd_crs_t desc      = Impl::make_crs_desc(A);
  auto* h               = handle->get_impl();
  kkamd_stream_t stream = reinterpret_cast<kkamd_stream_t>(space.hip_stream());
  // lazily create the per-matrix plan (the reference's tpl_rank1/2); SPMV_FAST_SETUP never analyses
  if (!h->plan && h->get_algorithm() != SPMV_FAST_SETUP && A.nnz() > 0 && mode[0] != Transpose[0] && mode[0] != ConjugateTranspose[0])
    Impl::kkamd_check(kkamd_spmv_plan_create(&h->plan, &desc, (int)h->get_algorithm(), stream));
  constexpr int vt = Impl::kkamd_scalar<typename YVector::non_const_value_type>::value;
  Kokkos::Profiling::pushRegion("KokkosSparse::spmv[KKAMD]");
  if constexpr (XVector::rank() == 1) {
    if (x.stride(0) != 1 || y.stride(0) != 1) {
      Impl::kkamd_check(kkamd_spmv_mv(h->plan, &desc, mode[0], (double)alpha, x.data(), (int64_t)x.stride(0), (int64_t)x.extent(0),
                                      (double)beta, y.data(), (int64_t)y.stride(0), (int64_t)y.extent(0), 1, vt, stream));
    } else {
      Impl::kkamd_check(kkamd_spmv(h->plan, &desc, mode[0], (double)alpha, x.data(), (double)beta, y.data(), vt, stream));
    }
  } else {
    Impl::kkamd_check(kkamd_spmv_mv(h->plan, &desc, mode[0], (double)alpha, x.data(), (int64_t)x.stride(0), (int64_t)x.stride(1),
                                    (double)beta, y.data(), (int64_t)y.stride(0), (int64_t)y.stride(1), (int64_t)x.extent(1), vt,
                                    stream));
  }
  Kokkos::Profiling::popRegion();
}

// overload 2: no execution space instance (:400-411)
template <class Handle, class AlphaType, class AMatrix, class XVector, class BetaType, class YVector,
          class = std::enable_if_t<!Kokkos::is_view<Handle>::value && !std::is_same<Handle, char>::value &&
                                   !std::is_same<Handle, const char>::value>>
void spmv(Handle* handle, const char mode[], const AlphaType& alpha, const AMatrix& A, const XVector& x,
          const BetaType& beta, const YVector& y) {
  spmv(typename Handle::ExecutionSpaceType(), handle, mode, alpha, A, x, beta, y);
}
// overload 3: execution space, no handle (:438-443): a throw-away SPMV_FAST_SETUP handle
template <class ExecutionSpace, class AlphaType, class AMatrix, class XVector, class BetaType, class YVector,
          class = std::enable_if_t<std::is_same<typename ExecutionSpace::execution_space, ExecutionSpace>::value>>
void spmv(const ExecutionSpace& space, const char mode[], const AlphaType& alpha, const AMatrix& A, const XVector& x,
          const BetaType& beta, const YVector& y) {
  SPMVHandle<typename AMatrix::device_type, AMatrix, XVector, YVector> handle(SPMV_FAST_SETUP);
  spmv(space, &handle, mode, alpha, A, x, beta, y);
}
// overload 4: neither (:464-474)
template <class AlphaType, class AMatrix, class XVector, class BetaType, class YVector>
void spmv(const char mode[], const AlphaType& alpha, const AMatrix& A, const XVector& x, const BetaType& beta,
          const YVector& y) {
  SPMVHandle<typename AMatrix::device_type, AMatrix, XVector, YVector> handle(SPMV_FAST_SETUP);
  spmv(typename AMatrix::execution_space(), &handle, mode, alpha, A, x, beta, y);
}

}  // namespace KokkosSparse
