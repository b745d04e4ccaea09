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
  // lazily create the per-matrix plan (the reference's tpl_rank1/2) on the first call of ANY mode: the cached-transpose path of
  // modes T / H hangs off the plan too (3.1 ms instead of 11.2 ms of atomics on C2); SPMV_FAST_SETUP never analyses
  if (!h->plan && h->get_algorithm() != SPMV_FAST_SETUP && A.nnz() > 0)
  {
    std::vector<const char*> keys; std::vector<int> vals;
    for (auto& kv : h->pending_) { keys.push_back(kv.first.c_str()); vals.push_back(kv.second); }
    if (h->vector_length > 0) { keys.push_back("lanes_per_row"); vals.push_back(h->vector_length); }
    Impl::kkamd_check(kkamd_spmv_plan_create_knobs(&h->plan, &desc, (int)h->get_algorithm(), keys.data(), vals.data(), (int)keys.size(), stream));
  }
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

// ---------------------------------------------------------------------------------------------------------------
// KokkosSparse::Experimental::spmv_struct (sparse/src/KokkosSparse_spmv.hpp:478-848): SpMV for a matrix that comes
// from a 3/5/9/7/27-point stencil; `structure` is a HOST view of the grid extents.  Same overload set as the reference:
// {space, no space} x {explicit RANK_ONE / RANK_TWO tag, rank deduced}.  Rank-2 with one column takes the structured
// path, more columns fall through to spmv (:803-831).
struct RANK_ONE {};
struct RANK_TWO {};
namespace Experimental {

template <class ExecutionSpace, class AlphaType, class AMatrix, class XVector, class BetaType, class YVector>
void spmv_struct(const ExecutionSpace& space, const char mode[], const int stencil_type,
                 const Kokkos::View<typename AMatrix::non_const_ordinal_type*, Kokkos::HostSpace>& structure,
                 const AlphaType& alpha, const AMatrix& A, const XVector& x, const BetaType& beta, const YVector& y,
                 const RANK_ONE&) {
  static_assert((int)XVector::rank() == (int)YVector::rank(), "KokkosSparse::spmv_struct: Vector ranks do not match.");
  static_assert((int)XVector::rank() == 1,
                "KokkosSparse::spmv_struct: Both Vector inputs must have rank 1 in order to call this specialization of spmv.");
  static_assert(!std::is_const<typename YVector::value_type>::value, "KokkosSparse::spmv_struct: Output Vector must be non-const.");
  // the reference only requires the vectors to be long enough (:504-523)
  if ((mode[0] == NoTranspose[0]) || (mode[0] == Conjugate[0])) {
    if ((x.extent(1) != y.extent(1)) || (static_cast<size_t>(A.numCols()) > static_cast<size_t>(x.extent(0))) ||
        (static_cast<size_t>(A.numRows()) > static_cast<size_t>(y.extent(0)))) {
      std::ostringstream os;
      os << "KokkosSparse::spmv_struct: Dimensions do not match: "
         << ", A: " << A.numRows() << " x " << A.numCols() << ", x: " << x.extent(0) << " x " << x.extent(1)
         << ", y: " << y.extent(0) << " x " << y.extent(1);
      KokkosKernels::Impl::throw_runtime_exception(os.str());
    }
  } else {
    if ((x.extent(1) != y.extent(1)) || (static_cast<size_t>(A.numCols()) > static_cast<size_t>(y.extent(0))) ||
        (static_cast<size_t>(A.numRows()) > static_cast<size_t>(x.extent(0)))) {
      std::ostringstream os;
      os << "KokkosSparse::spmv_struct: Dimensions do not match (transpose): "
         << ", A: " << A.numRows() << " x " << A.numCols() << ", x: " << x.extent(0) << " x " << x.extent(1)
         << ", y: " << y.extent(0) << " x " << y.extent(1);
      KokkosKernels::Impl::throw_runtime_exception(os.str());
    }
  }
  if (x.stride(0) != 1 || y.stride(0) != 1) {     // strided rank-1 views: the unstructured path handles them
    KokkosSparse::spmv(space, mode, alpha, A, x, beta, y);
    return;
  }
  kkamd_crs_t desc = Impl::make_crs_desc(A);
  int64_t ext[3]   = {1, 1, 1};
  const int ndim   = (int)structure.extent(0);
  for (int q = 0; q < ndim && q < 3; ++q) ext[q] = (int64_t)structure(q);
  constexpr int vt = Impl::kkamd_scalar<typename YVector::non_const_value_type>::value;
  Kokkos::Profiling::pushRegion("KokkosSparse::spmv_struct[KKAMD]");
  Impl::kkamd_check(kkamd_spmv_struct(&desc, mode[0], stencil_type, ndim, ext, (double)alpha, x.data(), (double)beta, y.data(), vt,
                                      reinterpret_cast<kkamd_stream_t>(space.hip_stream())));
  Kokkos::Profiling::popRegion();
}

template <class ExecutionSpace, class AlphaType, class AMatrix, class XVector, class BetaType, class YVector>
void spmv_struct(const ExecutionSpace& space, const char mode[], const int stencil_type,
                 const Kokkos::View<typename AMatrix::non_const_ordinal_type*, Kokkos::HostSpace>& structure,
                 const AlphaType& alpha, const AMatrix& A, const XVector& x, const BetaType& beta, const YVector& y,
                 const RANK_TWO&) {
  static_assert(XVector::rank() == YVector::rank(), "KokkosSparse::spmv: Vector ranks do not match.");
  static_assert(!std::is_const<typename YVector::value_type>::value, "KokkosSparse::spmv: Output Vector must be non-const.");
  if (x.extent(1) == 1 && x.extent(1) == y.extent(1)) {
    auto x0 = Kokkos::subview(x, Kokkos::ALL(), 0);
    auto y0 = Kokkos::subview(y, Kokkos::ALL(), 0);
    spmv_struct(space, mode, stencil_type, structure, alpha, A, x0, beta, y0, RANK_ONE());
    return;
  }
  KokkosSparse::spmv(space, mode, alpha, A, x, beta, y);
}

template <class AlphaType, class AMatrix, class XVector, class BetaType, class YVector, class Tag,
          class = std::enable_if_t<std::is_same<Tag, RANK_ONE>::value || std::is_same<Tag, RANK_TWO>::value>>
void spmv_struct(const char mode[], const int stencil_type,
                 const Kokkos::View<typename AMatrix::non_const_ordinal_type*, Kokkos::HostSpace>& structure,
                 const AlphaType& alpha, const AMatrix& A, const XVector& x, const BetaType& beta, const YVector& y,
                 const Tag& tag) {
  spmv_struct(typename AMatrix::execution_space{}, mode, stencil_type, structure, alpha, A, x, beta, y, tag);
}

template <class AlphaType, class AMatrix, class XVector, class BetaType, class YVector>
void spmv_struct(const char mode[], const int stencil_type,
                 const Kokkos::View<typename AMatrix::non_const_ordinal_type*, Kokkos::HostSpace>& structure,
                 const AlphaType& alpha, const AMatrix& A, const XVector& x, const BetaType& beta, const YVector& y) {
  using RANK_SPECIALISE = typename std::conditional<XVector::rank() == 2, RANK_TWO, RANK_ONE>::type;
  spmv_struct(mode, stencil_type, structure, alpha, A, x, beta, y, RANK_SPECIALISE());
}

template <class ExecutionSpace, class AlphaType, class AMatrix, class XVector, class BetaType, class YVector,
          class = std::enable_if_t<std::is_same<typename ExecutionSpace::execution_space, ExecutionSpace>::value>>
void spmv_struct(const ExecutionSpace& space, const char mode[], const int stencil_type,
                 const Kokkos::View<typename AMatrix::non_const_ordinal_type*, Kokkos::HostSpace>& structure,
                 const AlphaType& alpha, const AMatrix& A, const XVector& x, const BetaType& beta, const YVector& y) {
  using RANK_SPECIALISE = typename std::conditional<XVector::rank() == 2, RANK_TWO, RANK_ONE>::type;
  spmv_struct(space, mode, stencil_type, structure, alpha, A, x, beta, y, RANK_SPECIALISE());
}

}  // namespace Experimental

}  // namespace KokkosSparse
