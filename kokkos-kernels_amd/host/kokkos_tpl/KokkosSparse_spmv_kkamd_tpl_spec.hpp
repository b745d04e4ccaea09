// Reference-side binding of libkkamd's SpMV for a REAL kokkos-kernels build (Kokkos core present).
// Not compiled in this repository (no Kokkos here); it is what a maintainer adds next to the vendor wrappers:
//   part 1 goes at the end of sparse/tpls/KokkosSparse_spmv_tpl_spec_avail.hpp      (after :150)
//   part 2 goes at the end of sparse/tpls/KokkosSparse_spmv_tpl_spec_decl.hpp       (after :431)
//   part 3/4 likewise into ..._spmv_mv_tpl_spec_avail.hpp / ..._spmv_mv_tpl_spec_decl.hpp
// all guarded by a new CMake option KokkosKernels_ENABLE_TPL_KKAMD -> KOKKOSKERNELS_ENABLE_TPL_KKAMD
// (cmake/kokkoskernels_tpls.cmake:465-516, cmake/KokkosKernels_config.h.in:141-142) that adds
// -I<repo>/include and links libkkamd.so.
#pragma once
#if defined(KOKKOSKERNELS_ENABLE_TPL_KKAMD)
#include <kkamd.h>

namespace KokkosSparse {
namespace Impl {

// ---- part 1: availability (exact type tuples; int32 ordinals, int32 or 64-bit offsets, float/double) ----
#define KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(SCALAR, OFFSET, LAYOUT)                                              \
  template <>                                                                                                       \
  struct spmv_tpl_spec_avail<                                                                                       \
      Kokkos::HIP, KokkosSparse::Impl::SPMVHandleImpl<Kokkos::HIP, Kokkos::HIPSpace, SCALAR, OFFSET, int>,          \
      KokkosSparse::CrsMatrix<const SCALAR, const int, Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>,               \
                              Kokkos::MemoryTraits<Kokkos::Unmanaged>, const OFFSET>,                               \
      Kokkos::View<const SCALAR*, LAYOUT, Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>,                            \
                   Kokkos::MemoryTraits<Kokkos::Unmanaged | Kokkos::RandomAccess>>,                                 \
      Kokkos::View<SCALAR*, LAYOUT, Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>,                                  \
                   Kokkos::MemoryTraits<Kokkos::Unmanaged>>> {                                                      \
    enum : bool { value = true };                                                                                   \
  };
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(double, int, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(double, int, Kokkos::LayoutRight)
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(double, size_t, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(double, size_t, Kokkos::LayoutRight)
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(float, int, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(float, int, Kokkos::LayoutRight)
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(float, size_t, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(float, size_t, Kokkos::LayoutRight)
#undef KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD

// ---- part 2: the specialisation itself ---------------------------------------------------------------------
// per-matrix state stored where the vendor wrappers store theirs (sparse/src/KokkosSparse_spmv_handle.hpp:241)
struct KKAMD_CRS_SpMV_Data : public TPL_SpMV_Data<Kokkos::HIP> {
  KKAMD_CRS_SpMV_Data(const Kokkos::HIP& exec_) : TPL_SpMV_Data(exec_) {}
  ~KKAMD_CRS_SpMV_Data() { kkamd_spmv_plan_destroy(plan); }
  kkamd_spmv_plan_t* plan = nullptr;
};

inline void kkamd_safe_call(int status) {   // the *_SAFE_CALL convention: status -> std::runtime_error
  if (status != KKAMD_OK) {
    if (status == KKAMD_ERR_STATE) throw std::invalid_argument(kkamd_last_error());
    throw std::runtime_error(std::string("kkamd: ") + kkamd_last_error());
  }
}

template <class Handle, class AMatrix, class XVector, class YVector>
void spmv_kkamd(const Kokkos::HIP& exec, Handle* handle, const char mode[], typename YVector::const_value_type& alpha,
                const AMatrix& A, const XVector& x, typename YVector::const_value_type& beta, const YVector& y) {
  using offset_type = typename AMatrix::non_const_size_type;
  using value_type  = typename AMatrix::non_const_value_type;
  kkamd_crs_t d;
  d.num_rows = A.numRows(); d.num_cols = A.numCols(); d.nnz = A.nnz();
  d.d_row_map = A.graph.row_map.data(); d.d_entries = A.graph.entries.data(); d.d_values = A.values.data();
  d.offset_type = sizeof(offset_type) == 8 ? KKAMD_I64 : KKAMD_I32;
  d.value_type  = std::is_same<value_type, double>::value ? KKAMD_F64 : KKAMD_F32;
  kkamd_stream_t stream = reinterpret_cast<kkamd_stream_t>(exec.hip_stream());
  KKAMD_CRS_SpMV_Data* sub;
  if (handle->tpl_rank1) {
    sub = dynamic_cast<KKAMD_CRS_SpMV_Data*>(handle->tpl_rank1);
    if (!sub) throw std::runtime_error("KokkosSparse::spmv: subhandle is not set up for kkamd CRS");
    sub->set_exec_space(exec);
  } else {
    sub               = new KKAMD_CRS_SpMV_Data(exec);
    handle->tpl_rank1 = sub;
    if (handle->get_algorithm() != SPMV_FAST_SETUP)
      kkamd_safe_call(kkamd_spmv_plan_create(&sub->plan, &d, (int)handle->get_algorithm(), stream));
  }
  kkamd_safe_call(kkamd_spmv(sub->plan, &d, mode[0], (double)alpha, x.data(), (double)beta, y.data(),
                             std::is_same<typename YVector::non_const_value_type, double>::value ? KKAMD_F64 : KKAMD_F32,
                             stream));
}

#define KOKKOSSPARSE_SPMV_KKAMD(SCALAR, OFFSET, LAYOUT)                                                                 \
  template <>                                                                                                           \
  struct SPMV<Kokkos::HIP, KokkosSparse::Impl::SPMVHandleImpl<Kokkos::HIP, Kokkos::HIPSpace, SCALAR, OFFSET, int>,      \
              KokkosSparse::CrsMatrix<SCALAR const, int const, Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>,           \
                                      Kokkos::MemoryTraits<Kokkos::Unmanaged>, OFFSET const>,                           \
              Kokkos::View<SCALAR const*, LAYOUT, Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>,                        \
                           Kokkos::MemoryTraits<Kokkos::Unmanaged | Kokkos::RandomAccess>>,                             \
              Kokkos::View<SCALAR*, LAYOUT, Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>,                              \
                           Kokkos::MemoryTraits<Kokkos::Unmanaged>>,                                                    \
              true> {                                                                                                   \
    using device_type = Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>;                                                  \
    using Handle  = KokkosSparse::Impl::SPMVHandleImpl<Kokkos::HIP, Kokkos::HIPSpace, SCALAR, OFFSET, int>;             \
    using AMatrix = CrsMatrix<SCALAR const, int const, device_type, Kokkos::MemoryTraits<Kokkos::Unmanaged>, OFFSET const>; \
    using XVector = Kokkos::View<SCALAR const*, LAYOUT, device_type, Kokkos::MemoryTraits<Kokkos::Unmanaged | Kokkos::RandomAccess>>; \
    using YVector = Kokkos::View<SCALAR*, LAYOUT, device_type, Kokkos::MemoryTraits<Kokkos::Unmanaged>>;                \
    using coefficient_type = typename YVector::non_const_value_type;                                                    \
    static void spmv(const Kokkos::HIP& exec, Handle* handle, const char mode[], const coefficient_type& alpha,         \
                     const AMatrix& A, const XVector& x, const coefficient_type& beta, const YVector& y) {              \
      std::string label = "KokkosSparse::spmv[TPL_KKAMD," + Kokkos::ArithTraits<SCALAR>::name() + "]";                  \
      Kokkos::Profiling::pushRegion(label);                                                                             \
      spmv_kkamd(exec, handle, mode, alpha, A, x, beta, y);                                                             \
      Kokkos::Profiling::popRegion();                                                                                   \
    }                                                                                                                   \
  };
KOKKOSSPARSE_SPMV_KKAMD(double, int, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_KKAMD(double, int, Kokkos::LayoutRight)
KOKKOSSPARSE_SPMV_KKAMD(double, size_t, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_KKAMD(double, size_t, Kokkos::LayoutRight)
KOKKOSSPARSE_SPMV_KKAMD(float, int, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_KKAMD(float, int, Kokkos::LayoutRight)
KOKKOSSPARSE_SPMV_KKAMD(float, size_t, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_KKAMD(float, size_t, Kokkos::LayoutRight)
#undef KOKKOSSPARSE_SPMV_KKAMD

// ---- part 3/4: rank-2 (SPMV_MV).  Same pattern; the body unwraps strides instead of assuming a layout ---------
template <class Handle, class AMatrix, class XVector, class YVector>
void spmv_mv_kkamd(const Kokkos::HIP& exec, Handle* handle, const char mode[], typename YVector::const_value_type& alpha,
                   const AMatrix& A, const XVector& X, typename YVector::const_value_type& beta, const YVector& Y) {
  using offset_type = typename AMatrix::non_const_size_type;
  kkamd_crs_t d;
  d.num_rows = A.numRows(); d.num_cols = A.numCols(); d.nnz = A.nnz();
  d.d_row_map = A.graph.row_map.data(); d.d_entries = A.graph.entries.data(); d.d_values = A.values.data();
  d.offset_type = sizeof(offset_type) == 8 ? KKAMD_I64 : KKAMD_I32;
  d.value_type  = std::is_same<typename AMatrix::non_const_value_type, double>::value ? KKAMD_F64 : KKAMD_F32;
  kkamd_stream_t stream = reinterpret_cast<kkamd_stream_t>(exec.hip_stream());
  KKAMD_CRS_SpMV_Data* sub;
  if (handle->tpl_rank2) {
    sub = dynamic_cast<KKAMD_CRS_SpMV_Data*>(handle->tpl_rank2);
    if (!sub) throw std::runtime_error("KokkosSparse::spmv: subhandle is not set up for kkamd CRS");
    sub->set_exec_space(exec);
  } else {
    sub               = new KKAMD_CRS_SpMV_Data(exec);
    handle->tpl_rank2 = sub;
    if (handle->get_algorithm() != SPMV_FAST_SETUP)
      kkamd_safe_call(kkamd_spmv_plan_create(&sub->plan, &d, (int)handle->get_algorithm(), stream));
  }
  kkamd_safe_call(kkamd_spmv_mv(sub->plan, &d, mode[0], (double)alpha, X.data(), X.stride(0), X.stride(1), (double)beta,
                                Y.data(), Y.stride(0), Y.stride(1), X.extent(1),
                                std::is_same<typename YVector::non_const_value_type, double>::value ? KKAMD_F64 : KKAMD_F32,
                                stream));
}
// (spmv_mv_tpl_spec_avail<...> / SPMV_MV<..., false, true> specialisations follow the macro pattern above with
//  XVector = View<SCALAR const**, XL, ...>, YVector = View<SCALAR**, YL, ...> for XL, YL in {LayoutLeft, LayoutRight}.)

}  // namespace Impl
}  // namespace KokkosSparse
#endif  // KOKKOSKERNELS_ENABLE_TPL_KKAMD
