// APPEND to sparse/tpls/KokkosSparse_spmv_mv_tpl_spec_decl.hpp (after its closing #endif).
// Impl::SPMV_MV<..., integerScalarType = false, tpl_spec_avail = true> for the tuples of
// KokkosSparse_spmv_mv_tpl_spec_avail.append.hpp (pattern: KokkosSparse_spmv_mv_tpl_spec_decl.hpp:284-430): the Views'
// element strides go across as they are (any layout pair, any padding), the analysis lives in handle->tpl_rank2.
// Needs KokkosSparse_spmv_tpl_spec_decl.append.hpp (KKAMD_CRS_SpMV_Data, kkamd_subhandle) before it -- spmv_spec.hpp includes
// the rank-1 decl file first (sparse/impl/KokkosSparse_spmv_spec.hpp:268-270).
#ifndef KOKKOSPARSE_SPMV_MV_TPL_SPEC_DECL_KKAMD_HPP_
#define KOKKOSPARSE_SPMV_MV_TPL_SPEC_DECL_KKAMD_HPP_
#ifdef KOKKOSKERNELS_ENABLE_TPL_KKAMD
namespace KokkosSparse {
namespace Impl {

template <class Handle, class AMatrix, class XVector, class YVector>
void spmv_mv_kkamd(const Kokkos::HIP& exec, Handle* handle, const char mode[], typename YVector::const_value_type& alpha,
                   const AMatrix& A, const XVector& X, typename YVector::const_value_type& beta, const YVector& Y) {
  const kkamd_crs_t d      = kkamd_crs_desc(A);
  KKAMD_CRS_SpMV_Data* sub = kkamd_subhandle(exec, handle, handle->tpl_rank2, d);
  kkamd_safe_call(kkamd_spmv_mv(sub->plan, &d, mode[0], (double)alpha, X.data(), (int64_t)X.stride(0), (int64_t)X.stride(1),
                                (double)beta, Y.data(), (int64_t)Y.stride(0), (int64_t)Y.stride(1), (int64_t)X.extent(1),
                                std::is_same<typename YVector::non_const_value_type, double>::value ? KKAMD_F64 : KKAMD_F32,
                                reinterpret_cast<kkamd_stream_t>(exec.hip_stream())));
}

#define KOKKOSSPARSE_SPMV_MV_KKAMD(SCALAR, OFFSET, XL, YL)                                                              \
  template <>                                                                                                           \
  struct SPMV_MV<Kokkos::HIP, KokkosSparse::Impl::SPMVHandleImpl<Kokkos::HIP, Kokkos::HIPSpace, SCALAR, OFFSET, int>,   \
                 KokkosSparse::CrsMatrix<SCALAR const, int const, Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>,        \
                                         Kokkos::MemoryTraits<Kokkos::Unmanaged>, OFFSET const>,                        \
                 Kokkos::View<SCALAR const**, XL, Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>,                        \
                              Kokkos::MemoryTraits<Kokkos::Unmanaged | Kokkos::RandomAccess>>,                          \
                 Kokkos::View<SCALAR**, YL, Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>,                              \
                              Kokkos::MemoryTraits<Kokkos::Unmanaged>>,                                                 \
                 false, true> {                                                                                         \
    using device_type = Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>;                                                  \
    using Handle  = KokkosSparse::Impl::SPMVHandleImpl<Kokkos::HIP, Kokkos::HIPSpace, SCALAR, OFFSET, int>;             \
    using AMatrix = CrsMatrix<SCALAR const, int const, device_type, Kokkos::MemoryTraits<Kokkos::Unmanaged>, OFFSET const>; \
    using XVector = Kokkos::View<SCALAR const**, XL, device_type, Kokkos::MemoryTraits<Kokkos::Unmanaged | Kokkos::RandomAccess>>; \
    using YVector = Kokkos::View<SCALAR**, YL, device_type, Kokkos::MemoryTraits<Kokkos::Unmanaged>>;                   \
    using coefficient_type = typename YVector::non_const_value_type;                                                    \
    static void spmv_mv(const Kokkos::HIP& exec, Handle* handle, const char mode[], const coefficient_type& alpha,      \
                        const AMatrix& A, const XVector& x, const coefficient_type& beta, const YVector& y) {           \
      std::string label = "KokkosSparse::spmv_mv[TPL_KKAMD," + Kokkos::ArithTraits<SCALAR>::name() + "]";               \
      Kokkos::Profiling::pushRegion(label);                                                                             \
      spmv_mv_kkamd(exec, handle, mode, alpha, A, x, beta, y);                                                          \
      Kokkos::Profiling::popRegion();                                                                                   \
    }                                                                                                                   \
  };
#define KOKKOSSPARSE_SPMV_MV_KKAMD_LAYOUTS(SCALAR, OFFSET)                               \
  KOKKOSSPARSE_SPMV_MV_KKAMD(SCALAR, OFFSET, Kokkos::LayoutLeft, Kokkos::LayoutLeft)     \
  KOKKOSSPARSE_SPMV_MV_KKAMD(SCALAR, OFFSET, Kokkos::LayoutLeft, Kokkos::LayoutRight)    \
  KOKKOSSPARSE_SPMV_MV_KKAMD(SCALAR, OFFSET, Kokkos::LayoutRight, Kokkos::LayoutLeft)    \
  KOKKOSSPARSE_SPMV_MV_KKAMD(SCALAR, OFFSET, Kokkos::LayoutRight, Kokkos::LayoutRight)

#ifndef KOKKOSKERNELS_ENABLE_TPL_ROCSPARSE
KOKKOSSPARSE_SPMV_MV_KKAMD_LAYOUTS(double, int)
KOKKOSSPARSE_SPMV_MV_KKAMD_LAYOUTS(float, int)
#endif
KOKKOSSPARSE_SPMV_MV_KKAMD_LAYOUTS(double, size_t)
KOKKOSSPARSE_SPMV_MV_KKAMD_LAYOUTS(float, size_t)
#undef KOKKOSSPARSE_SPMV_MV_KKAMD_LAYOUTS
#undef KOKKOSSPARSE_SPMV_MV_KKAMD

}  // namespace Impl
}  // namespace KokkosSparse
#endif  // KOKKOSKERNELS_ENABLE_TPL_KKAMD
#endif  // KOKKOSPARSE_SPMV_MV_TPL_SPEC_DECL_KKAMD_HPP_
