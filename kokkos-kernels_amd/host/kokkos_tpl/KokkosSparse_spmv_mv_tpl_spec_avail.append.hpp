// APPEND to sparse/tpls/KokkosSparse_spmv_mv_tpl_spec_avail.hpp (after its closing #endif).
// Availability of the rank-2 SpMV (SPMV_MV) through libkkamd: all four layout pairs of X / Y (kkamd_spmv_mv takes element
// strides, include/kkamd.h), float / double, int / size_t offsets.  rocSPARSE coexistence as for rank 1
// (KokkosSparse_spmv_mv_tpl_spec_avail.hpp:108-134 claims the int-offset tuples).
#ifndef KOKKOSPARSE_SPMV_MV_TPL_SPEC_AVAIL_KKAMD_HPP_
#define KOKKOSPARSE_SPMV_MV_TPL_SPEC_AVAIL_KKAMD_HPP_
#ifdef KOKKOSKERNELS_ENABLE_TPL_KKAMD
namespace KokkosSparse {
namespace Impl {

#define KOKKOSSPARSE_SPMV_MV_TPL_SPEC_AVAIL_KKAMD(SCALAR, OFFSET, XL, YL)                                           \
  template <>                                                                                                       \
  struct spmv_mv_tpl_spec_avail<                                                                                    \
      Kokkos::HIP, KokkosSparse::Impl::SPMVHandleImpl<Kokkos::HIP, Kokkos::HIPSpace, SCALAR, OFFSET, int>,          \
      KokkosSparse::CrsMatrix<const SCALAR, const int, Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>,               \
                              Kokkos::MemoryTraits<Kokkos::Unmanaged>, const OFFSET>,                               \
      Kokkos::View<const SCALAR**, XL, Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>,                               \
                   Kokkos::MemoryTraits<Kokkos::Unmanaged | Kokkos::RandomAccess>>,                                 \
      Kokkos::View<SCALAR**, YL, Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>,                                     \
                   Kokkos::MemoryTraits<Kokkos::Unmanaged>>> {                                                      \
    enum : bool { value = true };                                                                                   \
  };
#define KOKKOSSPARSE_SPMV_MV_TPL_SPEC_AVAIL_KKAMD_LAYOUTS(SCALAR, OFFSET)                               \
  KOKKOSSPARSE_SPMV_MV_TPL_SPEC_AVAIL_KKAMD(SCALAR, OFFSET, Kokkos::LayoutLeft, Kokkos::LayoutLeft)     \
  KOKKOSSPARSE_SPMV_MV_TPL_SPEC_AVAIL_KKAMD(SCALAR, OFFSET, Kokkos::LayoutLeft, Kokkos::LayoutRight)    \
  KOKKOSSPARSE_SPMV_MV_TPL_SPEC_AVAIL_KKAMD(SCALAR, OFFSET, Kokkos::LayoutRight, Kokkos::LayoutLeft)    \
  KOKKOSSPARSE_SPMV_MV_TPL_SPEC_AVAIL_KKAMD(SCALAR, OFFSET, Kokkos::LayoutRight, Kokkos::LayoutRight)

#ifndef KOKKOSKERNELS_ENABLE_TPL_ROCSPARSE
KOKKOSSPARSE_SPMV_MV_TPL_SPEC_AVAIL_KKAMD_LAYOUTS(double, int)
KOKKOSSPARSE_SPMV_MV_TPL_SPEC_AVAIL_KKAMD_LAYOUTS(float, int)
#endif
KOKKOSSPARSE_SPMV_MV_TPL_SPEC_AVAIL_KKAMD_LAYOUTS(double, size_t)
KOKKOSSPARSE_SPMV_MV_TPL_SPEC_AVAIL_KKAMD_LAYOUTS(float, size_t)
#undef KOKKOSSPARSE_SPMV_MV_TPL_SPEC_AVAIL_KKAMD_LAYOUTS
#undef KOKKOSSPARSE_SPMV_MV_TPL_SPEC_AVAIL_KKAMD

}  // namespace Impl
}  // namespace KokkosSparse
#endif  // KOKKOSKERNELS_ENABLE_TPL_KKAMD
#endif  // KOKKOSPARSE_SPMV_MV_TPL_SPEC_AVAIL_KKAMD_HPP_
