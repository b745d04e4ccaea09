// APPEND to sparse/tpls/KokkosSparse_spmv_tpl_spec_avail.hpp (after its closing #endif).
// libkkamd (hand-written gfx950 kernels behind include/kkamd.h) as a TPL of kokkos-kernels: availability of the rank-1 SpMV.
// Coexistence with rocSPARSE: that TPL claims <float|double, offsets = rocsparse_int (int), ordinals int, HIPSpace |
// HIPManagedSpace> (KokkosSparse_spmv_tpl_spec_avail.hpp:126-150); a tuple can be specialised once, so the tuples both
// cover are specialised here only when rocSPARSE is off.  64-bit offsets (size_t) are claimed by nobody else and always are.
#ifndef KOKKOSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD_HPP_
#define KOKKOSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD_HPP_
#ifdef KOKKOSKERNELS_ENABLE_TPL_KKAMD
namespace KokkosSparse {
namespace Impl {

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

#ifndef KOKKOSKERNELS_ENABLE_TPL_ROCSPARSE
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(double, int, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(double, int, Kokkos::LayoutRight)
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(float, int, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(float, int, Kokkos::LayoutRight)
#endif
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(double, size_t, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(double, size_t, Kokkos::LayoutRight)
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(float, size_t, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD(float, size_t, Kokkos::LayoutRight)
#undef KOKKOSSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD

}  // namespace Impl
}  // namespace KokkosSparse
#endif  // KOKKOSKERNELS_ENABLE_TPL_KKAMD
#endif  // KOKKOSPARSE_SPMV_TPL_SPEC_AVAIL_KKAMD_HPP_
