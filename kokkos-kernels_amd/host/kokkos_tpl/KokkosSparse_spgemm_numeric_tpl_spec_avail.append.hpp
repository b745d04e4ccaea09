// APPEND to sparse/tpls/KokkosSparse_spgemm_numeric_tpl_spec_avail.hpp (after its closing #endif).
// Availability of the SpGEMM numeric phase through libkkamd; same tuples as the symbolic phase
// (KokkosSparse_spgemm_symbolic_tpl_spec_avail.append.hpp), rocSPARSE's all-int tuple (:82-110) left alone when it is on.
#ifndef KOKKOSPARSE_SPGEMM_NUMERIC_TPL_SPEC_AVAIL_KKAMD_HPP_
#define KOKKOSPARSE_SPGEMM_NUMERIC_TPL_SPEC_AVAIL_KKAMD_HPP_
#ifdef KOKKOSKERNELS_ENABLE_TPL_KKAMD
namespace KokkosSparse {
namespace Impl {

#define KKAMD_SPGEMM_VIEW(TYPE)                                                                                      \
  Kokkos::View<TYPE *, KokkosKernels::default_layout, Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>,                 \
               Kokkos::MemoryTraits<Kokkos::Unmanaged> >

#define SPGEMM_NUMERIC_AVAIL_KKAMD(SCALAR, OFFSET)                                                                   \
  template <>                                                                                                        \
  struct spgemm_numeric_tpl_spec_avail<                                                                              \
      KokkosKernels::Experimental::KokkosKernelsHandle<const OFFSET, const int, const SCALAR, Kokkos::HIP,           \
                                                       Kokkos::HIPSpace, Kokkos::HIPSpace>,                          \
      KKAMD_SPGEMM_VIEW(const OFFSET), KKAMD_SPGEMM_VIEW(const int), KKAMD_SPGEMM_VIEW(const SCALAR),                \
      KKAMD_SPGEMM_VIEW(const OFFSET), KKAMD_SPGEMM_VIEW(const int), KKAMD_SPGEMM_VIEW(const SCALAR),                \
      KKAMD_SPGEMM_VIEW(const OFFSET), KKAMD_SPGEMM_VIEW(int), KKAMD_SPGEMM_VIEW(SCALAR)> {                          \
    enum : bool { value = true };                                                                                    \
  };

#ifndef KOKKOSKERNELS_ENABLE_TPL_ROCSPARSE
SPGEMM_NUMERIC_AVAIL_KKAMD(float, int)
SPGEMM_NUMERIC_AVAIL_KKAMD(double, int)
#endif
SPGEMM_NUMERIC_AVAIL_KKAMD(float, size_t)
SPGEMM_NUMERIC_AVAIL_KKAMD(double, size_t)
#undef SPGEMM_NUMERIC_AVAIL_KKAMD
#undef KKAMD_SPGEMM_VIEW

}  // namespace Impl
}  // namespace KokkosSparse
#endif  // KOKKOSKERNELS_ENABLE_TPL_KKAMD
#endif  // KOKKOSPARSE_SPGEMM_NUMERIC_TPL_SPEC_AVAIL_KKAMD_HPP_
