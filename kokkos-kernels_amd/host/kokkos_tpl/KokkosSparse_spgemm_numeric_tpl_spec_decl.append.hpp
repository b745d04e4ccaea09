// APPEND to sparse/tpls/KokkosSparse_spgemm_numeric_tpl_spec_decl.hpp (after its closing #endif).
// Impl::SPGEMM_NUMERIC<..., tpl_spec_avail = true, eti_spec_avail> for the tuples of
// KokkosSparse_spgemm_numeric_tpl_spec_avail.append.hpp (pattern: SPGEMM_NUMERIC_DECL_ROCSPARSE, :259-329).  Plug-in contract
// (:288-329): entries(C) are filled only while !are_entries_computed() -- kkamd_spgemm_numeric keeps the structure of C in
// its handle and recomputes values alone on a repeated call (the reference's reuse case) -- then call_numeric is set.
// C leaves column-sorted, so the trailing sort_crs_matrix of the native path (impl/KokkosSparse_spgemm_numeric_spec.hpp:138-140)
// has nothing to do.
#ifndef KOKKOSPARSE_SPGEMM_NUMERIC_TPL_SPEC_DECL_KKAMD_HPP_
#define KOKKOSPARSE_SPGEMM_NUMERIC_TPL_SPEC_DECL_KKAMD_HPP_
#ifdef KOKKOSKERNELS_ENABLE_TPL_KKAMD
#include <kkamd.h>

namespace KokkosSparse {
namespace Impl {

inline void kkamd_spgemm_numeric_safe_call(int status) {
  if (status != KKAMD_OK) {
    if (status == KKAMD_ERR_STATE) throw std::invalid_argument(kkamd_last_error());
    throw std::runtime_error(std::string("kkamd: ") + kkamd_last_error());
  }
}

template <typename KernelHandle, typename ain_row_index_view_type, typename ain_nonzero_index_view_type,
          typename ain_nonzero_value_view_type, typename bin_row_index_view_type, typename bin_nonzero_index_view_type,
          typename bin_nonzero_value_view_type, typename cin_row_index_view_type, typename cin_nonzero_index_view_type,
          typename cin_nonzero_value_view_type>
void spgemm_numeric_kkamd(KernelHandle *kh, typename KernelHandle::nnz_lno_t m, typename KernelHandle::nnz_lno_t n,
                          typename KernelHandle::nnz_lno_t k, ain_row_index_view_type rowptrA,
                          ain_nonzero_index_view_type colidxA, ain_nonzero_value_view_type valuesA,
                          bin_row_index_view_type rowptrB, bin_nonzero_index_view_type colidxB,
                          bin_nonzero_value_view_type valuesB, cin_row_index_view_type rowptrC,
                          cin_nonzero_index_view_type colidxC, cin_nonzero_value_view_type valuesC) {
  auto *handle             = kh->get_spgemm_handle();
  kkamd_spgemm_handle_t *h = handle->get_kkamd_spgemm_handle();
  if (!h) throw std::invalid_argument("KokkosSparse::spgemm_numeric: must first call spgemm_symbolic with the same handle");
  // a caller that re-allocated entries(C) says so through the reference's flag: the structure is then written again
  kkamd_spgemm_numeric_safe_call(kkamd_spgemm_set(h, "entries_computed", handle->are_entries_computed() ? 1.0 : 0.0));
  kkamd_spgemm_numeric_safe_call(kkamd_spgemm_numeric(
      h, (int64_t)m, (int64_t)n, (int64_t)k, rowptrA.data(), colidxA.data(), valuesA.data(), rowptrB.data(), colidxB.data(), valuesB.data(),
      rowptrC.data(), (int32_t *)colidxC.data(), (void *)valuesC.data(), sizeof(typename KernelHandle::size_type) == 8 ? KKAMD_I64 : KKAMD_I32,
      std::is_same<typename KernelHandle::nnz_scalar_t, double>::value ? KKAMD_F64 : KKAMD_F32, nullptr));
  handle->set_computed_entries();
  handle->set_call_numeric();
}

#define KKAMD_SPGEMM_VIEW(TYPE)                                                                                      \
  Kokkos::View<TYPE *, KokkosKernels::default_layout, Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>,                 \
               Kokkos::MemoryTraits<Kokkos::Unmanaged> >

#define SPGEMM_NUMERIC_DECL_KKAMD(SCALAR, OFFSET, ETI_AVAIL)                                                         \
  template <>                                                                                                        \
  struct SPGEMM_NUMERIC<                                                                                             \
      KokkosKernels::Experimental::KokkosKernelsHandle<const OFFSET, const int, const SCALAR, Kokkos::HIP,           \
                                                       Kokkos::HIPSpace, Kokkos::HIPSpace>,                          \
      KKAMD_SPGEMM_VIEW(const OFFSET), KKAMD_SPGEMM_VIEW(const int), KKAMD_SPGEMM_VIEW(const SCALAR),                \
      KKAMD_SPGEMM_VIEW(const OFFSET), KKAMD_SPGEMM_VIEW(const int), KKAMD_SPGEMM_VIEW(const SCALAR),                \
      KKAMD_SPGEMM_VIEW(const OFFSET), KKAMD_SPGEMM_VIEW(int), KKAMD_SPGEMM_VIEW(SCALAR), true, ETI_AVAIL> {         \
    using KernelHandle =                                                                                             \
        KokkosKernels::Experimental::KokkosKernelsHandle<const OFFSET, const int, const SCALAR, Kokkos::HIP,         \
                                                         Kokkos::HIPSpace, Kokkos::HIPSpace>;                        \
    using c_offset_view_t = KKAMD_SPGEMM_VIEW(const OFFSET);                                                         \
    using c_int_view_t    = KKAMD_SPGEMM_VIEW(const int);                                                            \
    using int_view_t      = KKAMD_SPGEMM_VIEW(int);                                                                  \
    using c_scalar_view_t = KKAMD_SPGEMM_VIEW(const SCALAR);                                                         \
    using scalar_view_t   = KKAMD_SPGEMM_VIEW(SCALAR);                                                               \
    static void spgemm_numeric(KernelHandle *handle, typename KernelHandle::nnz_lno_t m,                             \
                               typename KernelHandle::nnz_lno_t n, typename KernelHandle::nnz_lno_t k,               \
                               c_offset_view_t row_mapA, c_int_view_t entriesA, c_scalar_view_t valuesA, bool,       \
                               c_offset_view_t row_mapB, c_int_view_t entriesB, c_scalar_view_t valuesB, bool,       \
                               c_offset_view_t row_mapC, int_view_t entriesC, scalar_view_t valuesC) {               \
      std::string label = "KokkosSparse::spgemm_numeric[TPL_KKAMD," + Kokkos::ArithTraits<SCALAR>::name() + "]";     \
      Kokkos::Profiling::pushRegion(label);                                                                          \
      spgemm_numeric_kkamd(handle, m, n, k, row_mapA, entriesA, valuesA, row_mapB, entriesB, valuesB, row_mapC,      \
                           entriesC, valuesC);                                                                       \
      Kokkos::Profiling::popRegion();                                                                                \
    }                                                                                                                \
  };
#define SPGEMM_NUMERIC_DECL_KKAMD_E(SCALAR, OFFSET) \
  SPGEMM_NUMERIC_DECL_KKAMD(SCALAR, OFFSET, true)   \
  SPGEMM_NUMERIC_DECL_KKAMD(SCALAR, OFFSET, false)

#ifndef KOKKOSKERNELS_ENABLE_TPL_ROCSPARSE
SPGEMM_NUMERIC_DECL_KKAMD_E(float, int)
SPGEMM_NUMERIC_DECL_KKAMD_E(double, int)
#endif
SPGEMM_NUMERIC_DECL_KKAMD_E(float, size_t)
SPGEMM_NUMERIC_DECL_KKAMD_E(double, size_t)
#undef SPGEMM_NUMERIC_DECL_KKAMD_E
#undef SPGEMM_NUMERIC_DECL_KKAMD
#undef KKAMD_SPGEMM_VIEW

}  // namespace Impl
}  // namespace KokkosSparse
#endif  // KOKKOSKERNELS_ENABLE_TPL_KKAMD
#endif  // KOKKOSPARSE_SPGEMM_NUMERIC_TPL_SPEC_DECL_KKAMD_HPP_
