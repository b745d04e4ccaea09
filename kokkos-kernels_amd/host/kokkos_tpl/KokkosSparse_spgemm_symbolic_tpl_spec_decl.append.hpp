// APPEND to sparse/tpls/KokkosSparse_spgemm_symbolic_tpl_spec_decl.hpp (after its closing #endif).
// Impl::SPGEMM_SYMBOLIC<..., tpl_spec_avail = true, eti_spec_avail> for the tuples of
// KokkosSparse_spgemm_symbolic_tpl_spec_avail.append.hpp (pattern: SPGEMM_SYMBOLIC_DECL_ROCSPARSE, :367-446).  The phase
// fills row_mapC and leaves nnz(C) in the handle (kkamd_spgemm_symbolic, include/kkamd.h); cross-phase state is the
// kkamd_spgemm_handle_t the patched SPGEMMHandle owns (KokkosSparse_spgemm_handle.hpp.patch).  The plug-in contract
// (:391-446): idempotent once called, sets c_nnz, call_symbolic, computed_rowptrs.
// Unlike rocSPARSE's, these kernels accept unsorted and unmerged rows of A and B; the reference's front end still checks
// sortedness in debug builds whenever a TPL is bound (sparse/src/KokkosSparse_spgemm_symbolic.hpp:128-141).
#ifndef KOKKOSPARSE_SPGEMM_SYMBOLIC_TPL_SPEC_DECL_KKAMD_HPP_
#define KOKKOSPARSE_SPGEMM_SYMBOLIC_TPL_SPEC_DECL_KKAMD_HPP_
#ifdef KOKKOSKERNELS_ENABLE_TPL_KKAMD
#include <kkamd.h>
#include <stdexcept>
#include <string>

namespace KokkosSparse {
namespace Impl {

inline void kkamd_spgemm_safe_call(int status) {
  if (status != KKAMD_OK) {
    if (status == KKAMD_ERR_STATE) throw std::invalid_argument(kkamd_last_error());
    throw std::runtime_error(std::string("kkamd: ") + kkamd_last_error());
  }
}

// the tuning state a caller left in the reference's handles travels as recorded hints / options (include/kkamd.h)
template <class KernelHandle>
inline void kkamd_spgemm_forward_options(KernelHandle *kh, kkamd_spgemm_handle_t *h) {
  auto *sh = kh->get_spgemm_handle();
  kkamd_spgemm_safe_call(kkamd_spgemm_set(h, "algorithm", (double)(int)sh->get_algorithm_type()));
  kkamd_spgemm_safe_call(kkamd_spgemm_set(h, "verbose", kh->get_verbose() ? 1.0 : 0.0));
  kkamd_spgemm_safe_call(kkamd_spgemm_set(h, "team_work_size", (double)kh->get_set_team_work_size()));
  kkamd_spgemm_safe_call(kkamd_spgemm_set(h, "shmem_size", (double)kh->get_shmem_size()));
  kkamd_spgemm_safe_call(kkamd_spgemm_set(h, "suggested_team_size", (double)kh->get_set_suggested_team_size()));
  kkamd_spgemm_safe_call(kkamd_spgemm_set(h, "suggested_vector_size", (double)kh->get_set_suggested_vector_size()));
  kkamd_spgemm_safe_call(kkamd_spgemm_set(h, "dynamic_scheduling", kh->is_dynamic_scheduling() ? 1.0 : 0.0));
  kkamd_spgemm_safe_call(kkamd_spgemm_set(h, "min_hash_size_scale", (double)sh->get_min_hash_size_scale()));
  kkamd_spgemm_safe_call(kkamd_spgemm_set(h, "first_level_hash_cut_off", sh->get_first_level_hash_cut_off()));
  if (sh->get_compression_cut_off() > 0.0 && sh->get_compression_cut_off() <= 1.0)
    kkamd_spgemm_safe_call(kkamd_spgemm_set(h, "compression_cut_off", sh->get_compression_cut_off()));
}

template <typename KernelHandle, typename ain_row_index_view_type, typename ain_nonzero_index_view_type,
          typename bin_row_index_view_type, typename bin_nonzero_index_view_type, typename cin_row_index_view_type>
void spgemm_symbolic_kkamd(KernelHandle *kh, typename KernelHandle::nnz_lno_t m, typename KernelHandle::nnz_lno_t n,
                           typename KernelHandle::nnz_lno_t k, ain_row_index_view_type rowptrA,
                           ain_nonzero_index_view_type colidxA, bin_row_index_view_type rowptrB,
                           bin_nonzero_index_view_type colidxB, cin_row_index_view_type rowptrC) {
  auto *handle = kh->get_spgemm_handle();
  if (handle->is_symbolic_called()) return;
  handle->create_kkamd_spgemm_handle();
  kkamd_spgemm_handle_t *h = handle->get_kkamd_spgemm_handle();
  kkamd_spgemm_forward_options(kh, h);
  int64_t nnz_C = 0;
  // the phase synchronises the stream it is given (the default stream here, like the reference's KKMEM phases:
  // sparse/impl/KokkosSparse_spgemm_impl_kkmem.hpp:1440,1467): nnz(C) must reach the host before C can be allocated
  kkamd_spgemm_safe_call(kkamd_spgemm_symbolic(h, (int64_t)m, (int64_t)n, (int64_t)k, rowptrA.data(), colidxA.data(), rowptrB.data(),
                                               colidxB.data(), (void *)rowptrC.data(),
                                               sizeof(typename KernelHandle::size_type) == 8 ? KKAMD_I64 : KKAMD_I32, &nnz_C,
                                               nullptr));
  handle->set_c_nnz((typename KernelHandle::size_type)nnz_C);
  handle->set_call_symbolic();
  handle->set_computed_rowptrs();
}

#define KKAMD_SPGEMM_VIEW(TYPE)                                                                                      \
  Kokkos::View<TYPE *, KokkosKernels::default_layout, Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>,                 \
               Kokkos::MemoryTraits<Kokkos::Unmanaged> >

#define SPGEMM_SYMBOLIC_DECL_KKAMD(SCALAR, OFFSET, ETI_AVAIL)                                                        \
  template <>                                                                                                        \
  struct SPGEMM_SYMBOLIC<                                                                                            \
      KokkosKernels::Experimental::KokkosKernelsHandle<const OFFSET, const int, const SCALAR, Kokkos::HIP,           \
                                                       Kokkos::HIPSpace, Kokkos::HIPSpace>,                          \
      KKAMD_SPGEMM_VIEW(const OFFSET), KKAMD_SPGEMM_VIEW(const int), KKAMD_SPGEMM_VIEW(const OFFSET),                \
      KKAMD_SPGEMM_VIEW(const int), KKAMD_SPGEMM_VIEW(OFFSET), true, ETI_AVAIL> {                                    \
    using KernelHandle =                                                                                             \
        KokkosKernels::Experimental::KokkosKernelsHandle<const OFFSET, const int, const SCALAR, Kokkos::HIP,         \
                                                         Kokkos::HIPSpace, Kokkos::HIPSpace>;                        \
    using c_offset_view_t = KKAMD_SPGEMM_VIEW(const OFFSET);                                                         \
    using c_int_view_t    = KKAMD_SPGEMM_VIEW(const int);                                                            \
    using offset_view_t   = KKAMD_SPGEMM_VIEW(OFFSET);                                                               \
    static void spgemm_symbolic(KernelHandle *handle, typename KernelHandle::nnz_lno_t m,                            \
                                typename KernelHandle::nnz_lno_t n, typename KernelHandle::nnz_lno_t k,              \
                                c_offset_view_t row_mapA, c_int_view_t entriesA, bool, c_offset_view_t row_mapB,     \
                                c_int_view_t entriesB, bool, offset_view_t row_mapC, bool) {                         \
      std::string label = "KokkosSparse::spgemm_symbolic[TPL_KKAMD," + Kokkos::ArithTraits<SCALAR>::name() + "]";    \
      Kokkos::Profiling::pushRegion(label);                                                                          \
      spgemm_symbolic_kkamd(handle, m, n, k, row_mapA, entriesA, row_mapB, entriesB, row_mapC);                      \
      Kokkos::Profiling::popRegion();                                                                                \
    }                                                                                                                \
  };
#define SPGEMM_SYMBOLIC_DECL_KKAMD_E(SCALAR, OFFSET) \
  SPGEMM_SYMBOLIC_DECL_KKAMD(SCALAR, OFFSET, true)   \
  SPGEMM_SYMBOLIC_DECL_KKAMD(SCALAR, OFFSET, false)

#ifndef KOKKOSKERNELS_ENABLE_TPL_ROCSPARSE
SPGEMM_SYMBOLIC_DECL_KKAMD_E(float, int)
SPGEMM_SYMBOLIC_DECL_KKAMD_E(double, int)
#endif
SPGEMM_SYMBOLIC_DECL_KKAMD_E(float, size_t)
SPGEMM_SYMBOLIC_DECL_KKAMD_E(double, size_t)
#undef SPGEMM_SYMBOLIC_DECL_KKAMD_E
#undef SPGEMM_SYMBOLIC_DECL_KKAMD
#undef KKAMD_SPGEMM_VIEW

}  // namespace Impl
}  // namespace KokkosSparse
#endif  // KOKKOSKERNELS_ENABLE_TPL_KKAMD
#endif  // KOKKOSPARSE_SPGEMM_SYMBOLIC_TPL_SPEC_DECL_KKAMD_HPP_
