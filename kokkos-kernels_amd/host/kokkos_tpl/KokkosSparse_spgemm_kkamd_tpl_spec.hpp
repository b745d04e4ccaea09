// Reference-side binding of libkkamd's SpGEMM for a REAL kokkos-kernels build.  Not compiled here.
//   availability  -> end of sparse/tpls/KokkosSparse_spgemm_symbolic_tpl_spec_avail.hpp (:72-95 pattern) and
//                    ..._spgemm_numeric_tpl_spec_avail.hpp (:82-110)
//   declarations  -> end of ..._spgemm_symbolic_tpl_spec_decl.hpp / ..._spgemm_numeric_tpl_spec_decl.hpp
//   handle state  -> one member in SPGEMMHandle (sparse/src/KokkosSparse_spgemm_handle.hpp, next to
//                    rocsparse_spgemm_handle :516-531):  kkamd_spgemm_handle_t* kkamd_handle = nullptr;
//                    destroyed in ~SPGEMMHandle with kkamd_spgemm_destroy().
#pragma once
#if defined(KOKKOSKERNELS_ENABLE_TPL_KKAMD)
#include <kkamd.h>

namespace KokkosSparse {
namespace Impl {

template <class KernelHandle, class ARow, class AEnt, class BRow, class BEnt, class CRow>
void spgemm_symbolic_kkamd(KernelHandle* handle, typename KernelHandle::nnz_lno_t m, typename KernelHandle::nnz_lno_t n,
                           typename KernelHandle::nnz_lno_t k, ARow rowptrA, AEnt colidxA, BRow rowptrB, BEnt colidxB,
                           CRow rowptrC) {
  auto* sh = handle->get_spgemm_handle();
  if (sh->is_symbolic_called()) return;                                   // idempotent, like every other back-end
  if (!sh->kkamd_handle) kkamd_safe_call(kkamd_spgemm_create(&sh->kkamd_handle));
  int64_t nnzC = 0;
  kkamd_safe_call(kkamd_spgemm_symbolic(sh->kkamd_handle, m, n, k, rowptrA.data(), colidxA.data(), rowptrB.data(),
                                        colidxB.data(), rowptrC.data(),
                                        sizeof(typename KernelHandle::size_type) == 8 ? KKAMD_I64 : KKAMD_I32, &nnzC,
                                        nullptr /* the phase synchronises: nnz(C) must reach the host */));
  sh->set_c_nnz(nnzC);                    // the flags the plug-in must set:
  sh->set_call_symbolic();                // sparse/tpls/KokkosSparse_spgemm_symbolic_tpl_spec_decl.hpp:391-446
  sh->set_computed_rowptrs();
}

template <class KernelHandle, class ARow, class AEnt, class AVal, class BRow, class BEnt, class BVal, class CRow, class CEnt, class CVal>
void spgemm_numeric_kkamd(KernelHandle* handle, typename KernelHandle::nnz_lno_t m, typename KernelHandle::nnz_lno_t n,
                          typename KernelHandle::nnz_lno_t k, ARow rowptrA, AEnt colidxA, AVal valuesA, BRow rowptrB,
                          BEnt colidxB, BVal valuesB, CRow rowptrC, CEnt colidxC, CVal valuesC) {
  auto* sh = handle->get_spgemm_handle();
  kkamd_safe_call(kkamd_spgemm_numeric(sh->kkamd_handle, m, n, k, rowptrA.data(), colidxA.data(), valuesA.data(),
                                       rowptrB.data(), colidxB.data(), valuesB.data(), rowptrC.data(), colidxC.data(),
                                       valuesC.data(), sizeof(typename KernelHandle::size_type) == 8 ? KKAMD_I64 : KKAMD_I32,
                                       std::is_same<typename KernelHandle::nnz_scalar_t, double>::value ? KKAMD_F64 : KKAMD_F32,
                                       nullptr));
  sh->set_computed_entries();             // entries are (re)written, column-sorted, on every numeric call
  sh->set_call_numeric();
}

// SPGEMM_SYMBOLIC<KokkosKernelsHandle<const OFFSET, const int, const SCALAR, Kokkos::HIP, Kokkos::HIPSpace,
// Kokkos::HIPSpace>, View<const OFFSET*,...>, View<const int*,...>, ..., true, ETI> and the matching
// SPGEMM_NUMERIC specialisation forward to the two functions above (macro pattern of
// SPGEMM_SYMBOLIC_DECL_ROCSPARSE / SPGEMM_NUMERIC_DECL_ROCSPARSE), for SCALAR in {float, double} and
// OFFSET in {int, size_t} -- 64-bit offsets are NOT covered by rocSPARSE today and are needed for
// R-MAT-sized products whose nnz(C) exceeds 2^31.

}  // namespace Impl
}  // namespace KokkosSparse
#endif
