// KokkosSparse::spgemm_symbolic / spgemm_numeric (view level and matrix level) and spgemm<CMatrix> (no reuse).
// Reference: sparse/src/KokkosSparse_spgemm_symbolic.hpp:25-182, ..._spgemm_numeric.hpp:31-242,
// ..._spgemm.hpp:40-61,119-129,170-218.  Kept: argument order and meaning (A is m x n, B is n x k), transposes
// rejected, "handle must carry an SpGEMM sub-handle" (std::invalid_argument), matrix-level wrappers allocating
// row_map C before and entries/values after symbolic from get_c_nnz(), empty products short-circuited.
// Impl::SPGEMM_SYMBOLIC / SPGEMM_NUMERIC are replaced by the two C-ABI calls.
#pragma once
#include "KokkosKernels_Handle.hpp"
#include "KokkosSparse_CrsMatrix.hpp"

namespace KokkosSparse {

template <class KernelHandle, class ARow, class AEnt, class BRow, class BEnt, class CRow>
void spgemm_symbolic(KernelHandle* handle, typename KernelHandle::const_nnz_lno_t m, typename KernelHandle::const_nnz_lno_t n,
                     typename KernelHandle::const_nnz_lno_t k, ARow row_mapA, AEnt entriesA, bool transposeA, BRow row_mapB,
                     BEnt entriesB, bool transposeB, CRow row_mapC, bool /*computeRowptrs*/ = false) {
  static_assert(std::is_same<typename ARow::non_const_value_type, typename CRow::non_const_value_type>::value,
                "KokkosSparse::spgemm_symbolic: Output type of row map C must match the size_type of A.");
  if (transposeA || transposeB) throw std::runtime_error("KokkosSparse::spgemm_symbolic: transposing A or B is not yet supported");
  auto* sh = handle->get_spgemm_handle();
  if (!sh) throw std::invalid_argument("KokkosSparse::spgemm_symbolic: the given KernelHandle does not have an SpGEMM handle associated with it.");
  if ((size_t)row_mapC.extent(0) != (size_t)m + 1) throw std::runtime_error("KokkosSparse::spgemm_symbolic: row_mapC must have m + 1 entries");
  int64_t c_nnz = 0;
  Kokkos::Profiling::pushRegion("KokkosSparse::spgemm_symbolic[KKAMD]");
  Impl::kkamd_check(kkamd_spgemm_symbolic(sh->native(), m, n, k, row_mapA.data(), entriesA.data(), row_mapB.data(), entriesB.data(),
                                          (void*)row_mapC.data(), Impl::kkamd_offset<typename ARow::non_const_value_type>::value,
                                          &c_nnz, nullptr));
  Kokkos::Profiling::popRegion();
}

template <class KernelHandle, class ARow, class AEnt, class AVal, class BRow, class BEnt, class BVal, class CRow, class CEnt, class CVal>
void spgemm_numeric(KernelHandle* handle, typename KernelHandle::const_nnz_lno_t m, typename KernelHandle::const_nnz_lno_t n,
                    typename KernelHandle::const_nnz_lno_t k, ARow row_mapA, AEnt entriesA, AVal valuesA, bool transposeA,
                    BRow row_mapB, BEnt entriesB, BVal valuesB, bool transposeB, CRow row_mapC, CEnt entriesC, CVal valuesC) {
  if (transposeA || transposeB) throw std::runtime_error("KokkosSparse::spgemm_numeric: transposing A or B is not yet supported");
  auto* sh = handle->get_spgemm_handle();
  if (!sh) throw std::invalid_argument("KokkosSparse::spgemm_numeric: the given KernelHandle does not have an SpGEMM handle associated with it.");
  Kokkos::Profiling::pushRegion("KokkosSparse::spgemm_numeric[KKAMD]");
  Impl::kkamd_check(kkamd_spgemm_numeric(sh->native(), m, n, k, row_mapA.data(), entriesA.data(), valuesA.data(), row_mapB.data(),
                                         entriesB.data(), valuesB.data(), row_mapC.data(), (int32_t*)entriesC.data(),
                                         (void*)valuesC.data(), Impl::kkamd_offset<typename ARow::non_const_value_type>::value,
                                         Impl::kkamd_scalar<typename AVal::non_const_value_type>::value, nullptr));
  Kokkos::Profiling::popRegion();
}

template <class KernelHandle, class AMatrix, class BMatrix, class CMatrix>
void spgemm_symbolic(KernelHandle& kh, const AMatrix& A, const bool Amode, const BMatrix& B, const bool Bmode, CMatrix& C) {
  using row_map_type = typename CMatrix::row_map_type::non_const_type;
  using entries_type = typename CMatrix::index_type::non_const_type;
  using values_type  = typename CMatrix::values_type::non_const_type;
  row_map_type row_mapC(Kokkos::view_alloc(Kokkos::WithoutInitializing, "non_const_lnow_row"), A.numRows() + 1);
  entries_type entriesC;
  values_type valuesC;
  KokkosSparse::spgemm_symbolic(&kh, A.numRows(), B.numRows(), B.numCols(), A.graph.row_map, A.graph.entries, Amode,
                                B.graph.row_map, B.graph.entries, Bmode, row_mapC);
  const size_t c_nnz_size = kh.get_spgemm_handle()->get_c_nnz();
  if (c_nnz_size) {
    entriesC = entries_type(Kokkos::view_alloc(Kokkos::WithoutInitializing, "entriesC"), c_nnz_size);
    valuesC  = values_type(Kokkos::view_alloc(Kokkos::WithoutInitializing, "valuesC"), c_nnz_size);
  }
  C = CMatrix("C=AB", A.numRows(), B.numCols(), c_nnz_size, valuesC, row_mapC, entriesC);
}

template <class KernelHandle, class AMatrix, class BMatrix, class CMatrix>
void spgemm_numeric(KernelHandle& kh, const AMatrix& A, const bool Amode, const BMatrix& B, const bool Bmode, CMatrix& C) {
  KokkosSparse::spgemm_numeric(&kh, A.numRows(), B.numRows(), B.numCols(), A.graph.row_map, A.graph.entries, A.values, Amode,
                               B.graph.row_map, B.graph.entries, B.values, Bmode, C.graph.row_map, C.graph.entries, C.values);
}

template <class CMatrix, class AMatrix, class BMatrix>
CMatrix spgemm(const AMatrix& A, const bool Amode, const BMatrix& B, const bool Bmode) {
  if (Amode || Bmode) throw std::invalid_argument("KokkosSparse::spgemm: transposing A and/or B is not yet supported");
  if (A.numCols() != B.numRows())
    throw std::invalid_argument("KokkosSparse::spgemm: op(A) and op(B) have incompatible dimensions for multiplication");
  if constexpr (!std::is_void<typename CMatrix::memory_traits>::value) {
    if (CMatrix::memory_traits::is_unmanaged)
      throw std::invalid_argument("KokkosSparse::spgemm: C must not have the Unmanaged memory trait, because spgemm needs to allocate its Views");
  }
  if (!A.numRows() || !A.numCols() || !B.numCols() || !A.nnz() || !B.nnz()) {
    typename CMatrix::row_map_type::non_const_type row_mapC("C rowmap", A.numRows() + 1);
    typename CMatrix::index_type entriesC;
    typename CMatrix::values_type valuesC;
    return CMatrix("C", A.numRows(), B.numCols(), 0, valuesC, row_mapC, entriesC);
  }
  using KH = KokkosKernels::Experimental::KokkosKernelsHandle<typename CMatrix::non_const_size_type, typename CMatrix::non_const_ordinal_type,
                                                              typename CMatrix::non_const_value_type, typename CMatrix::execution_space,
                                                              typename CMatrix::memory_space, typename CMatrix::memory_space>;
  KH kh;
  kh.create_spgemm_handle();
  CMatrix C;
  spgemm_symbolic(kh, A, false, B, false, C);
  spgemm_numeric(kh, A, false, B, false, C);
  kh.destroy_spgemm_handle();
  return C;
}

}  // namespace KokkosSparse
