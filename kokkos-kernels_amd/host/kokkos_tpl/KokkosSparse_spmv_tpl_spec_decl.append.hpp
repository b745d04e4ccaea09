// APPEND to sparse/tpls/KokkosSparse_spmv_tpl_spec_decl.hpp (after its closing #endif).
// Impl::SPMV<..., tpl_spec_avail = true> for the tuples of KokkosSparse_spmv_tpl_spec_avail.append.hpp: unwraps the Views to
// raw device pointers + extents + the execution space's HIP stream and calls kkamd_spmv (include/kkamd.h).  The per-matrix
// analysis lives where the vendor wrappers keep theirs: a TPL_SpMV_Data subclass in handle->tpl_rank1
// (sparse/src/KokkosSparse_spmv_handle.hpp:76-107,241); set_exec_space fences the old stream when the handle moves.
#ifndef KOKKOSPARSE_SPMV_TPL_SPEC_DECL_KKAMD_HPP_
#define KOKKOSPARSE_SPMV_TPL_SPEC_DECL_KKAMD_HPP_
#ifdef KOKKOSKERNELS_ENABLE_TPL_KKAMD
#include <kkamd.h>
#include <stdexcept>
#include <string>

namespace KokkosSparse {
namespace Impl {

struct KKAMD_CRS_SpMV_Data : public TPL_SpMV_Data<Kokkos::HIP> {
  KKAMD_CRS_SpMV_Data(const Kokkos::HIP& exec_) : TPL_SpMV_Data<Kokkos::HIP>(exec_) {}
  ~KKAMD_CRS_SpMV_Data() { kkamd_spmv_plan_destroy(plan); }
  kkamd_spmv_plan_t* plan = nullptr;
};

inline void kkamd_safe_call(int status) {   // the *_SAFE_CALL convention of the vendor wrappers: status -> exception
  if (status != KKAMD_OK) {
    if (status == KKAMD_ERR_STATE) throw std::invalid_argument(kkamd_last_error());
    throw std::runtime_error(std::string("kkamd: ") + kkamd_last_error());
  }
}

template <class AMatrix>
inline kkamd_crs_t kkamd_crs_desc(const AMatrix& A) {
  kkamd_crs_t d;
  d.num_rows = A.numRows(); d.num_cols = A.numCols(); d.nnz = (int64_t)A.nnz();
  d.d_row_map = A.graph.row_map.data(); d.d_entries = A.graph.entries.data(); d.d_values = A.values.data();
  d.offset_type = sizeof(typename AMatrix::non_const_size_type) == 8 ? KKAMD_I64 : KKAMD_I32;
  d.value_type  = std::is_same<typename AMatrix::non_const_value_type, double>::value ? KKAMD_F64 : KKAMD_F32;
  return d;
}

// the sub-handle of rank R (tpl_rank1 / tpl_rank2): created with the plan on first use, re-bound to the stream afterwards
template <class Handle>
inline KKAMD_CRS_SpMV_Data* kkamd_subhandle(const Kokkos::HIP& exec, Handle* handle, TPL_SpMV_Data<Kokkos::HIP>*& slot,
                                            const kkamd_crs_t& d) {
  KKAMD_CRS_SpMV_Data* sub;
  if (slot) {
    sub = dynamic_cast<KKAMD_CRS_SpMV_Data*>(slot);
    if (!sub) throw std::runtime_error("KokkosSparse::spmv: subhandle is not set up for kkamd CRS");
    sub->set_exec_space(exec);
  } else {
    sub  = new KKAMD_CRS_SpMV_Data(exec);
    slot = sub;
    if (handle->get_algorithm() != SPMV_FAST_SETUP)
      kkamd_safe_call(kkamd_spmv_plan_create(&sub->plan, &d, (int)handle->get_algorithm(),
                                             reinterpret_cast<kkamd_stream_t>(exec.hip_stream())));
  }
  return sub;
}

template <class Handle, class AMatrix, class XVector, class YVector>
void spmv_kkamd(const Kokkos::HIP& exec, Handle* handle, const char mode[], typename YVector::const_value_type& alpha,
                const AMatrix& A, const XVector& x, typename YVector::const_value_type& beta, const YVector& y) {
  const kkamd_crs_t d      = kkamd_crs_desc(A);
  KKAMD_CRS_SpMV_Data* sub = kkamd_subhandle(exec, handle, handle->tpl_rank1, d);
  kkamd_safe_call(kkamd_spmv(sub->plan, &d, mode[0], (double)alpha, x.data(), (double)beta, y.data(),
                             std::is_same<typename YVector::non_const_value_type, double>::value ? KKAMD_F64 : KKAMD_F32,
                             reinterpret_cast<kkamd_stream_t>(exec.hip_stream())));
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

#ifndef KOKKOSKERNELS_ENABLE_TPL_ROCSPARSE
KOKKOSSPARSE_SPMV_KKAMD(double, int, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_KKAMD(double, int, Kokkos::LayoutRight)
KOKKOSSPARSE_SPMV_KKAMD(float, int, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_KKAMD(float, int, Kokkos::LayoutRight)
#endif
KOKKOSSPARSE_SPMV_KKAMD(double, size_t, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_KKAMD(double, size_t, Kokkos::LayoutRight)
KOKKOSSPARSE_SPMV_KKAMD(float, size_t, Kokkos::LayoutLeft)
KOKKOSSPARSE_SPMV_KKAMD(float, size_t, Kokkos::LayoutRight)
#undef KOKKOSSPARSE_SPMV_KKAMD

}  // namespace Impl
}  // namespace KokkosSparse
#endif  // KOKKOSKERNELS_ENABLE_TPL_KKAMD
#endif  // KOKKOSPARSE_SPMV_TPL_SPEC_DECL_KKAMD_HPP_
