// KokkosSparse::SPMVAlgorithm / SPMVHandle -- reference: sparse/src/KokkosSparse_spmv_handle.hpp:32-86,217-349.
// The per-matrix analysis the reference keeps in tpl_rank1 / tpl_rank2 (:241-242) is a kkamd_spmv_plan here:
// created lazily by the first spmv call, bound to one matrix for life (:273-277), released with the handle.
#pragma once
#include "Kokkos_Shim.hpp"
#include "kkamd_status.hpp"

namespace KokkosSparse {

enum SPMVAlgorithm {
  SPMV_DEFAULT, SPMV_FAST_SETUP, SPMV_NATIVE, SPMV_MERGE_PATH, SPMV_NATIVE_MERGE_PATH,
  SPMV_BSR_V41, SPMV_BSR_V42, SPMV_BSR_TC   // BsrMatrix only: rejected for CrsMatrix as in the reference (:319-327)
};

inline const char* get_spmv_algorithm_name(SPMVAlgorithm a) {
  switch (a) {
    case SPMV_DEFAULT: return "SPMV_DEFAULT";
    case SPMV_FAST_SETUP: return "SPMV_FAST_SETUP";
    case SPMV_NATIVE: return "SPMV_NATIVE";
    case SPMV_MERGE_PATH: return "SPMV_MERGE_PATH";
    case SPMV_NATIVE_MERGE_PATH: return "SPMV_NATIVE_MERGE_PATH";
    case SPMV_BSR_V41: return "SPMV_BSR_V41";
    case SPMV_BSR_V42: return "SPMV_BSR_V42";
    case SPMV_BSR_TC: return "SPMV_BSR_TC";
  }
  throw std::invalid_argument("SPMVHandle::get_algorithm_name: unknown algorithm");
}
inline bool is_spmv_algorithm_native(SPMVAlgorithm a) {
  switch (a) {
    case SPMV_NATIVE: case SPMV_NATIVE_MERGE_PATH: case SPMV_BSR_V41: case SPMV_BSR_V42: case SPMV_BSR_TC: return true;
    default: return false;
  }
}

namespace Impl {
template <class ExecutionSpace, class MemorySpace, class Scalar, class Offset, class Ordinal>
struct SPMVHandleImpl {
  using ExecutionSpaceType = ExecutionSpace;
  using ImplType           = SPMVHandleImpl;
  explicit SPMVHandleImpl(SPMVAlgorithm algo_) : algo(algo_) {}
  ~SPMVHandleImpl() { if (plan) kkamd_spmv_plan_destroy(plan); }
  SPMVHandleImpl(const SPMVHandleImpl&)            = delete;
  SPMVHandleImpl& operator=(const SPMVHandleImpl&) = delete;
  ImplType* get_impl() { return this; }
  SPMVAlgorithm get_algorithm() const { return algo; }
  const SPMVAlgorithm algo = SPMV_DEFAULT;
  kkamd_spmv_plan_t* plan  = nullptr;     // the reference's tpl_rank1 / tpl_rank2
  // Expert knobs: the reference's public members (:243-252), assignable directly as there.  vector_length has a meaning
  // here: lanes per row of the no-analysis row kernel (knob "lanes_per_row"; -1 = automatic from nnz / row, like the
  // reference).  team_size, rows_per_thread and force_static / force_dynamic_schedule shape Kokkos TeamPolicy / RangePolicy
  // launches of the reference's native kernels; a TPL path of the reference (rocSPARSE, cuSPARSE) never reads them, and
  // neither do the gfx950 kernels: they are kept so that code which sets them compiles and runs unchanged.
  int team_size               = -1;
  int64_t rows_per_thread     = -1;
  bool force_static_schedule  = false;
  bool force_dynamic_schedule = false;
  int vector_length = -1;
  // any plan knob of include/kkamd.h by name (applied when the plan is created, or to the live plan)
  void set_knob(const char* key, int value) {
    if (plan) KokkosSparse::Impl::kkamd_check(kkamd_spmv_plan_set(plan, key, value));
    else pending_.emplace_back(key, value);
  }
  // the caller wrote to A.values: re-ordered copies the plan keeps (cached transpose of modes T / H, column-slab copy) copy them again
  // at the next call (kkamd_spmv_plan_values_changed; required under the knob values_tracking = 1, optional otherwise)
  void values_changed() { if (plan) KokkosSparse::Impl::kkamd_check(kkamd_spmv_plan_values_changed(plan)); }
  std::vector<std::pair<std::string, int>> pending_;
};
}  // namespace Impl

template <class DeviceType, class AMatrix, class XVector, class YVector>
struct SPMVHandle : public Impl::SPMVHandleImpl<typename DeviceType::execution_space, typename AMatrix::memory_space,
                                                typename AMatrix::non_const_value_type, typename AMatrix::non_const_size_type,
                                                typename AMatrix::non_const_ordinal_type> {
  using ImplType = Impl::SPMVHandleImpl<typename DeviceType::execution_space, typename AMatrix::memory_space,
                                        typename AMatrix::non_const_value_type, typename AMatrix::non_const_size_type,
                                        typename AMatrix::non_const_ordinal_type>;
  using AMatrixType        = AMatrix;
  using XVectorType        = XVector;
  using YVectorType        = YVector;
  using ExecutionSpaceType = typename DeviceType::execution_space;
  static_assert(XVector::rank() == YVector::rank(), "SPMVHandle: ranks of XVector and YVector must match.");

  explicit SPMVHandle(SPMVAlgorithm algo_ = SPMV_DEFAULT) : ImplType(algo_) {
    switch (algo_) {
      case SPMV_BSR_V41: case SPMV_BSR_V42: case SPMV_BSR_TC:
        throw std::invalid_argument(std::string("SPMVHandle: algorithm ") + get_spmv_algorithm_name(algo_) +
                                    " cannot be used if A is a CrsMatrix");
      default:;
    }
  }
  SPMVAlgorithm get_algorithm() const { return this->algo; }
  ImplType* get_impl() { return static_cast<ImplType*>(this); }
};

namespace Impl {
template <class> struct is_spmv_handle : std::false_type {};
template <class... P> struct is_spmv_handle<SPMVHandle<P...>> : std::true_type {};
}  // namespace Impl

}  // namespace KokkosSparse
