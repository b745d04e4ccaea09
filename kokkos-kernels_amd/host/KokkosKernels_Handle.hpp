// KokkosKernels::Experimental::KokkosKernelsHandle -- the slice the SpGEMM path uses
// (reference: sparse/src/KokkosKernels_Handle.hpp:37-66,281-343,385-482): create / get / destroy of the SpGEMM
// sub-handle plus the tuning setters.  set_verbose acts (the library prints the chosen algorithm, row bins and compression
// decision, like KOKKOSKERNELS_VERBOSE); the team / vector / shared-memory setters describe Kokkos TeamPolicy launches that
// do not exist here and throw std::runtime_error instead of being accepted and ignored.
#pragma once
#include "KokkosSparse_spgemm_handle.hpp"

namespace KokkosKernels { namespace Experimental {

template <class size_type_, class lno_t_, class scalar_t_, class ExecutionSpace, class TemporaryMemorySpace,
          class PersistentMemorySpace>
class KokkosKernelsHandle {
 public:
  using size_type        = std::remove_const_t<size_type_>;
  using nnz_lno_t        = std::remove_const_t<lno_t_>;
  using nnz_scalar_t     = std::remove_const_t<scalar_t_>;
  using const_size_type  = const size_type;
  using const_nnz_lno_t  = const nnz_lno_t;
  using const_nnz_scalar_t = const nnz_scalar_t;
  using HandleExecSpace  = ExecutionSpace;
  using SPGEMMHandleType = KokkosSparse::SPGEMMHandle<size_type, nnz_lno_t, nnz_scalar_t, ExecutionSpace,
                                                     TemporaryMemorySpace, PersistentMemorySpace>;
  KokkosKernelsHandle() = default;
  ~KokkosKernelsHandle() { destroy_spgemm_handle(); }
  KokkosKernelsHandle(const KokkosKernelsHandle&)            = delete;
  KokkosKernelsHandle& operator=(const KokkosKernelsHandle&) = delete;

  void create_spgemm_handle(KokkosSparse::SPGEMMAlgorithm algo = KokkosSparse::SPGEMM_DEFAULT) {
    destroy_spgemm_handle();
    spgemm_ = new SPGEMMHandleType(algo);
    if (verbose_) spgemm_->set_verbose(true);
  }
  SPGEMMHandleType* get_spgemm_handle() { return spgemm_; }
  void destroy_spgemm_handle() { delete spgemm_; spgemm_ = nullptr; }

  // tuning knobs of the reference (defaults: shmem 16128 B, dynamic scheduling, ...): no gfx950 equivalent -> they throw
  void set_team_work_size(int) { unsupported("team_work_size"); }
  void set_shmem_size(size_t) { unsupported("shmem_size"); }
  void set_suggested_team_size(int) { unsupported("suggested_team_size"); }
  void set_suggested_vector_size(int) { unsupported("suggested_vector_size"); }
  void set_dynamic_scheduling(bool) { unsupported("dynamic_scheduling"); }
  void set_verbose(bool v) { verbose_ = v; if (spgemm_) spgemm_->set_verbose(v); }
  bool get_verbose() const { return verbose_; }
 private:
  static void unsupported(const char* what) {
    throw std::runtime_error(std::string("KokkosKernelsHandle::set_") + what + ": tunes Kokkos team launches of the reference's kernels; "
                             "the gfx950 implementation has no counterpart (LDS tables and launch shapes follow the row bins)");
  }
  SPGEMMHandleType* spgemm_ = nullptr;
  bool verbose_             = false;
};

}}  // namespace KokkosKernels::Experimental
