// KokkosKernels::Experimental::KokkosKernelsHandle -- the slice the SpGEMM path uses
// (reference: sparse/src/KokkosKernels_Handle.hpp:37-66,281-343,385-482): create / get / destroy of the SpGEMM
// sub-handle plus the tuning setters, accepted for source compatibility.
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
  }
  SPGEMMHandleType* get_spgemm_handle() { return spgemm_; }
  void destroy_spgemm_handle() { delete spgemm_; spgemm_ = nullptr; }

  // tuning knobs of the reference (defaults: shmem 16128 B, dynamic scheduling, ...); no gfx950 equivalent
  void set_team_work_size(int) {}
  void set_shmem_size(size_t) {}
  void set_suggested_team_size(int) {}
  void set_suggested_vector_size(int) {}
  void set_dynamic_scheduling(bool) {}
  void set_verbose(bool v) { verbose_ = v; }
  bool get_verbose() const { return verbose_; }
 private:
  SPGEMMHandleType* spgemm_ = nullptr;
  bool verbose_             = false;
};

}}  // namespace KokkosKernels::Experimental
