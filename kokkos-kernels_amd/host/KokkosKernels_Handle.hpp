// KokkosKernels::Experimental::KokkosKernelsHandle -- the slice the SpGEMM path uses
// (reference: sparse/src/KokkosKernels_Handle.hpp:37-66,281-343,380-482): create / get / destroy of the SpGEMM
// sub-handle plus the tuning setters.  set_verbose acts (the library prints the chosen algorithm, row bins and compression
// decision, like KOKKOSKERNELS_VERBOSE).  The team / vector / shared-memory / scheduling setters are HINTS in the reference
// too -- its rocSPARSE and cuSPARSE paths take and ignore them, and its driver and unit tests set them before every spgemm
// (perf_test/sparse/KokkosSparse_spgemm.cpp:311-317, sparse/unit_test/Test_Sparse_spgemm.hpp:91-92) -- so they are accepted,
// remembered (the get_set_* getters of the reference return them), forwarded to the library as recorded hints (reported
// under verbose) and have no effect: launch shapes and LDS tables follow the row bins.
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
    // hints given before the sub-handle existed (the reference's order: setters first, then create_spgemm_handle)
    if (team_work_size != -1) spgemm_->set("team_work_size", team_work_size);
    if (shared_memory_size != 16128) spgemm_->set("shmem_size", (double)shared_memory_size);
    if (suggested_team_size != -1) spgemm_->set("suggested_team_size", suggested_team_size);
    if (vector_size != -1) spgemm_->set("suggested_vector_size", vector_size);
    if (!use_dynamic_scheduling) spgemm_->set("dynamic_scheduling", 0);
  }
  SPGEMMHandleType* get_spgemm_handle() { return spgemm_; }
  void destroy_spgemm_handle() { delete spgemm_; spgemm_ = nullptr; }

  // hints of the reference (:380-465; defaults :203-215): remembered, forwarded, without effect on gfx950
  void set_team_work_size(const int v) { team_work_size = v; hint("team_work_size", v); }
  int get_set_team_work_size() { return team_work_size; }
  int get_team_work_size(const int team_size, const int /*concurrency*/, const nnz_lno_t /*overall_work_size*/) {
    return team_work_size != -1 ? team_work_size : team_size;                      // the reference's answer for Exec_HIP (:393-404)
  }
  void set_shmem_size(const size_t v) { shared_memory_size = v; hint("shmem_size", (double)v); }
  size_t get_shmem_size() { return shared_memory_size; }
  void set_suggested_team_size(const int v) { suggested_team_size = v; hint("suggested_team_size", v); }
  int get_set_suggested_team_size() { return suggested_team_size; }
  void set_suggested_vector_size(int v) { vector_size = v; hint("suggested_vector_size", v); }
  int get_set_suggested_vector_size() { return vector_size; }
  void set_dynamic_scheduling(const bool v) { use_dynamic_scheduling = v; hint("dynamic_scheduling", v ? 1 : 0); }
  bool is_dynamic_scheduling() { return use_dynamic_scheduling; }
  void set_verbose(bool v) { verbose_ = v; if (spgemm_) spgemm_->set_verbose(v); }
  bool get_verbose() const { return verbose_; }
 private:
  void hint(const char* key, double v) { if (spgemm_) spgemm_->set(key, v); }
  int team_work_size        = -1;        // the reference's defaults (:203-215)
  size_t shared_memory_size = 16128;
  int suggested_team_size   = -1;
  int vector_size           = -1;
  bool use_dynamic_scheduling = true;
  SPGEMMHandleType* spgemm_ = nullptr;
  bool verbose_             = false;
};

}}  // namespace KokkosKernels::Experimental
