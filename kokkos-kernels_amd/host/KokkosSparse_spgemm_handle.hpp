// SPGEMMAlgorithm / SPGEMMHandle -- reference: sparse/src/KokkosSparse_spgemm_handle.hpp:44-93,231-247,427-501,
// 562-616,749-784.  Cross-phase state (c_nnz, row flops, max nnz per row, phase flags) lives in the library's
// kkamd_spgemm_handle; this class forwards to it.  Every option setter either ACTS (algorithm, accumulator, compression,
// compression cut-off, verbose) or THROWS std::runtime_error (knobs of the reference's Kokkos team launches and two-level
// hash tables, and the host-sequential SPGEMM_DEBUG / SPGEMM_SERIAL, which have no counterpart here): nothing is swallowed.
#pragma once
#include <algorithm>
#include <string>
#include "Kokkos_Shim.hpp"
#include "kkamd_status.hpp"

namespace KokkosSparse {

enum SPGEMMAlgorithm {
  SPGEMM_KK, SPGEMM_KK_DENSE, SPGEMM_KK_MEMORY, SPGEMM_KK_LP,
  SPGEMM_DEFAULT, SPGEMM_DEBUG, SPGEMM_SERIAL, SPGEMM_KK_SPEED, SPGEMM_KK_MEMSPEED
};
enum SPGEMMAccumulator { SPGEMM_ACC_DEFAULT, SPGEMM_ACC_DENSE, SPGEMM_ACC_SPARSE };

inline SPGEMMAlgorithm StringToSPGEMMAlgorithm(std::string& name) {
  if (name == "SPGEMM_DEFAULT") return SPGEMM_KK;
  if (name == "SPGEMM_KK" || name == "KKSPGEMM") return SPGEMM_KK;
  if (name == "SPGEMM_KK_MEMORY" || name == "KKMEM") return SPGEMM_KK_MEMORY;
  if (name == "SPGEMM_KK_DENSE" || name == "KKDENSE") return SPGEMM_KK_DENSE;
  if (name == "SPGEMM_KK_LP" || name == "KKLP") return SPGEMM_KK_LP;
  if (name == "SPGEMM_KK_MEMSPEED" || name == "KKMEMSPEED") return SPGEMM_KK_MEMSPEED;
  if (name == "SPGEMM_KK_SPEED" || name == "KKSPEED") return SPGEMM_KK_SPEED;
  if (name == "SPGEMM_DEBUG" || name == "SPGEMM_SERIAL" || name == "KKDEBUG") return SPGEMM_SERIAL;
  throw std::runtime_error("Invalid SPGEMMAlgorithm name");
}

template <class size_type_, class lno_t_, class scalar_t_, class ExecutionSpace, class TemporaryMemorySpace,
          class PersistentMemorySpace>
class SPGEMMHandle {
 public:
  using size_type    = std::remove_const_t<size_type_>;
  using nnz_lno_t    = std::remove_const_t<lno_t_>;
  using nnz_scalar_t = std::remove_const_t<scalar_t_>;
  explicit SPGEMMHandle(SPGEMMAlgorithm a = SPGEMM_DEFAULT) : algorithm_type(a) {
    Impl::kkamd_check(kkamd_spgemm_create(&h_));
    try { set("algorithm", (double)(int)a); } catch (...) { kkamd_spgemm_destroy(h_); h_ = nullptr; throw; }
  }
  ~SPGEMMHandle() { if (h_) kkamd_spgemm_destroy(h_); }
  SPGEMMHandle(const SPGEMMHandle&)            = delete;
  SPGEMMHandle& operator=(const SPGEMMHandle&) = delete;

  kkamd_spgemm_handle_t* native() const { return h_; }
  SPGEMMAlgorithm get_algorithm_type() const { return algorithm_type; }
  void set_algorithm_type(const SPGEMMAlgorithm& a) { set("algorithm", (double)(int)a); algorithm_type = a; }
  size_type get_c_nnz() { return (size_type)query(0); }
  int64_t get_mults() { return query(1); }                       // the reference's original_overall_flops / 2
  nnz_lno_t get_max_result_nnz() { return (nnz_lno_t)query(3); }
  bool is_symbolic_called() { return query(4) != 0; }
  bool is_numeric_called() { return query(5) != 0; }
  bool are_rowptrs_computed() { return is_symbolic_called(); }
  bool are_entries_computed() { return is_numeric_called(); }
  // option setters (:295-306,618-623): forwarded to kkamd_spgemm_set -- they act or throw, see the header comment
  void set_compression(bool on) { set("compression", on ? 1.0 : 0.0); }
  void set_compression_cut_off(double c) { set("compression_cut_off", c); }
  void set_accumulator_type(const SPGEMMAccumulator& a) { set("accumulator", (double)(int)a); }
  void set_sort_option(int o) { set("sort_option", (double)o); }
  void set_min_hash_size_scale(int v) { set("min_hash_size_scale", (double)v); }
  void set_first_level_hash_cut_off(double v) { set("first_level_hash_cut_off", v); }
  void set_verbose(bool v) { set("verbose", v ? 1.0 : 0.0); }
  bool is_compressed() { return query(6) != 0; }                 // the last symbolic phase ran on the compressed B
  void set(const char* key, double value) { Impl::kkamd_check(kkamd_spgemm_set(h_, key, value)); }
 private:
  int64_t query(int what) { int64_t v = 0; Impl::kkamd_check(kkamd_spgemm_get(h_, what, &v)); return v; }
  SPGEMMAlgorithm algorithm_type;
  kkamd_spgemm_handle_t* h_ = nullptr;
};

}  // namespace KokkosSparse
