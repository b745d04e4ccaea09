// SPGEMMAlgorithm / SPGEMMHandle -- reference: sparse/src/KokkosSparse_spgemm_handle.hpp:44-93,231-317,427-501,
// 562-694,749-784.  Cross-phase state (c_nnz, row flops, max nnz per row, phase flags) lives in the library's
// kkamd_spgemm_handle; this class forwards to it.  Option setters that choose something this implementation has ACT
// (algorithm, accumulator, compression, compression cut-off, verbose); the reference's tuning hints (hash scale, first-level
// cut-off, MKL options, read/write cost, compression steps, MaxColDenseAcc) are kept as the public members / getters the
// reference has, forwarded to the library as recorded hints and without effect, like in the reference's own TPL paths.
// SPGEMM_DEBUG / SPGEMM_SERIAL (host-sequential in the reference) run the device hash algorithm: the public spgemm_numeric
// sorts every algorithm's rows (impl/KokkosSparse_spgemm_numeric_spec.hpp:138-140), so C is the same matrix.
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
  bool are_rowflops_computed() { return is_symbolic_called(); }
  // option setters that act (:301-303,618-623)
  void set_compression(bool on) { compress_second_matrix = on; set("compression", on ? 1.0 : 0.0); }
  bool get_compression() { return compress_second_matrix; }
  void set_compression_cut_off(double c) { compression_cut_off = c; set("compression_cut_off", c); }
  double get_compression_cut_off() { return compression_cut_off; }
  void set_accumulator_type(const SPGEMMAccumulator& a) { accumulator_type = a; set("accumulator", (double)(int)a); }
  SPGEMMAccumulator get_accumulator_type() const { return accumulator_type; }
  void set_verbose(bool v) { set("verbose", v ? 1.0 : 0.0); }
  // hints (:295-317,364-366,684-694): remembered for the getters, recorded by the library, without effect
  void set_first_level_hash_cut_off(double v) { first_level_hash_cut_off = v; set("first_level_hash_cut_off", v); }
  double get_first_level_hash_cut_off() { return first_level_hash_cut_off; }
  void set_min_hash_size_scale(int v) { min_hash_size_scale = v; set("min_hash_size_scale", (double)v); }
  int get_min_hash_size_scale() { return min_hash_size_scale; }
  void set_read_write_cost_calc(bool v) { calculate_read_write_cost = v; set("read_write_cost_calc", v ? 1.0 : 0.0); }
  int get_read_write_cost_calc() { return calculate_read_write_cost; }
  void set_mkl_sort_option(int v) { mkl_sort_option = v; set("mkl_sort_option", (double)v); }
  int get_mkl_sort_option() { return mkl_sort_option; }
  void set_multi_color_scale(double v) { multi_color_scale = v; set("multi_color_scale", v); }
  double get_multi_color_scale() { return multi_color_scale; }
  void set_compression_steps(bool single) { is_compression_single_step = single; set("compression_steps", single ? 1.0 : 0.0); }
  bool get_compression_step() { return is_compression_single_step; }
  void set_sort_option(int o) { set("sort_option", (double)o); }               // rows of C always leave sorted
  // public data members the reference's driver assigns directly (perf_test/sparse/KokkosSparse_spgemm.cpp:381-388); defaults :463-470
  size_t MaxColDenseAcc      = 250001;
  bool mkl_keep_output       = true;
  bool mkl_convert_to_1base  = true;
  bool is_compression_single_step = false;
  bool is_compressed() { return query(6) != 0; }                 // the last symbolic phase ran on the compressed B
  void set(const char* key, double value) { Impl::kkamd_check(kkamd_spgemm_set(h_, key, value)); }
 private:
  int64_t query(int what) { int64_t v = 0; Impl::kkamd_check(kkamd_spgemm_get(h_, what, &v)); return v; }
  SPGEMMAlgorithm algorithm_type;
  SPGEMMAccumulator accumulator_type = SPGEMM_ACC_DEFAULT;
  bool compress_second_matrix   = false;   // this implementation's default: B compression measured slower on gfx950
  double compression_cut_off    = 0.85;
  double first_level_hash_cut_off = 0.50;
  int min_hash_size_scale       = 1;
  bool calculate_read_write_cost = false;
  int mkl_sort_option           = 7;
  double multi_color_scale      = 1.0;
  kkamd_spgemm_handle_t* h_ = nullptr;
};

}  // namespace KokkosSparse
