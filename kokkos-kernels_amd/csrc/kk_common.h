// kk_common.h -- shared host/device helpers of libkkamd (status + error text, launch geometry,
// wave-level reductions).  gfx950 facts used here: 64-lane wavefronts, 256 CUs in 8 XCDs with the
// dispatcher placing workgroup b on XCD b % 8 (speed only -- never relied on for correctness).
#pragma once
#include "kk_rt.h"
#include "../../include/kkamd.h"
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <string>

namespace kk {

constexpr int kWave    = 64;
constexpr int kNumXcd  = 8;
constexpr int kBlock   = 256;   // 4 waves: one per SIMD of a CU

inline std::string& last_error_ref() { static thread_local std::string e; return e; }
inline int fail(int code, const char* fmt, ...) {
  char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  last_error_ref() = buf; return code;
}
#define KK_HIP(expr)                                                                            \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return kk::fail(KKAMD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define KK_LAUNCH_CHECK() KK_HIP(hipGetLastError())

inline hipStream_t to_hip(kkamd_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Logical work-item id for physical workgroup `pid` out of `nwg`: each XCD (own 4 MiB L2) walks one
// contiguous eighth of the logical range, so neighbouring logical tiles -- which share x-vector lines
// and B rows -- hit the same L2.  Bijective for any nwg (the first nwg % 8 XCDs take one extra).
__host__ __device__ __forceinline__ int64_t xcd_remap(int64_t pid, int64_t nwg) {
  const int64_t q = nwg / kNumXcd, rem = nwg % kNumXcd;
  const int64_t x = pid % kNumXcd, i = pid / kNumXcd;
  return x * q + (x < rem ? x : rem) + i;
}

// Grouped order: the launch is cut into blocks of 8*G consecutive logical tiles and XCD x takes G CONSECUTIVE tiles of
// every block (physical workgroups b, b+8, ..., which are dispatched close in time).  Neighbouring tiles -- which share
// x lines -- then meet in one L2, while all eight XCDs still sweep the same few-MB region of the matrix together (the DRAM
// page locality the fully contiguous order gives up).  G a power of two; the incomplete last block keeps dispatch order.
__host__ __device__ __forceinline__ int64_t xcd_group_order(int64_t pid, int64_t nwg, int G) {
  const int sh = 3 + __builtin_ctz((unsigned)G);              // shifts only: G and the XCD count are powers of two
  const int64_t blk = pid >> sh;
  if (((blk + 1) << sh) > nwg) return pid;
  const int within = (int)(pid & (((int64_t)1 << sh) - 1));
  return (blk << sh) + (int64_t)(within & (kNumXcd - 1)) * G + (within >> 3);
}
// mode 0: dispatch order, 1: XCD-contiguous, >= 2: grouped with G = mode
__host__ __device__ __forceinline__ int64_t xcd_order(int64_t pid, int64_t nwg, int mode) {
  return mode == 0 ? pid : (mode == 1 ? xcd_remap(pid, nwg) : xcd_group_order(pid, nwg, mode));
}

// sum across the `width` lanes (power of two, <= 64) of an aligned lane group; result in every lane.
template <class T> __device__ __forceinline__ T group_sum(T v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// a device buffer that frees itself (host-side temporaries of the analysis; error paths return early)
struct DevBuf {
  void* p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { reset(); }
  hipError_t alloc(size_t bytes) { reset(); return hipMalloc(&p, bytes ? bytes : 1); }
  void reset() { if (p) { (void)hipFree(p); p = nullptr; } }
  void* release() { void* q = p; p = nullptr; return q; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Profiling ranges with the reference's label strings (Kokkos::Profiling::pushRegion at sparse/src/KokkosSparse_spmv.hpp:261-266,
// sparse/tpls/KokkosSparse_spmv_tpl_spec_decl.hpp:411-414, ...spgemm_symbolic_tpl_spec_decl.hpp:343-344): emitted as roctx
// ranges, so rocprofv3 --marker-trace shows "KokkosSparse::spmv[TPL_KKAMD,double]" around the kernels.  libroctx64 is
// looked up at run time; without it the ranges cost one predictable branch.
void trace_push(const char* label);
void trace_pop();
struct TraceRange {
  explicit TraceRange(const char* label) { trace_push(label); }
  ~TraceRange() { trace_pop(); }
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;
};
// KokkosKernelsHandle::set_verbose / KOKKOSKERNELS_VERBOSE: what was chosen (kernels, bins, tile modes) and phase times, on stdout
extern int g_verbose;

// argument checks shared by the SpMV entry points (kk_spmv.hip)
int check_crs(const kkamd_crs_t* A);
int parse_mode(char mode, bool* trans);

// knobs whose key starts with "spgemm_" (kk_spgemm.hip); reached through kkamd_set_default
int spgemm_set_default(const char* key, int value);
// kk_spmv_struct.hip: 1 = XCD-contiguous workgroup order in the interior kernel (knob "struct_remap")
extern int g_struct_remap;
extern int g_struct_strip;        // kkamd_spmv_struct: lines per XCD strip of the strip order (0 = off)
extern int g_struct_group;        // kkamd_spmv_struct: grouped XCD order of the interior workgroups (0 = dispatch order)
extern int g_struct_lds_pad_kb;  // measurement aid: extra dynamic LDS per interior workgroup (lowers occupancy)

template <class T> struct scalar_tag;
template <> struct scalar_tag<float>  { static constexpr int value = KKAMD_F32; };
template <> struct scalar_tag<double> { static constexpr int value = KKAMD_F64; };

}  // namespace kk
