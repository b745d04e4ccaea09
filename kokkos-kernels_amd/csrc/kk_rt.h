// kk_rt.h -- the one place the kernels' execution vocabulary comes from.
// Product build (hipcc --offload-arch=gfx950): the HIP runtime.  There is no CPU path.
// -DKK_EMU (tests/emu only): the fiber-based SIMT emulator used by `-m "not gpu"` logic tests.
#pragma once
#ifdef KK_EMU
#include "kk_emu.h"
#define KK_LAUNCH(kernel, grid, block, smem, stream, ...)                                   \
  do {                                                                                     \
    if (std::getenv("KK_EMU_TRACE")) std::fprintf(stderr, "kk_emu: launch %s\n", #kernel); \
    kk_emu::launch(dim3(grid), dim3(block), (size_t)(smem), [=]() { kernel(__VA_ARGS__); }); \
  } while (0)
#define KK_NT_LOAD(p) (*(p))
#define KK_NT_STORE(p, v) (*(p) = (v))
#define KK_DYN_SMEM(T, name) T* name = reinterpret_cast<T*>(kk_emu::S().dyn_smem)
#define KK_DEVICE_ONLY(...)
#define KK_UNROLL
#define KK_UNROLL4
#define KK_NOUNROLL
#define KK_WAVE_SYNC() kk_emu::sync_wave()
#define KK_QUAD_PERM(v, ctrl) kk_emu::quad_perm((v), (ctrl))
#define KK_UMUL24(a, b) ((unsigned)(a) * (unsigned)(b))
#define KK_GLDS16(gsrc, lds_wave_base, lane) std::memcpy((char*)(lds_wave_base) + 16 * (lane), (const void*)(gsrc), 16)
#define KK_GLDS_WAIT()
// v_mfma_f64_16x16x4f64 under the emulator: the operand layout probed on gfx950 (tools/probes/probe_mfma_f64.hip)
#define KK_MFMA_F64_16X16X4(a, b, c) kk_emu::mfma_f64_16x16x4((a), (b), (c))
#define KK_UNIFORM(v) (v)
#else
#include <hip/hip_runtime.h>
#define KK_LAUNCH(kernel, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (size_t)(smem), stream, __VA_ARGS__)
#define KK_NT_LOAD(p) __builtin_nontemporal_load(p)
#define KK_NT_STORE(p, v) __builtin_nontemporal_store((v), (p))
#define KK_DYN_SMEM(T, name)                                            \
  extern __shared__ __attribute__((aligned(16))) char kk_dyn_smem_[];   \
  T* name = reinterpret_cast<T*>(kk_dyn_smem_)
#define KK_DEVICE_ONLY(...) __VA_ARGS__
#define KK_UNROLL _Pragma("unroll")
#define KK_UNROLL4 _Pragma("unroll 4")
#define KK_NOUNROLL _Pragma("nounroll")
#define KK_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
// DPP quad permute of a 32-bit value: lane (4q+j) receives the value of lane 4q + ((ctrl >> 2j) & 3)
#define KK_QUAD_PERM(v, ctrl) __builtin_amdgcn_mov_dpp((v), (ctrl), 0xf, 0xf, true)
// full-rate 24-bit multiply (v_mul_u32_u24); the 32-bit v_mul_lo_u32 is quarter rate
#define KK_UMUL24(a, b) __umul24((a), (b))
// Asynchronous 16-byte global -> LDS copy (global_load_lds_dwordx4, gfx950): every lane names its own source; the destination
// is the wave-uniform LDS address `lds_wave_base` plus 16 * lane.  hipcc does not track the copy: KK_GLDS_WAIT (s_waitcnt 0)
// must precede the barrier that publishes the data.
#define KK_GLDS16(gsrc, lds_wave_base, lane)                                                                       \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),                          \
                                   (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0)
#define KK_GLDS_WAIT() __builtin_amdgcn_s_waitcnt(0)
// D(16x16) += A(16x4) B(4x16) on the matrix core, fp64.  Lane l holds A[l % 16][l / 16], B[l / 16][l % 16] and, in register r of
// the accumulator, D[4 r + l / 16][l % 16] (layout probed on the hardware: tools/probes/probe_mfma_f64.hip, profiles/round2)
#define KK_MFMA_F64_16X16X4(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
// a 32-bit value the program knows to be the same in every lane of the wave, moved to a scalar register
#define KK_UNIFORM(v) __builtin_amdgcn_readfirstlane(v)
#endif
typedef double kk_f64x4 __attribute__((vector_size(32)));
