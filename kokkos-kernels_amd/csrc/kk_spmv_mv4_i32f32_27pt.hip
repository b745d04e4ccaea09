// kk_spmv_mv4_i32f32_27pt.hip -- the plane-marching rank-2 kernel, int32_t offsets, float matrix values, stencil 27pt: a code object of its own (see kk_spmv_mv4.h)
#define KK_MV4_INSTANTIATE
#include "kk_spmv_mv4.h"
namespace kk {
template int launch_mv4_stencil<int32_t, float, 9, kMv4Pat27>(const kkamd_mv4_plan*, const kkamd_crs_t*, const double*, int64_t, int64_t, double*, int64_t, int64_t, double, double, hipStream_t, int, int, bool, bool);
}
