// kk_spmv_mv4_i64f32_gen9.hip -- the plane-marching rank-2 kernel, int64_t offsets, float matrix values, stencil gen9: a code object of its own (see kk_spmv_mv4.h)
#define KK_MV4_INSTANTIATE
#include "kk_spmv_mv4.h"
namespace kk {
template int launch_mv4_stencil<int64_t, float, 9, 0u>(const kkamd_mv4_plan*, const kkamd_crs_t*, const double*, int64_t, int64_t, double*, int64_t, int64_t, double, double, hipStream_t, int, int, bool, bool);
}
