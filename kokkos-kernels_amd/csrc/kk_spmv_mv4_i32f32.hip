// kk_spmv_mv4_i32f32.hip -- the plane-marching rank-2 kernel for int32_t offsets and float matrix values: a code object of its own (see kk_spmv_mv4.h)
#include "kk_spmv_mv4.h"
namespace kk {
template int launch_mv4<int32_t, float>(const kkamd_spmv_plan*, const kkamd_crs_t*, const double*, int64_t, int64_t, double*, int64_t, int64_t, double, double, hipStream_t, int, int);
}
