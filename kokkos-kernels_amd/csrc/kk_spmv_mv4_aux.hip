// kk_spmv_mv4_aux.hip -- the gather kernel for the rows the plane-marching kernel leaves over and the analysis kernels (verify, list) (a code object of its own: see kk_spmv_mv4.h)
#define KK_MV4_INSTANTIATE
#include "kk_spmv_mv4.h"
namespace kk {
#define KK_MV4_ROWS(OT, AT_) template int launch_mv4_rows<OT, AT_>(const kkamd_mv4_plan*, const kkamd_crs_t*, const double*, int64_t, int64_t, double*, int64_t, int64_t, double, double, hipStream_t, int, int);
KK_MV4_FOR_ALL(KK_MV4_ROWS)
#undef KK_MV4_ROWS
template int launch_mv4_verify<int32_t>(int64_t, const int32_t*, const int32_t*, Mv4Tab, Mv4Tab, int, int, int, int32_t*, uint32_t*, unsigned long long*, hipStream_t);
template int launch_mv4_verify<int64_t>(int64_t, const int64_t*, const int32_t*, Mv4Tab, Mv4Tab, int, int, int, int64_t*, uint32_t*, unsigned long long*, hipStream_t);
template int launch_mv4_list<int32_t>(int64_t, const int32_t*, int32_t*, unsigned long long*, hipStream_t);
template int launch_mv4_list<int64_t>(int64_t, const int64_t*, int32_t*, unsigned long long*, hipStream_t);
}
