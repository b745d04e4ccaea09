// TEST INFRASTRUCTURE (tests/test_tpl_binding.py): parsed with g++ -fsyntax-only against the REFERENCE's own
// sparse/impl/KokkosSparse_spmv_spec.hpp, with the reference's sparse/tpls/KokkosSparse_spmv{,_mv}_tpl_spec_{avail,decl}.hpp
// replaced by copies to which kokkos-kernels_amd/host/kokkos_tpl/*.append.hpp were appended (the integration step of
// INTEGRATION.md).  Builds the internal types exactly as the public KokkosSparse::spmv does
// (sparse/src/KokkosSparse_spmv.hpp:155-196) from user-level types and asserts that the unification layer picks the KKAMD
// specialisation for each of them.
#include <KokkosSparse_spmv_spec.hpp>

namespace {
using device = Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>;

template <class Scalar, class Offset, class XLayout, class YLayout, int Rank>
struct Tuple {
  using AMatrix  = KokkosSparse::CrsMatrix<Scalar, int, device, void, Offset>;                   // what a user declares
  using XVector  = std::conditional_t<Rank == 1, Kokkos::View<Scalar*, XLayout, device>, Kokkos::View<Scalar**, XLayout, device>>;
  using YVector  = std::conditional_t<Rank == 1, Kokkos::View<Scalar*, YLayout, device>, Kokkos::View<Scalar**, YLayout, device>>;
  using Handle   = KokkosSparse::SPMVHandle<device, AMatrix, XVector, YVector>;
  using HandleImpl = typename Handle::ImplType;
  // sparse/src/KokkosSparse_spmv.hpp:158-196
  using AMatrix_Internal = KokkosSparse::CrsMatrix<typename AMatrix::const_value_type, typename AMatrix::const_ordinal_type,
                                                   typename AMatrix::device_type, Kokkos::MemoryTraits<Kokkos::Unmanaged>,
                                                   typename AMatrix::const_size_type>;
  using XVector_Internal = Kokkos::View<typename XVector::const_data_type, typename XVector::array_layout, typename XVector::device_type,
                                        Kokkos::MemoryTraits<Kokkos::Unmanaged | Kokkos::RandomAccess>>;
  using YVector_Internal = Kokkos::View<typename YVector::non_const_data_type, typename YVector::array_layout, typename YVector::device_type,
                                        Kokkos::MemoryTraits<Kokkos::Unmanaged>>;
  static constexpr bool rank1_avail =
      KokkosSparse::Impl::spmv_tpl_spec_avail<Kokkos::HIP, HandleImpl, AMatrix_Internal, XVector_Internal, YVector_Internal>::value;
  static constexpr bool rank2_avail =
      KokkosSparse::Impl::spmv_mv_tpl_spec_avail<Kokkos::HIP, HandleImpl, AMatrix_Internal, XVector_Internal, YVector_Internal>::value;
  static void call() {
    Kokkos::HIP exec;
    Handle handle;
    AMatrix_Internal A;
    XVector_Internal x;
    YVector_Internal y;
    // the specialisations (not the primary templates) carry the member aliases Handle / AMatrix / XVector / YVector
    using Spec = std::conditional_t<Rank == 1, KokkosSparse::Impl::SPMV<Kokkos::HIP, HandleImpl, AMatrix_Internal, XVector_Internal, YVector_Internal>,
                                    KokkosSparse::Impl::SPMV_MV<Kokkos::HIP, HandleImpl, AMatrix_Internal, XVector_Internal, YVector_Internal>>;
    static_assert(std::is_same<typename Spec::Handle, HandleImpl>::value && std::is_same<typename Spec::AMatrix, AMatrix_Internal>::value &&
                  std::is_same<typename Spec::XVector, XVector_Internal>::value && std::is_same<typename Spec::YVector, YVector_Internal>::value,
                  "the KKAMD specialisation was not selected");
    if constexpr (Rank == 1)
      KokkosSparse::Impl::SPMV<Kokkos::HIP, HandleImpl, AMatrix_Internal, XVector_Internal, YVector_Internal>::spmv(
          exec, handle.get_impl(), "N", Scalar(1), A, x, Scalar(0), y);
    else
      KokkosSparse::Impl::SPMV_MV<Kokkos::HIP, HandleImpl, AMatrix_Internal, XVector_Internal, YVector_Internal>::spmv_mv(
          exec, handle.get_impl(), "N", Scalar(1), A, x, Scalar(0), y);
  }
};

using L = Kokkos::LayoutLeft;
using R = Kokkos::LayoutRight;
#define CHECK_RANK1(S, O, LAY) static_assert(Tuple<S, O, LAY, LAY, 1>::rank1_avail, "rank-1 tuple not bound: " #S " " #O " " #LAY); \
  template struct Tuple<S, O, LAY, LAY, 1>;
#define CHECK_RANK2(S, O, XL, YL) static_assert(Tuple<S, O, XL, YL, 2>::rank2_avail, "rank-2 tuple not bound: " #S " " #O " " #XL " " #YL); \
  template struct Tuple<S, O, XL, YL, 2>;
#define CHECK_ALL(S, O) CHECK_RANK1(S, O, L) CHECK_RANK1(S, O, R) CHECK_RANK2(S, O, L, L) CHECK_RANK2(S, O, L, R) CHECK_RANK2(S, O, R, L) CHECK_RANK2(S, O, R, R)
#ifndef KOKKOSKERNELS_ENABLE_TPL_ROCSPARSE   // rocSPARSE claims the int-offset tuples (spmv_tpl_spec_avail.hpp:126-150); KKAMD leaves them alone then
CHECK_ALL(double, int)
CHECK_ALL(float, int)
#endif
CHECK_ALL(double, size_t)
CHECK_ALL(float, size_t)
// not claimed: 64-bit ordinals, complex scalars -> the native Kokkos path
static_assert(!KokkosSparse::Impl::spmv_tpl_spec_avail<
                  Kokkos::HIP, KokkosSparse::Impl::SPMVHandleImpl<Kokkos::HIP, Kokkos::HIPSpace, double, int, int64_t>,
                  KokkosSparse::CrsMatrix<const double, const int64_t, device, Kokkos::MemoryTraits<Kokkos::Unmanaged>, const int>,
                  Kokkos::View<const double*, L, device, Kokkos::MemoryTraits<Kokkos::Unmanaged | Kokkos::RandomAccess>>,
                  Kokkos::View<double*, L, device, Kokkos::MemoryTraits<Kokkos::Unmanaged>>>::value, "int64 ordinals must stay native");
}  // namespace
int main() { return 0; }
