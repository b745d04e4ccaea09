// TEST INFRASTRUCTURE (tests/test_tpl_binding.py): parsed with g++ -fsyntax-only against the REFERENCE's own
// sparse/impl/KokkosSparse_spgemm_{symbolic,numeric}_spec.hpp and sparse/src/KokkosKernels_Handle.hpp, with the reference's
// sparse/tpls/KokkosSparse_spgemm_{symbolic,numeric}_tpl_spec_{avail,decl}.hpp replaced by copies to which
// kokkos-kernels_amd/host/kokkos_tpl/*.append.hpp were appended and sparse/src/KokkosSparse_spgemm_handle.hpp by a copy with
// KokkosSparse_spgemm_handle.hpp.patch applied.  Builds the internal types the way the public spgemm_symbolic / spgemm_numeric
// do (sparse/src/KokkosSparse_spgemm_symbolic.hpp:60-117, ..._numeric.hpp:66-160) and asserts the KKAMD specialisations are chosen.
#include <KokkosSparse_spgemm_symbolic_spec.hpp>
#include <KokkosSparse_spgemm_numeric_spec.hpp>

namespace {
using device = Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>;
template <class T> using UView = Kokkos::View<T*, KokkosKernels::default_layout, device, Kokkos::MemoryTraits<Kokkos::Unmanaged>>;

template <class Scalar, class Offset>
struct Tuple {
  // what a user declares (perf_test/sparse/KokkosSparse_spgemm.cpp: KokkosKernelsHandle<size_type, lno_t, scalar_t, ExecSpace, MemSpace, MemSpace>)
  using UserHandle = KokkosKernels::Experimental::KokkosKernelsHandle<Offset, int, Scalar, Kokkos::HIP, Kokkos::HIPSpace, Kokkos::HIPSpace>;
  // sparse/src/KokkosSparse_spgemm_symbolic.hpp:60-75: the const handle the unification layer is instantiated with
  using const_handle_type = KokkosKernels::Experimental::KokkosKernelsHandle<typename UserHandle::const_size_type, typename UserHandle::const_nnz_lno_t,
                                                                             typename UserHandle::const_nnz_scalar_t, typename UserHandle::HandleExecSpace,
                                                                             typename UserHandle::HandleTempMemorySpace, typename UserHandle::HandlePersistentMemorySpace>;
  using Sym = KokkosSparse::Impl::SPGEMM_SYMBOLIC<const_handle_type, UView<const Offset>, UView<const int>, UView<const Offset>, UView<const int>, UView<Offset>>;
  using Num = KokkosSparse::Impl::SPGEMM_NUMERIC<const_handle_type, UView<const Offset>, UView<const int>, UView<const Scalar>, UView<const Offset>, UView<const int>,
                                                 UView<const Scalar>, UView<const Offset>, UView<int>, UView<Scalar>>;
  static constexpr bool sym_avail = KokkosSparse::Impl::spgemm_symbolic_tpl_spec_avail<const_handle_type, UView<const Offset>, UView<const int>, UView<const Offset>,
                                                                                      UView<const int>, UView<Offset>>::value;
  static constexpr bool num_avail = KokkosSparse::Impl::spgemm_numeric_tpl_spec_avail<const_handle_type, UView<const Offset>, UView<const int>, UView<const Scalar>,
                                                                                     UView<const Offset>, UView<const int>, UView<const Scalar>, UView<const Offset>,
                                                                                     UView<int>, UView<Scalar>>::value;
  static void call() {
    // the specialisations (not the primary templates) carry the member alias KernelHandle
    static_assert(std::is_same<typename Sym::KernelHandle, const_handle_type>::value && std::is_same<typename Num::KernelHandle, const_handle_type>::value,
                  "the KKAMD specialisation was not selected");
    const_handle_type kh;
    kh.create_spgemm_handle();
    UView<const Offset> rmA, rmB; UView<const int> eA, eB; UView<const Scalar> vA, vB; UView<Offset> rmC; UView<int> eC; UView<Scalar> vC;
    Sym::spgemm_symbolic(&kh, 1, 1, 1, rmA, eA, false, rmB, eB, false, rmC, false);
    UView<const Offset> rmCc(rmC);
    Num::spgemm_numeric(&kh, 1, 1, 1, rmA, eA, vA, false, rmB, eB, vB, false, rmCc, eC, vC);
  }
};
#define CHECK(S, O) static_assert(Tuple<S, O>::sym_avail && Tuple<S, O>::num_avail, "SpGEMM tuple not bound: " #S " " #O); template struct Tuple<S, O>;
#ifndef KOKKOSKERNELS_ENABLE_TPL_ROCSPARSE   // rocSPARSE claims the all-int tuple (spgemm_symbolic_tpl_spec_avail.hpp:72-95)
CHECK(double, int)
CHECK(float, int)
#endif
CHECK(double, size_t)
CHECK(float, size_t)
}  // namespace
int main() { return 0; }
