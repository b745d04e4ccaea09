// Kokkos_Shim.hpp -- a minimal stand-in for the slice of Kokkos core the KokkosSparse public headers touch
// (View / Device / HIP execution space / deep_copy / create_mirror_view / Profiling regions), so the
// drop-in headers in this directory compile and run in an environment without Kokkos (Kokkos
// core is neither installed nor vendored here).  With real Kokkos present this file is NOT used -- the
// library is bound through the TPL specialisations in host/kokkos_tpl/ instead.
//
// Only what the hot path's API needs is modelled: rank-1/rank-2 Views over HIPSpace or HostSpace,
// LayoutLeft / LayoutRight, managed (reference counted) and Unmanaged memory, const conversion.
#pragma once
#include <hip/hip_runtime_api.h>
#include "../../include/kkamd.h"
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace Kokkos {

struct LayoutLeft {};
struct LayoutRight {};
struct HostSpace { static constexpr bool on_device = false; };
struct HIPSpace { static constexpr bool on_device = true; };

class HIP {
 public:
  using execution_space = HIP;
  using memory_space    = HIPSpace;
  HIP() : stream_(nullptr) {}
  explicit HIP(hipStream_t s) : stream_(s) {}
  hipStream_t hip_stream() const { return stream_; }
  void fence(const std::string& = std::string()) const { (void)hipStreamSynchronize(stream_); }
  bool operator==(const HIP& o) const { return stream_ == o.stream_; }
  bool operator!=(const HIP& o) const { return stream_ != o.stream_; }
  static const char* name() { return "HIP"; }
 private:
  hipStream_t stream_;
};
class Serial {
 public:
  using execution_space = Serial;
  using memory_space    = HostSpace;
  void fence(const std::string& = std::string()) const {}
  static const char* name() { return "Serial"; }
};
using DefaultExecutionSpace     = HIP;
using DefaultHostExecutionSpace = Serial;

template <class Exec, class Mem> struct Device {
  using execution_space = Exec;
  using memory_space    = Mem;
  using device_type     = Device<Exec, Mem>;
};

enum MemoryTraitsFlags : unsigned { Unmanaged = 0x1, RandomAccess = 0x2 };
template <unsigned F> struct MemoryTraits {
  static constexpr bool is_unmanaged     = (F & Unmanaged) != 0;
  static constexpr bool is_random_access = (F & RandomAccess) != 0;
};
using MemoryUnmanaged = MemoryTraits<Unmanaged>;

struct WithoutInitializing_t {};
static constexpr WithoutInitializing_t WithoutInitializing{};
struct ViewAllocProp { std::string label; bool initialize = true; };
inline ViewAllocProp view_alloc(const std::string& label) { return ViewAllocProp{label, true}; }
inline ViewAllocProp view_alloc(WithoutInitializing_t, const std::string& label) { return ViewAllocProp{label, false}; }
inline ViewAllocProp view_alloc(const std::string& label, WithoutInitializing_t) { return ViewAllocProp{label, false}; }

inline void initialize(int&, char**) {}
inline void initialize() {}
inline void finalize() { (void)kkamd_release_scratch(); }   // the library's process-wide scratch (SpGEMM bitmap pool, handle-less SpMV plan) goes with the run time
inline void fence(const std::string& = std::string()) { (void)hipDeviceSynchronize(); }

namespace Profiling {
// roctx ranges through the library (rocprofv3 --marker-trace shows them like a Kokkos Tools connector would)
inline void pushRegion(const std::string& label) { (void)kkamd_trace_push(label.c_str()); }
inline void popRegion() { (void)kkamd_trace_pop(); }
}  // namespace Profiling

namespace Impl {
// ---- property pack parsing ---------------------------------------------------------------
template <class T> struct is_layout : std::false_type {};
template <> struct is_layout<LayoutLeft> : std::true_type {};
template <> struct is_layout<LayoutRight> : std::true_type {};
template <class T> struct is_memtraits : std::false_type {};
template <unsigned F> struct is_memtraits<MemoryTraits<F>> : std::true_type {};
template <class T, class = void> struct space_of { using type = void; };
template <> struct space_of<HIPSpace> { using type = Device<HIP, HIPSpace>; };
template <> struct space_of<HostSpace> { using type = Device<Serial, HostSpace>; };
template <> struct space_of<HIP> { using type = Device<HIP, HIPSpace>; };
template <> struct space_of<Serial> { using type = Device<Serial, HostSpace>; };
template <class E, class M> struct space_of<Device<E, M>> { using type = Device<E, M>; };

template <class... P> struct pick_layout { using type = LayoutLeft; };
template <class P0, class... P> struct pick_layout<P0, P...> {
  using type = std::conditional_t<is_layout<P0>::value, P0, typename pick_layout<P...>::type>;
};
template <class... P> struct pick_traits { using type = MemoryTraits<0>; };
template <class P0, class... P> struct pick_traits<P0, P...> {
  using type = std::conditional_t<is_memtraits<P0>::value, P0, typename pick_traits<P...>::type>;
};
template <class... P> struct pick_device { using type = Device<HIP, HIPSpace>; };
template <class P0, class... P> struct pick_device<P0, P...> {
  using type = std::conditional_t<!std::is_void<typename space_of<P0>::type>::value, typename space_of<P0>::type,
                                  typename pick_device<P...>::type>;
};
template <class D> struct data_traits;
template <class T> struct data_traits<T*> { using value_type = T; static constexpr int rank = 1; };
template <class T> struct data_traits<T**> { using value_type = T; static constexpr int rank = 2; };

inline void check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace Impl

template <class DataType, class... Props>
class View {
 public:
  using traits               = View;
  using data_type            = DataType;
  using value_type           = typename Impl::data_traits<DataType>::value_type;
  using non_const_value_type = std::remove_const_t<value_type>;
  using const_value_type     = std::add_const_t<value_type>;
  using array_layout         = typename Impl::pick_layout<Props...>::type;
  using device_type          = typename Impl::pick_device<Props...>::type;
  using execution_space      = typename device_type::execution_space;
  using memory_space         = typename device_type::memory_space;
  using memory_traits        = typename Impl::pick_traits<Props...>::type;
  using size_type            = size_t;
  static constexpr int Rank  = Impl::data_traits<DataType>::rank;
  static constexpr size_t rank() { return (size_t)Rank; }
  using non_const_data_type = std::conditional_t<Rank == 1, non_const_value_type*, non_const_value_type**>;
  using const_data_type     = std::conditional_t<Rank == 1, const_value_type*, const_value_type**>;
  using non_const_type      = View<non_const_data_type, array_layout, device_type, memory_traits>;
  using const_type          = View<const_data_type, array_layout, device_type, memory_traits>;
  using host_mirror_type    = View<non_const_data_type, array_layout, Device<Serial, HostSpace>>;
  using HostMirror          = host_mirror_type;

  View() = default;
  // managed allocation
  explicit View(const std::string& label, size_t n0 = 0, size_t n1 = 1) { allocate(ViewAllocProp{label, true}, n0, n1); }
  explicit View(const ViewAllocProp& prop, size_t n0 = 0, size_t n1 = 1) { allocate(prop, n0, n1); }
  // unmanaged wrap of existing memory
  View(value_type* ptr, size_t n0, size_t n1 = 1) : ptr_(ptr), n0_(n0), n1_(Rank == 2 ? n1 : 1) { default_strides(); }
  View(value_type* ptr, size_t n0, size_t n1, size_t s0, size_t s1) : ptr_(ptr), n0_(n0), n1_(n1), s0_(s0), s1_(s1) {}
  // converting copy (adds const / drops ownership traits); shares the allocation
  template <class D2, class... P2, class = std::enable_if_t<
      std::is_same<std::remove_const_t<typename View<D2, P2...>::value_type>, non_const_value_type>::value &&
      (std::is_const<value_type>::value || !std::is_const<typename View<D2, P2...>::value_type>::value) &&
      View<D2, P2...>::Rank == Rank>>
  View(const View<D2, P2...>& o)
      : ptr_(const_cast<value_type*>(o.data())), n0_(o.extent(0)), n1_(o.extent(1)), s0_(o.stride(0)), s1_(o.stride(1)),
        own_(o.ownership()), label_(o.label()) {}

  value_type* data() const { return ptr_; }
  size_t extent(int i) const { return i == 0 ? n0_ : (i == 1 ? n1_ : 1); }
  int extent_int(int i) const { return (int)extent(i); }
  size_t stride(int i) const { return i == 0 ? s0_ : s1_; }
  size_t stride_0() const { return s0_; }
  size_t stride_1() const { return s1_; }
  size_t size() const { return n0_ * n1_; }
  size_t span() const { return size(); }
  bool span_is_contiguous() const { return true; }
  const std::string& label() const { return label_; }
  const std::shared_ptr<void>& ownership() const { return own_; }
  bool is_allocated() const { return ptr_ != nullptr; }

  // element access: host-space views only (device data is reached through deep_copy)
  template <class M = memory_space> std::enable_if_t<!M::on_device, value_type&> operator()(size_t i) const { return ptr_[i * s0_]; }
  template <class M = memory_space> std::enable_if_t<!M::on_device, value_type&> operator()(size_t i, size_t j) const { return ptr_[i * s0_ + j * s1_]; }
  template <class M = memory_space> std::enable_if_t<!M::on_device, value_type&> operator[](size_t i) const { return ptr_[i * s0_]; }

 private:
  void default_strides() {
    if (Rank == 1) { s0_ = 1; s1_ = n0_; }
    else if (std::is_same<array_layout, LayoutLeft>::value) { s0_ = 1; s1_ = n0_; }
    else { s0_ = n1_; s1_ = 1; }
  }
  void allocate(const ViewAllocProp& prop, size_t n0, size_t n1) {
    n0_ = n0; n1_ = Rank == 2 ? n1 : 1; label_ = prop.label; default_strides();
    const size_t bytes = n0_ * n1_ * sizeof(value_type);
    if (bytes == 0) return;
    void* p = nullptr;
    if (memory_space::on_device) {
      Impl::check(hipMalloc(&p, bytes), "Kokkos::View allocation (hipMalloc)");
      if (prop.initialize) Impl::check(hipMemset(p, 0, bytes), "Kokkos::View initialisation");
      own_ = std::shared_ptr<void>(p, [](void* q) { (void)hipFree(q); });
    } else {
      p = prop.initialize ? std::calloc(1, bytes) : std::malloc(bytes);
      if (!p) throw std::bad_alloc();
      own_ = std::shared_ptr<void>(p, [](void* q) { std::free(q); });
    }
    ptr_ = static_cast<value_type*>(p);
  }
  value_type* ptr_ = nullptr;
  size_t n0_ = 0, n1_ = 1, s0_ = 1, s1_ = 0;
  std::shared_ptr<void> own_;
  std::string label_;
};

template <class T> struct is_view : std::false_type {};
template <class D, class... P> struct is_view<View<D, P...>> : std::true_type {};
template <class D, class... P> struct is_view<const View<D, P...>> : std::true_type {};

template <class Space, class MemSpace> struct SpaceAccessibility {
  static constexpr bool accessible = std::is_same<typename Space::memory_space, MemSpace>::value;
};

template <class V> typename V::host_mirror_type create_mirror_view(const V& v) {
  return typename V::host_mirror_type(view_alloc(WithoutInitializing, v.label() + "_mirror"), v.extent(0), v.extent(1));
}
template <class V> typename V::host_mirror_type create_mirror(const V& v) { return create_mirror_view(v); }

// deep_copy between views of identical extents and layout (contiguous), any pair of spaces
template <class D1, class... P1, class D2, class... P2>
void deep_copy(const View<D1, P1...>& dst, const View<D2, P2...>& src) {
  if (dst.extent(0) != src.extent(0) || dst.extent(1) != src.extent(1))
    throw std::runtime_error("Kokkos::deep_copy: extents do not match");
  const size_t bytes = dst.size() * sizeof(typename View<D1, P1...>::value_type);
  if (!bytes) return;
  if (dst.stride(0) != src.stride(0) || dst.stride(1) != src.stride(1))
    throw std::runtime_error("Kokkos::deep_copy (shim): layouts differ");
  Impl::check(hipMemcpy((void*)dst.data(), (const void*)src.data(), bytes, hipMemcpyDefault), "Kokkos::deep_copy");
}
template <class E, class D1, class... P1, class D2, class... P2>
void deep_copy(const E& exec, const View<D1, P1...>& dst, const View<D2, P2...>& src) { exec.fence(); deep_copy(dst, src); }
// fill
template <class D1, class... P1>
void deep_copy(const View<D1, P1...>& dst, const typename View<D1, P1...>::non_const_value_type& v) {
  using T = typename View<D1, P1...>::non_const_value_type;
  const size_t n = dst.size();
  if (!n) return;
  std::vector<T> tmp(n, v);
  Impl::check(hipMemcpy((void*)dst.data(), tmp.data(), n * sizeof(T), hipMemcpyDefault), "Kokkos::deep_copy(fill)");
}

struct ALL_t {};
inline ALL_t ALL() { return ALL_t(); }
// column j of a rank-2 view as a (possibly strided) rank-1 view
template <class T, class... P> View<T*, P...> subview(const View<T**, P...>& v, ALL_t, size_t j) {
  return View<T*, P...>(v.data() + j * v.stride(1), v.extent(0), 1, v.stride(0), v.extent(0));
}
// columns [j0, j1) of a rank-2 view
template <class T, class... P> View<T**, P...> subview(const View<T**, P...>& v, ALL_t, std::pair<int, int> r) {
  return View<T**, P...>(v.data() + (size_t)r.first * v.stride(1), v.extent(0), (size_t)(r.second - r.first), v.stride(0), v.stride(1));
}
template <class A, class B> using pair = std::pair<A, B>;

}  // namespace Kokkos

namespace KokkosKernels {
using default_scalar  = double;
using default_lno_t   = int;
using default_size_type = int;
using default_layout  = Kokkos::LayoutLeft;
using default_device  = Kokkos::Device<Kokkos::HIP, Kokkos::HIPSpace>;
namespace Impl {
// common/src/KokkosKernels_Error.hpp:26
inline void throw_runtime_exception(const std::string& msg) { throw std::runtime_error(msg); }
}  // namespace Impl
}  // namespace KokkosKernels
