// TEST INFRASTRUCTURE -- a declarations-only stand-in for Kokkos core, just deep enough for `g++ -fsyntax-only` to parse the
// REFERENCE's own headers (sparse/src/KokkosSparse_CrsMatrix.hpp, KokkosSparse_spmv_handle.hpp, sparse/impl/*_spec.hpp, ...)
// with Kokkos::HIP enabled, so that tests/test_tpl_binding.py can prove that the TPL specialisations under
// kokkos-kernels_amd/host/kokkos_tpl/ name exactly the type tuples the reference instantiates.  Nothing here runs.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>
#include <iostream>
#include <algorithm>
#include <functional>
#include <map>
#include <memory>
#include <set>

#define KOKKOS_ENABLE_HIP
#define KOKKOS_ENABLE_SERIAL
#define KOKKOS_VERSION 40700
#define KOKKOS_VERSION_GREATER_EQUAL(a, b, c) (KOKKOS_VERSION >= ((a)*10000 + (b)*100 + (c)))
#define KOKKOS_VERSION_LESS(a, b, c) (KOKKOS_VERSION < ((a)*10000 + (b)*100 + (c)))
#define KOKKOS_INLINE_FUNCTION inline
#define KOKKOS_FORCEINLINE_FUNCTION inline
#define KOKKOS_FUNCTION
#define KOKKOS_DEFAULTED_FUNCTION
#define KOKKOS_LAMBDA [=]
#define KOKKOS_CLASS_LAMBDA [=, *this]
#define KOKKOS_IMPL_HIP_SAFE_CALL(x) (void)(x)
#define KOKKOS_ASSERT(x) (void)(x)
#define KOKKOS_DEPRECATED
#define KOKKOS_DEPRECATED_WITH_COMMENT(x)

#include <hip/hip_runtime_api.h>   // the real HIP runtime API declarations (hipStream_t, hipError_t, ...), as Kokkos' HIP backend pulls them in

namespace Kokkos {

struct LayoutLeft {};
struct LayoutRight {};
struct LayoutStride {};
struct HostSpace { using memory_space = HostSpace; using execution_space = struct Serial; static const char* name() { return "Host"; } };
struct HIPSpace { using memory_space = HIPSpace; using execution_space = class HIP; static const char* name() { return "HIP"; } };
struct HIPManagedSpace { using memory_space = HIPManagedSpace; using execution_space = class HIP; };
struct HIPHostPinnedSpace { using memory_space = HIPHostPinnedSpace; using execution_space = class HIP; };
struct AnonymousSpace {};

class HIP {
 public:
  using execution_space = HIP;
  using memory_space    = HIPSpace;
  using size_type       = unsigned;
  using array_layout    = LayoutLeft;
  using scratch_memory_space = HIPSpace;
  HIP() = default;
  explicit HIP(hipStream_t s) : stream_(s) {}
  hipStream_t hip_stream() const { return stream_; }
  void fence(const std::string& = std::string()) const {}
  int concurrency() const { return 1; }
  bool operator==(const HIP& o) const { return stream_ == o.stream_; }
  bool operator!=(const HIP& o) const { return stream_ != o.stream_; }
  static const char* name() { return "HIP"; }
 private:
  hipStream_t stream_ = nullptr;
};
struct Serial {
  using execution_space = Serial;
  using memory_space    = HostSpace;
  using size_type       = size_t;
  using array_layout    = LayoutRight;
  using scratch_memory_space = HostSpace;
  void fence(const std::string& = std::string()) const {}
  int concurrency() const { return 1; }
  bool operator==(const Serial&) const { return true; }
  bool operator!=(const Serial&) const { return false; }
  static const char* name() { return "Serial"; }
};
using DefaultExecutionSpace     = HIP;
using DefaultHostExecutionSpace = Serial;

template <class Exec, class Mem> struct Device {
  using execution_space = Exec;
  using memory_space    = Mem;
  using device_type     = Device<Exec, Mem>;
};

enum MemoryTraitsFlags : unsigned { Unmanaged = 0x1, RandomAccess = 0x2, Atomic = 0x4, Restrict = 0x8, Aligned = 0x10 };
template <unsigned F = 0> struct MemoryTraits {
  using memory_traits = MemoryTraits<F>;
  static constexpr bool is_unmanaged     = (F & Unmanaged) != 0;
  static constexpr bool is_random_access = (F & RandomAccess) != 0;
  static constexpr unsigned impl_value   = F;
};
using MemoryManaged   = MemoryTraits<0>;
using MemoryUnmanaged = MemoryTraits<Unmanaged>;
template <class T> struct is_memory_traits : std::false_type {};
template <unsigned F> struct is_memory_traits<MemoryTraits<F>> : std::true_type {};
template <class T> inline constexpr bool is_memory_traits_v = is_memory_traits<T>::value;
template <class T> struct is_execution_space : std::false_type {};
template <> struct is_execution_space<HIP> : std::true_type {};
template <> struct is_execution_space<Serial> : std::true_type {};
template <class T> inline constexpr bool is_execution_space_v = is_execution_space<T>::value;
template <class T> struct is_memory_space : std::false_type {};
template <> struct is_memory_space<HIPSpace> : std::true_type {};
template <> struct is_memory_space<HIPManagedSpace> : std::true_type {};
template <> struct is_memory_space<HostSpace> : std::true_type {};
template <class T> struct is_device : std::false_type {};
template <class E, class M> struct is_device<Device<E, M>> : std::true_type {};
template <class T> struct is_array_layout : std::false_type {};
template <> struct is_array_layout<LayoutLeft> : std::true_type {};
template <> struct is_array_layout<LayoutRight> : std::true_type {};
template <> struct is_array_layout<LayoutStride> : std::true_type {};

struct WithoutInitializing_t {};
static constexpr WithoutInitializing_t WithoutInitializing{};
struct ViewAllocProp { std::string label; };
template <class... A> ViewAllocProp view_alloc(const A&...) { return ViewAllocProp{}; }

inline void fence(const std::string& = std::string()) {}
[[noreturn]] inline void abort(const char*) { std::abort(); }
template <class... A> void printf(const char*, A...) {}

namespace Profiling {
inline void pushRegion(const std::string&) {}
inline void popRegion() {}
}  // namespace Profiling

namespace Impl {
template <class T, class = void> struct space_of { using type = void; };
template <> struct space_of<HIPSpace> { using type = Device<HIP, HIPSpace>; };
template <> struct space_of<HIPManagedSpace> { using type = Device<HIP, HIPManagedSpace>; };
template <> struct space_of<HostSpace> { using type = Device<Serial, HostSpace>; };
template <> struct space_of<HIP> { using type = Device<HIP, HIPSpace>; };
template <> struct space_of<Serial> { using type = Device<Serial, HostSpace>; };
template <class E, class M> struct space_of<Device<E, M>> { using type = Device<E, M>; };
template <class... P> struct pick_layout { using type = void; };
template <class P0, class... P> struct pick_layout<P0, P...> {
  using type = std::conditional_t<is_array_layout<P0>::value, P0, typename pick_layout<P...>::type>;
};
template <class... P> struct pick_traits { using type = MemoryTraits<0>; };
template <class P0, class... P> struct pick_traits<P0, P...> {
  using type = std::conditional_t<is_memory_traits<P0>::value, P0, typename pick_traits<P...>::type>;
};
template <class... P> struct pick_device { using type = Device<HIP, HIPSpace>; };
template <class P0, class... P> struct pick_device<P0, P...> {
  using type = std::conditional_t<!std::is_void<typename space_of<P0>::type>::value, typename space_of<P0>::type,
                                  typename pick_device<P...>::type>;
};
template <class D> struct data_traits { using value_type = D; static constexpr unsigned rank = 0; };
template <class T> struct data_traits<T*> { using value_type = typename data_traits<T>::value_type; static constexpr unsigned rank = data_traits<T>::rank + 1; };
template <class T, size_t N> struct data_traits<T[N]> { using value_type = typename data_traits<T>::value_type; static constexpr unsigned rank = data_traits<T>::rank + 1; };
template <class V, unsigned R> struct add_stars { using type = typename add_stars<V, R - 1>::type*; };
template <class V> struct add_stars<V, 0> { using type = V; };
}  // namespace Impl

template <class DataType, class... Props>
struct ViewTraits {
  using data_type            = DataType;
  using value_type           = typename Impl::data_traits<DataType>::value_type;
  using non_const_value_type = std::remove_const_t<value_type>;
  using const_value_type     = std::add_const_t<value_type>;
  static constexpr unsigned rank = Impl::data_traits<DataType>::rank;
  using device_type          = typename Impl::pick_device<Props...>::type;
  using execution_space      = typename device_type::execution_space;
  using memory_space         = typename device_type::memory_space;
  using array_layout         = std::conditional_t<std::is_void<typename Impl::pick_layout<Props...>::type>::value,
                                                  typename execution_space::array_layout, typename Impl::pick_layout<Props...>::type>;
  using memory_traits        = typename Impl::pick_traits<Props...>::type;
  using size_type            = size_t;
  using non_const_data_type  = typename Impl::add_stars<non_const_value_type, rank>::type;
  using const_data_type      = typename Impl::add_stars<const_value_type, rank>::type;
  using host_mirror_space    = Device<Serial, HostSpace>;
  using HostMirrorSpace      = host_mirror_space;
  using specialize           = void;
  static constexpr bool is_managed = !memory_traits::is_unmanaged;
};

template <class DataType, class... Props>
class View : public ViewTraits<DataType, Props...> {
 public:
  using traits               = ViewTraits<DataType, Props...>;
  using typename traits::value_type;
  using typename traits::non_const_value_type;
  using typename traits::const_value_type;
  using typename traits::array_layout;
  using typename traits::device_type;
  using typename traits::memory_traits;
  using typename traits::non_const_data_type;
  using typename traits::const_data_type;
  using reference_type       = value_type&;
  using pointer_type         = value_type*;
  static constexpr unsigned Rank = traits::rank;
  static constexpr size_t rank() { return Rank; }
  static constexpr size_t rank_dynamic = Rank;
  using non_const_type   = View<non_const_data_type, array_layout, device_type, memory_traits>;
  using const_type       = View<const_data_type, array_layout, device_type, memory_traits>;
  using host_mirror_type = View<non_const_data_type, array_layout, Device<Serial, HostSpace>>;
  using HostMirror       = host_mirror_type;
  using uniform_type     = View;
  using uniform_const_type = const_type;
  using uniform_runtime_nomemspace_type = View;
  using uniform_runtime_const_nomemspace_type = const_type;

  View() = default;
  View(const View&) = default;
  View& operator=(const View&) = default;
  template <class... A> explicit View(const std::string&, A...) {}
  template <class... A> explicit View(const ViewAllocProp&, A...) {}
  template <class... A> View(value_type* p, A...) : ptr_(p) {}
  template <class D2, class... P2> View(const View<D2, P2...>& o) : ptr_(const_cast<value_type*>(o.data())) {}
  template <class D2, class... P2> View& operator=(const View<D2, P2...>& o) { ptr_ = const_cast<value_type*>(o.data()); return *this; }

  value_type* data() const { return ptr_; }
  size_t extent(int) const { return 0; }
  int extent_int(int) const { return 0; }
  size_t stride(int) const { return 1; }
  size_t stride_0() const { return 1; }
  size_t stride_1() const { return 1; }
  template <class I> void stride(I*) const {}
  size_t size() const { return 0; }
  size_t span() const { return 0; }
  bool span_is_contiguous() const { return true; }
  std::string label() const { return std::string(); }
  bool is_allocated() const { return ptr_ != nullptr; }
  int use_count() const { return 0; }
  array_layout layout() const { return array_layout(); }
  template <class... I> value_type& operator()(I...) const { return *ptr_; }
  template <class I> value_type& operator[](I) const { return *ptr_; }
  template <class... I> value_type& access(I...) const { return *ptr_; }
 private:
  value_type* ptr_ = nullptr;
};

template <class T> struct is_view : std::false_type {};
template <class D, class... P> struct is_view<View<D, P...>> : std::true_type {};
template <class D, class... P> struct is_view<const View<D, P...>> : std::true_type {};
template <class T> inline constexpr bool is_view_v = is_view<T>::value;

template <class Space, class MemSpace> struct SpaceAccessibility {
  static constexpr bool accessible = true, assignable = true, deepcopy = true;
};

template <class V> typename V::host_mirror_type create_mirror_view(const V&) { return typename V::host_mirror_type(); }
template <class S, class V> V create_mirror_view(const S&, const V& v) { return v; }
template <class V> typename V::host_mirror_type create_mirror(const V&) { return typename V::host_mirror_type(); }
template <class S, class V> V create_mirror_view_and_copy(const S&, const V& v) { return v; }
template <class... A> void deep_copy(const A&...) {}
template <class V, class... A> void resize(V&, A...) {}
template <class V, class... A> void realloc(V&, A...) {}

struct ALL_t { constexpr const ALL_t& operator()() const { return *this; } };
static constexpr ALL_t ALL{};
template <class A, class B> using pair = std::pair<A, B>;
template <class A, class B> std::pair<A, B> make_pair(A a, B b) { return std::pair<A, B>(a, b); }
template <class V, class... A> V subview(const V& v, A...) { return v; }

template <class T> T atomic_fetch_add(T* p, T v) { T o = *p; *p += v; return o; }
template <class T, class U> void atomic_add(T* p, U v) { *p += v; }
template <class T, class U> void atomic_store(T* p, U v) { *p = v; }
template <class T> T atomic_load(T* p) { return *p; }
template <class T> bool atomic_compare_exchange_strong(T* p, T c, T v) { if (*p == c) { *p = v; return true; } return false; }

template <class... P> struct RangePolicy { using member_type = int; template <class... A> RangePolicy(A...) {} };
template <class... P> struct TeamPolicy {
  struct member_type { int league_rank() const { return 0; } int team_rank() const { return 0; } int team_size() const { return 1; } int league_size() const { return 1; } void team_barrier() const {} };
  template <class... A> TeamPolicy(A...) {}
  template <class F, class T> int team_size_recommended(const F&, const T&) const { return 1; }
  template <class F, class T> int team_size_max(const F&, const T&) const { return 1; }
};
struct AUTO_t {};
static constexpr AUTO_t AUTO{};
struct ParallelForTag {};
struct ParallelReduceTag {};
template <class... A> void parallel_for(const A&...) {}
template <class... A> void parallel_reduce(const A&...) {}
template <class... A> void parallel_scan(const A&...) {}
template <class M, class I> int TeamThreadRange(const M&, I) { return 0; }
template <class M, class I, class J> int TeamThreadRange(const M&, I, J) { return 0; }
template <class M, class I> int ThreadVectorRange(const M&, I) { return 0; }
template <class M, class I, class J> int ThreadVectorRange(const M&, I, J) { return 0; }
template <class M, class I> int TeamVectorRange(const M&, I) { return 0; }
template <class M> int PerTeam(const M&) { return 0; }
template <class M> int PerThread(const M&) { return 0; }
template <class... A> void single(const A&...) {}
template <class T, class... S> struct Max { using value_type = T; template <class... A> Max(A&...) {} };
template <class T, class... S> struct Min { using value_type = T; template <class... A> Min(A&...) {} };
template <class T, class... S> struct Sum { using value_type = T; template <class... A> Sum(A&...) {} };
template <class T, class... S> struct MinMax { using value_type = T; template <class... A> MinMax(A&...) {} };
template <class T> struct MinMaxScalar { T min_val, max_val; };
template <class T> struct reduction_identity { static T sum() { return T(); } static T max() { return T(); } static T min() { return T(); } };
template <class... P> struct Schedule {};
struct Dynamic {};
struct Static {};
template <class T> struct LaunchBounds {};

template <class T> class complex {
 public:
  using value_type = T;
  complex() = default;
  complex(T r, T i = T()) : re_(r), im_(i) {}
  T real() const { return re_; }
  T imag() const { return im_; }
 private:
  T re_ = T(), im_ = T();
};
namespace Experimental {
struct half_t { half_t() = default; half_t(float) {} operator float() const { return 0.f; } };
struct bhalf_t { bhalf_t() = default; bhalf_t(float) {} operator float() const { return 0.f; } };
template <class T> struct finite_max { static constexpr T value = std::numeric_limits<T>::max(); };
template <class T> struct finite_min { static constexpr T value = std::numeric_limits<T>::lowest(); };
template <class T> struct epsilon { static constexpr T value = std::numeric_limits<T>::epsilon(); };
}  // namespace Experimental

inline bool is_initialized() { return true; }
inline void initialize(int&, char**) {}
inline void finalize() {}
inline void push_finalize_hook(void (*)()) {}
template <class F> void push_finalize_hook(F) {}

}  // namespace Kokkos
// ---- containers, math and the rest of what the reference's common/ and sparse/ headers name ------------------------------
namespace Kokkos {
using std::ceil; using std::floor; using std::log2; using std::log; using std::sqrt; using std::pow; using std::abs; using std::fabs;
using std::exp; using std::min; using std::max; using std::isnan; using std::isinf; using std::round; using std::fmin; using std::fmax;
struct UnorderedMapInsertResult { bool success() const { return true; } bool existing() const { return false; } bool failed() const { return false; } unsigned index() const { return 0; } };
template <class K, class V, class D = Device<HIP, HIPSpace>, class H = void, class E = void>
class UnorderedMap {
 public:
  using size_type = unsigned;
  template <class... A> UnorderedMap(A...) {}
  template <class... A> UnorderedMapInsertResult insert(A...) const { return UnorderedMapInsertResult(); }
  template <class Q> size_type find(const Q&) const { return 0; }
  bool valid_at(size_type) const { return true; }
  size_type size() const { return 0; }
  size_type capacity() const { return 0; }
  bool failed_insert() const { return false; }
  bool rehash(size_type) { return true; }
  void clear() {}
  template <class I> K key_at(I) const { return K(); }
  template <class I> V& value_at(I) const { static V v; return v; }
  static constexpr size_type invalid_index = ~0u;
};
template <class D = Device<HIP, HIPSpace>> class Bitset { public: template <class... A> Bitset(A...) {} bool test(unsigned) const { return false; } bool set(unsigned) const { return true; } bool reset(unsigned) const { return true; } void clear() {} unsigned size() const { return 0; } unsigned count() const { return 0; } };
template <class D = Device<HIP, HIPSpace>> class ConstBitset { public: template <class... A> ConstBitset(A...) {} bool test(unsigned) const { return false; } unsigned size() const { return 0; } };
template <class... A> void sort(const A&...) {}
class Timer { public: double seconds() const { return 0; } void reset() {} };
template <class D = Device<HIP, HIPSpace>> struct Random_XorShift64_Pool { struct generator_type { template <class... A> int rand(A...) { return 0; } template <class... A> unsigned urand(A...) { return 0; } template <class... A> double drand(A...) { return 0; } template <class... A> float frand(A...) { return 0; } template <class... A> int64_t rand64(A...) { return 0; } template <class... A> uint64_t urand64(A...) { return 0; } }; template <class... A> Random_XorShift64_Pool(A...) {} generator_type get_state() const { return generator_type(); } void free_state(const generator_type&) const {} };
template <class G, class T> struct rand { template <class... A> static T draw(G&, A...) { return T(); } static T max() { return T(); } };
template <class V, class P, class... A> void fill_random(const V&, P&, A...) {}
template <class E, class V, class P, class... A> void fill_random(const E&, const V&, P&, A...) {}
template <class T, size_t N = 0> struct Array { T m[N ? N : 1]; T& operator[](size_t i) { return m[i]; } const T& operator[](size_t i) const { return m[i]; } };
template <class... P> struct MDRangePolicy { template <class... A> MDRangePolicy(A...) {} };
template <unsigned N, class... P> struct Rank {};
template <class T> struct IndexType {};
template <class T> struct WorkTag {};
template <class S> struct ScratchMemorySpace {};
template <class T> T clamp(T v, T, T) { return v; }
struct InitializationSettings {};
struct ScopeGuard { template <class... A> ScopeGuard(A&&...) {} };
inline int device_id() { return 0; }
inline int num_threads() { return 1; }
template <class... A> void memory_fence(A...) {}
template <class T, class U> T atomic_fetch_or(T* p, U v) { T o = *p; *p |= v; return o; }
template <class T, class U> T atomic_fetch_and(T* p, U v) { T o = *p; *p &= v; return o; }
template <class T, class U> T atomic_fetch_max(T* p, U v) { T o = *p; if (v > o) *p = v; return o; }
template <class T, class U> T atomic_fetch_min(T* p, U v) { T o = *p; if (v < o) *p = v; return o; }
template <class T, class U> void atomic_max(T* p, U v) { if (v > *p) *p = v; }
template <class T, class U> void atomic_min(T* p, U v) { if (v < *p) *p = v; }
template <class T, class U> void atomic_or(T* p, U v) { *p |= v; }
template <class T> void atomic_increment(T* p) { ++*p; }
template <class T> void atomic_decrement(T* p) { --*p; }
template <class T, class U> T atomic_exchange(T* p, U v) { T o = *p; *p = v; return o; }
template <class T> T atomic_compare_exchange(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
template <class T, class U> T atomic_add_fetch(T* p, U v) { *p += v; return *p; }
template <class T, class U> T atomic_fetch_sub(T* p, U v) { T o = *p; *p -= v; return o; }
namespace Impl {
template <class T> T clz(T) { return T(); }
template <class T> constexpr bool always_false = false;
}  // namespace Impl
namespace Experimental {
template <class T> T clz_builtin(T) { return T(); }
template <class T> int countl_zero_builtin(T) { return 0; }
template <class T> int popcount_builtin(T) { return 0; }
template <class T> struct norm_min { static constexpr T value = std::numeric_limits<T>::min(); };
template <class T> struct infinity { static constexpr T value = std::numeric_limits<T>::infinity(); };
template <class T> struct quiet_NaN { static constexpr T value = std::numeric_limits<T>::quiet_NaN(); };
template <class T> struct digits { static constexpr int value = std::numeric_limits<T>::digits; };
template <class T> inline constexpr T finite_max_v = std::numeric_limits<T>::max();
template <class T> inline constexpr T finite_min_v = std::numeric_limits<T>::lowest();
template <class T> inline constexpr T epsilon_v = std::numeric_limits<T>::epsilon();
template <class... A> void partition_space(A...) {}
}  // namespace Experimental
}  // namespace Kokkos
#include <Kokkos_ArithTraits.hpp>
