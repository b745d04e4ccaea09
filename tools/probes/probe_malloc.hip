// What does a hipMalloc / hipFree pair of the SpGEMM store's size cost?  (The store went back to the device with the last handle; the
// bench's symbolic phase then took 1.7 s in two of three repetitions.)  Usage: probe_malloc [GB] [background GB held]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 14.0, held = argc > 2 ? atof(argv[2]) : 0.0;
  void* bg = nullptr;
  if (held > 0) { if (hipMalloc(&bg, (size_t)(held * 1e9)) != hipSuccess) { printf("background allocation failed\n"); return 1; } hipMemset(bg, 1, (size_t)(held * 1e9)); hipDeviceSynchronize(); }
  for (int i = 0; i < 6; ++i) {
    void* p = nullptr;
    const double t0 = now();
    if (hipMalloc(&p, (size_t)(gb * 1e9)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    const double t1 = now();
    hipMemset(p, 0, 1 << 20); hipDeviceSynchronize();
    const double t2 = now();
    hipMemset(p, 0, (size_t)(gb * 1e9)); hipDeviceSynchronize();
    const double t3 = now();
    hipFree(p);
    const double t4 = now();
    printf("%.0f GB (%.0f GB held): hipMalloc %8.2f ms, first touch of 1 MB %6.2f ms, memset of all %8.2f ms, hipFree %8.2f ms\n", gb, held, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
  }
  // the stream-ordered allocator with everything kept in its pool
  hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
  unsigned long long thr = ~0ull; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
  for (int i = 0; i < 4; ++i) {
    void* p = nullptr;
    const double t0 = now();
    if (hipMallocAsync(&p, (size_t)(gb * 1e9), 0) != hipSuccess) { printf("hipMallocAsync failed\n"); return 1; }
    hipStreamSynchronize(0);
    const double t1 = now();
    hipFreeAsync(p, 0); hipStreamSynchronize(0);
    const double t2 = now();
    printf("%.0f GB: hipMallocAsync %8.2f ms, hipFreeAsync %8.2f ms\n", gb, (t1 - t0) * 1e3, (t2 - t1) * 1e3);
  }
  const double t0 = now(); hipMemPoolTrimTo(pool, 0); const double t1 = now();
  size_t fr = 0, tot = 0; hipMemGetInfo(&fr, &tot);
  printf("hipMemPoolTrimTo(0) %8.2f ms; free afterwards %.1f GB of %.1f\n", (t1 - t0) * 1e3, fr / 1e9, tot / 1e9);
  return 0;
}
