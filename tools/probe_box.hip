// Box probe: host/GPU facts the scheduler design depends on (pinned H2D/D2H rate, pinning limits).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  size_t gb = argc > 1 ? atol(argv[1]) : 2;
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d mem=%.1f GB clock=%d\n", p.name, p.multiProcessorCount, p.totalGlobalMem / 1e9, p.clockRate);
  size_t bytes = gb << 30;
  void *h, *d; 
  auto t0 = std::chrono::steady_clock::now();
  CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
  auto t1 = std::chrono::steady_clock::now();
  printf("hipHostMalloc %zu GB: %.3f s\n", gb, std::chrono::duration<double>(t1 - t0).count());
  memset(h, 1, bytes);
  CK(hipMalloc(&d, bytes));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int it = 0; it < 3; ++it) {
    CK(hipEventRecord(a, s)); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s)); float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("H2D pinned %zu GB: %.2f ms = %.2f GB/s\n", gb, ms, bytes / ms / 1e6);
  }
  for (int it = 0; it < 2; ++it) {
    CK(hipEventRecord(a, s)); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s)); CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s)); float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("D2H pinned %zu GB: %.2f ms = %.2f GB/s\n", gb, ms, bytes / ms / 1e6);
  }
  // malloc + hipHostRegister path (what the NUMA/CXL tier uses)
  void* m = aligned_alloc(4096, bytes); memset(m, 2, bytes);
  t0 = std::chrono::steady_clock::now();
  hipError_t e = hipHostRegister(m, bytes, hipHostRegisterDefault);
  t1 = std::chrono::steady_clock::now();
  printf("hipHostRegister %zu GB: %s %.3f s\n", gb, hipGetErrorString(e), std::chrono::duration<double>(t1 - t0).count());
  if (e == hipSuccess) {
    CK(hipEventRecord(a, s)); CK(hipMemcpyAsync(d, m, bytes, hipMemcpyHostToDevice, s)); CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s)); float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("H2D registered %zu GB: %.2f ms = %.2f GB/s\n", gb, ms, bytes / ms / 1e6);
  }
  // concurrent H2D + D2H
  hipStream_t s2; CK(hipStreamCreate(&s2)); void* d2; CK(hipMalloc(&d2, bytes));
  auto w0 = std::chrono::steady_clock::now();
  CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
  if (e == hipSuccess) CK(hipMemcpyAsync(m, d2, bytes, hipMemcpyDeviceToHost, s2));
  CK(hipDeviceSynchronize());
  auto w1 = std::chrono::steady_clock::now();
  printf("bidir %zu GB each: %.2f ms\n", gb, std::chrono::duration<double>(w1 - w0).count() * 1e3);
  return 0;
}
