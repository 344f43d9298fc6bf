// Host -> device over PCIe: the copy engine (hipMemcpyAsync from pinned memory) against a kernel that reads the pinned buffer
// through its device mapping, and both at once.  build: hipcc --offload-arch=gfx950 -O3 tools/h2d_kernel_bench.hip -o tools/h2d_kernel_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void pull_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n, int unroll_dummy) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i + 3 * stride < n; i += 4 * stride) {
    uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  char *h, *d;
  CK(hipHostMalloc((void**)&h, bytes, hipHostMallocDefault));
  memset(h, 1, bytes);
  CK(hipMalloc((void**)&d, bytes));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, s0));
    CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s0));
    CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("copy engine, one 2 GB copy: %.2f GB/s\n", bytes / ms / 1e6);
  }
  {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, s0));
    CK(hipMemcpyAsync(d, h, bytes / 2, hipMemcpyHostToDevice, s0));
    CK(hipMemcpyAsync(d + bytes / 2, h + bytes / 2, bytes / 2, hipMemcpyHostToDevice, s1));
    CK(hipStreamSynchronize(s1));
    CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("copy engine, two streams x 1 GB: %.2f GB/s\n", bytes / ms / 1e6);
  }
  void* hd;
  CK(hipHostGetDevicePointer(&hd, h, 0));
  for (int wgs : {32, 64, 128, 256, 512, 1024, 2048}) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s0));
      hipLaunchKernelGGL(pull_kernel, dim3(wgs), dim3(256), 0, s0, (const uint4*)hd, (uint4*)d, bytes / 16, 0);
      CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("kernel pull, %4d workgroups: %.2f GB/s\n", wgs, bytes / ms / 1e6);
  }
  {
    // both at once: the engine moves the first half, a 256-workgroup kernel pulls the second
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, s0));
    CK(hipMemcpyAsync(d, h, bytes / 2, hipMemcpyHostToDevice, s0));
    hipLaunchKernelGGL(pull_kernel, dim3(256), dim3(256), 0, s1, (const uint4*)((char*)hd + bytes / 2), (uint4*)(d + bytes / 2), bytes / 32, 0);
    CK(hipStreamSynchronize(s1));
    CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("engine + kernel, 1 GB each: %.2f GB/s\n", bytes / ms / 1e6);
  }
  return 0;
}
