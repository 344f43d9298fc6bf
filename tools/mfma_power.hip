// Sustained MFMA rate on random vs zero register operands, 16x16x32 vs 32x32x16 bf16 (no memory traffic):
// how much of the prefill GEMM's gap to the 2.5 PFLOP/s peak is the chip clocking down, and does the shape matter?
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o tools/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ bf16x8 mk(uint32_t seed, int zero) {
  uint32_t w[4];
  for (int i = 0; i < 4; ++i) {
    seed = seed * 1664525u + 1013904223u;
    uint32_t lo = 0x3c00u + ((seed >> 8) & 0x3ff), hi = 0xbc00u + ((seed >> 20) & 0x3ff);   // ~ +-0.01 .. +-0.03, random mantissas
    w[i] = zero ? 0u : (lo | (hi << 16));
  }
  return __builtin_bit_cast(bf16x8, uint4{w[0], w[1], w[2], w[3]});
}

__global__ __launch_bounds__(512) void k16(float* out, int iters, int zero) {
  bf16x8 a[4], b[8];
  for (int i = 0; i < 4; ++i) a[i] = mk(threadIdx.x * 31 + i + blockIdx.x * 977, zero);
  for (int j = 0; j < 8; ++j) b[j] = mk(threadIdx.x * 17 + j * 5 + blockIdx.x * 131, zero);
  f32x4 acc[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ __launch_bounds__(512) void k32(float* out, int iters, int zero) {
  bf16x8 a[2], b[4];
  for (int i = 0; i < 2; ++i) a[i] = mk(threadIdx.x * 31 + i + blockIdx.x * 977, zero);
  for (int j = 0; j < 4; ++j) b[j] = mk(threadIdx.x * 17 + j * 5 + blockIdx.x * 131, zero);
  f32x16 acc[2][4];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x16{0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)       // same flops per iteration as k16: 16 x 32768 = 32 x 16384
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  float* out; hipMalloc(&out, 1024 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 200000, grid = 256;
  for (int zero = 0; zero < 2; ++zero)
    for (int shape = 0; shape < 2; ++shape) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (shape == 0) k16<<<grid, 512>>>(out, iters, zero); else k32<<<grid, 512>>>(out, iters, zero);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)grid * 8 * iters * 32 * 16384.0;
        if (rep) printf("%s %s: %.1f ms  %.0f TFLOP/s\n", shape ? "32x32x16" : "16x16x32", zero ? "zeros " : "random", ms, flops / ms / 1e9);
      }
    }
  return 0;
}
