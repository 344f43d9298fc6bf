// How long do the 12 ds_read_b128 fragment reads of one k-step take for the 4 waves of a group, alone and beside the
// other group's 32 MFMAs (with and without s_setprio)?  build: hipcc --offload-arch=gfx950 -O3 tools/lds_read_bench.hip -o tools/lds_read_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>   // 0: readers only; 1: + MFMA waves (no prio); 2: + MFMA waves with setprio(1); 3: MFMA waves only
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  for (int i = tid; i < 32768; i += 512) ((uint32_t*)smem)[i] = i * 2654435761u;
  __syncthreads();
  const int wn = wave & 3, wm = wave >> 2;
  const int frag_off = l15 * 64 + ((lq ^ ((-(l15 >> 2)) & 3)) << 4);
  const char* wfrag = smem + wn * 4096 + frag_off;
  const char* xfrag = smem + 16384 + wm * 8192 + frag_off;
  f32x4 acc[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  bf16x8 a[4], b[8];
  for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(wfrag + i * 1024));
  for (int j = 0; j < 8; ++j) b[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(xfrag + j * 1024));
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < 4 && MODE != 3) {
    for (int it = 0; it < iters; ++it) {
      const int off = (it & 3) * 32768;
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(wfrag + off + i * 1024));
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(xfrag + off + j * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
      asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
    }
  } else if (wave >= 4 && MODE != 0) {
    for (int it = 0; it < iters; ++it) {
      if (MODE == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      if (MODE == 2) __builtin_amdgcn_s_setprio(0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0];
  s += (float)a[0][0] + (float)b[0][0];
  sink[blockIdx.x * 512 + tid] = s;
}

int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 64); hipMalloc(&sink, 256 * 512 * 4);
  const int iters = 20000;
  unsigned long long h[8];
#define RUN(M, label)                                                                                          \
  hipFuncSetAttribute((const void*)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);                  \
  k<M><<<256, 512, 131072>>>(out, sink, iters); k<M><<<256, 512, 131072>>>(out, sink, iters);                  \
  hipDeviceSynchronize(); hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);                                        \
  printf("%-34s reader wave: %6.0f cycles/iter   mfma wave: %6.0f cycles/iter\n", label, (double)h[0] / iters, (double)h[4] / iters);
  RUN(0, "12 ds_read_b128, 4 waves alone")
  RUN(3, "32 MFMA, 4 waves alone")
  RUN(1, "readers + MFMA waves")
  RUN(2, "readers + MFMA waves, setprio(1)")
  return 0;
}
