// Per-CU global -> LDS fill rate: LDS-DMA (global_load_lds_dwordx4) vs register staging (global_load_dwordx4 + ds_write_b128),
// by number of issuing waves and by row segment (128-byte or 64-byte pieces of a row).  Source is a small L2-resident
// buffer (or a big one with BIG=1).  build: hipcc --offload-arch=gfx950 -O3 tools/fill_bench.hip -o tools/fill_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define GL_AS1(p) ((__attribute__((address_space(1))) void*)(p))
#define LDS_AS3(p) ((__attribute__((address_space(3))) void*)(p))

// MODE 0: LDS-DMA; 1: register staging.  SEG: bytes of one row segment (128 or 64).  Each wave moves 1 KB per instruction.
template <int MODE, int SEG, int INFL>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, long row_stride, long span, int iters, int waves, float* sink, int share) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave >= waves) return;
  constexpr int LPR = SEG / 16;                 // lanes per row segment
  // SWZ=1: lane order inside a row segment permuted by the row (the source-side XOR swizzle of the GEMM kernels); SWZ=2: 32-byte pairs kept
  const int swz_mode = share >> 16; share &= 0xffff;
  const int row = lane / LPR;
  int chunk = lane % LPR;
  if (swz_mode == 1) chunk ^= (row >> 1) & (LPR - 1);
  if (swz_mode == 2) chunk ^= ((row >> 1) & (LPR / 2 - 1)) << 1;
  const int col = chunk * 16;
  // this wave's 512-row window starts somewhere inside the span (power of two); no division in the loop
  const char* base = src + ((((long)((blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / share)) * 8 + wave) * 512 * row_stride) & (span - 1) & ~15L) + (long)row * row_stride + col;
  char* lds = smem + wave * (INFL * 1024) + (swz_mode == 3 ? 65536 : 0);   // SWZ=3: destinations above 64 KB
  uint4 regs[INFL];
  long off = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < INFL; ++u) {
      const char* p;
      if (swz_mode == 4) {   // SWZ=4 (INFL = 8): GEMM-like streaming -- this wave owns 64 rows, one 128-B line of each per iteration,
                             // k advances one line per iteration, a new 512-row window every 112 iterations
        const long win = (it / 112) * 256L * 4096 * row_stride;
        p = src + ((((long)((blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / share)) * 8 + wave) * 512 * row_stride + win) & (span - 1) & ~127L)
            + (long)((u & 7) * 8 + row) * row_stride + (long)(((it * (INFL / 8 > 0 ? INFL / 8 : 1)) + (u >> 3)) % 112) * 128 + col;
      } else p = base + (long)(((it * INFL + u) & 63) * (64 / LPR)) * row_stride;
      if (MODE == 0 || MODE == 2) __builtin_amdgcn_global_load_lds(GL_AS1(p), LDS_AS3(lds + u * 1024), 16, 0, 0);
      else regs[u] = *(const uint4*)p;
    }
    if (MODE == 2) {   // the GEMM loop's sync skeleton: half the pieces, barriers, the other half, counted wait, barrier
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else {
#pragma unroll
      for (int u = 0; u < INFL; ++u) *(uint4*)(lds + u * 1024 + lane * 16) = regs[u];
    }
    off += 1;
  }
  __syncthreads();
  if (tid == 0) sink[blockIdx.x] = (float)smem[off & 1023];
}

template <int MODE, int SEG, int INFL>
void run(const char* src, long stride, long span, int waves, const char* label) {
  float* sink; hipMalloc(&sink, 4096);
  const int iters = 4000;
  hipFuncSetAttribute((const void*)k<MODE, SEG, INFL>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int share = (getenv("SHARE") ? atoi(getenv("SHARE")) : 1) | ((getenv("SWZ") ? atoi(getenv("SWZ")) : 0) << 16);
  k<MODE, SEG, INFL><<<256, 512, 131072>>>(src, stride, span, iters, waves, sink, share);
  hipEventRecord(e0);
  k<MODE, SEG, INFL><<<256, 512, 131072>>>(src, stride, span, iters, waves, sink, share);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double bytes = 256.0 * waves * iters * INFL * 1024.0;
  printf("%-44s waves=%d in-flight/wave=%d KB: %7.1f GB/s per CU  (%.2f TB/s chip)\n", label, waves, INFL, bytes / ms / 1e6 / 256, bytes / ms / 1e9);
  hipFree(sink);
}

int main() {
  const long stride = 14336;                        // a K = 7168 bf16 row
  const long span = getenv("SPAN_MB") ? (long)atoi(getenv("SPAN_MB")) * 1024 * 1024 : (getenv("BIG") ? (long)2048 * 1024 * 1024 : (long)48 * 1024 * 1024);
  char* src; hipMalloc(&src, span + (16 << 20)); hipMemset(src, 1, span + (16 << 20));
  run<0, 128, 4>(src, stride, span, 8, "LDS-DMA, 128-B segments");
  run<0, 128, 8>(src, stride, span, 8, "LDS-DMA, 128-B segments");
  run<0, 128, 8>(src, stride, span, 4, "LDS-DMA, 128-B segments");
  run<0, 64, 8>(src, stride, span, 8, "LDS-DMA, 64-B segments");
  run<2, 128, 8>(src, stride, span, 8, "LDS-DMA + vmcnt(4) + 4 barriers per 8 KB");
  run<0, 128, 16>(src, stride, span, 8, "LDS-DMA, 128-B segments");
  run<0, 128, 12>(src, stride, span, 8, "LDS-DMA, 128-B segments");
  run<1, 128, 4>(src, stride, span, 8, "global_load + ds_write, 128-B segments");
  run<1, 128, 8>(src, stride, span, 8, "global_load + ds_write, 128-B segments");
  run<1, 128, 8>(src, stride, span, 4, "global_load + ds_write, 128-B segments");
  run<1, 64, 8>(src, stride, span, 8, "global_load + ds_write, 64-B segments");
  return 0;
}
