// Does the activation operand of a skinny GEMM slow the weight stream of its workgroup, and does the PATH it takes matter?
// One 512-thread workgroup per CU: waves 0-3 stream W from HBM into an LDS ring by LDS-DMA (as lia_chain / lia_gemm_skinny2 do), waves
// 4-7 fetch XKB KB of "x" per step from a small L2-resident buffer -- by LDS-DMA too (XMODE 1), through VGPRs + ds_write (XMODE 2), or
// not at all (XMODE 0).  No MFMA: only the two streams.  Reported: the W rate.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_x_bench.hip -o tools/stream_x_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define GL_AS1(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_AS3(p) ((__attribute__((address_space(3))) void*)(p))
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  else static_assert(N < 0, "imm");
}

// W step = 4 waves x WPER KB; x step = 4 waves x XPER KB; DW / DX steps in flight
template <int WPER, int DW, int XPER, int DX, int XMODE>
__global__ __launch_bounds__(512) void sx_kernel(const char* __restrict__ W, long bytes_per_wg, const char* __restrict__ X, long x_bytes, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WST = 4 * WPER * 1024, XST = 4 * (XPER > 0 ? XPER : 1) * 1024;
  char* wring = smem;
  char* xring = smem + (DW + 1) * WST;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loads_w = wave < 4;
  const int w4 = wave & 3;
  const char* wb = W + (long)blockIdx.x * bytes_per_wg;
  const long steps = bytes_per_wg / WST;
  unsigned acc = 0;
  auto issue_w = [&](long s) {
    char* st = wring + (s % (DW + 1)) * WST;
#pragma unroll
    for (int j = 0; j < WPER; ++j) {
      const int q = w4 * WPER + j;
      // eight rows x 128 B per instruction (row stride 8 KB), as the GEMMs read a [N][K] matrix
      const char* src = wb + s * WST + (long)q * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds(GL_AS1(src), LDS_AS3(st + q * 1024), 16, 0, 2);
    }
  };
  u32x4 xr[XPER > 0 ? XPER * (DX + 1) : 1];
  auto x_src = [&](long s, int j) {
    const long off = ((s * 4 * XPER + w4 * XPER + j) * 1024 + lane * 16) % x_bytes;
    return X + off;
  };
  auto issue_x = [&](long s) {
    if constexpr (XMODE == 1) {
      char* st = xring + (s % (DX + 1)) * XST;
#pragma unroll
      for (int j = 0; j < XPER; ++j) __builtin_amdgcn_global_load_lds(GL_AS1(x_src(s, j)), LDS_AS3(st + (w4 * XPER + j) * 1024), 16, 0, 0);
    }
  };
  if (loads_w) { for (long s = 0; s < DW && s < steps; ++s) issue_w(s); }
  else if (XMODE == 1) { for (long s = 0; s < DX && s < steps; ++s) issue_x(s); }
  for (long s = 0; s < steps; ++s) {
    if (loads_w) {
      if (s + DW < steps) { issue_w(s + DW); wait_vm<DW * WPER>(); } else wait_vm<0>();
    } else if constexpr (XMODE == 1) {
      if (s + DX < steps) { issue_x(s + DX); wait_vm<DX * XPER>(); } else wait_vm<0>();
    } else if constexpr (XMODE == 2) {
      // through VGPRs: load this step's XPER KB, wait, write them to LDS (one step deep: the loads of step s overlap the W stream anyway)
      u32x4 v[XPER];
#pragma unroll
      for (int j = 0; j < XPER; ++j) v[j] = *(const u32x4*)x_src(s, j);
      char* st = xring + (s % (DX + 1)) * XST;
#pragma unroll
      for (int j = 0; j < XPER; ++j) *(u32x4*)(st + (w4 * XPER + j) * 1024 + lane * 16) = v[j];
    }
    __builtin_amdgcn_s_barrier();
    if (s == steps - 1) acc += *(unsigned*)(wring + lane * 4) + *(unsigned*)(xring + lane * 4);
  }
  if (acc == 0x12345u) sink[0] = acc;
  (void)xr;
}

template <int WPER, int DW, int XPER, int DX, int XMODE>
static void run(const char* name, const char* W, size_t total, const char* X, long x_bytes, unsigned* sink, hipStream_t st) {
  constexpr int WST = 4 * WPER * 1024, XST = 4 * (XPER > 0 ? XPER : 1) * 1024;
  const size_t lds = (size_t)(DW + 1) * WST + (size_t)(DX + 1) * XST;
  CK(hipFuncSetAttribute((const void*)sx_kernel<WPER, DW, XPER, DX, XMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int n_wg = 256;
  const long per = (long)(total / n_wg) / WST * WST;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((sx_kernel<WPER, DW, XPER, DX, XMODE>), dim3(n_wg), dim3(512), lds, st, W, per, X, x_bytes, sink);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("%-58s W step %2d KB x %d in flight, x %2d KB/step mode %d, LDS %3zu KB: %7.1f us  W %5.2f TB/s (%.1f GB/s per CU), x %5.2f TB/s\n", name, WST / 1024, DW,
         XPER * 4, XMODE, lds / 1024, best * 1e3, (double)per * n_wg / best / 1e9, (double)per / best / 1e6, XMODE ? (double)per / WST * (XPER * 4096.0) * n_wg / best / 1e9 : 0.0);
}

int main() {
  const size_t total = (size_t)1 << 30;
  char* W; CK(hipMalloc(&W, total)); CK(hipMemset(W, 1, total));
  const long x_bytes = 1 << 20;     // x[128][4096] bf16: L2-resident
  char* X; CK(hipMalloc(&X, x_bytes)); CK(hipMemset(X, 2, x_bytes));
  unsigned* sink; CK(hipMalloc(&sink, 64));
  hipStream_t st; CK(hipStreamCreate(&st));
  // M = 128, BN = 128: 16 KB of W and 16 KB of x per 64-column chunk
  run<4, 4, 0, 1, 0>("W only (BN 128)", W, total, X, x_bytes, sink, st);
  run<4, 4, 4, 3, 1>("M 128 / BN 128: x by LDS-DMA", W, total, X, x_bytes, sink, st);
  run<4, 4, 4, 1, 2>("M 128 / BN 128: x through VGPRs", W, total, X, x_bytes, sink, st);
  run<4, 4, 2, 3, 1>("M 64 / BN 128: x by LDS-DMA", W, total, X, x_bytes, sink, st);
  run<4, 4, 2, 1, 2>("M 64 / BN 128: x through VGPRs", W, total, X, x_bytes, sink, st);
  // BN = 256: 32 KB of W per chunk
  run<8, 3, 0, 1, 0>("W only (BN 256)", W, total, X, x_bytes, sink, st);
  run<8, 3, 4, 2, 1>("M 128 / BN 256: x by LDS-DMA", W, total, X, x_bytes, sink, st);
  run<8, 3, 4, 1, 2>("M 128 / BN 256: x through VGPRs", W, total, X, x_bytes, sink, st);
  run<8, 3, 2, 2, 1>("M 64 / BN 256: x by LDS-DMA", W, total, X, x_bytes, sink, st);
  run<8, 3, 2, 1, 2>("M 64 / BN 256: x through VGPRs", W, total, X, x_bytes, sink, st);
  return 0;
}
