// From the bare two-stream kernel of tools/stream_x_bench.hip (W from HBM by LDS-DMA on waves 0-3, x from L2 by LDS-DMA on waves 4-7,
// one barrier per 64-column chunk) towards a GEMM, one ingredient at a time -- where does the weight rate drop?
//   LEVEL 0: the two streams only      LEVEL 1: + every wave's fragment reads (ds_read_b128, swizzled rows: conflict-free)
//   LEVEL 2: + the MFMAs (16x16x32 bf16), M = 128 rows of x, BN = 64 * RT weight rows per chunk, waves tiled 2 (x) x 4 (W)
// The result is garbage-in / garbage-out (no epilogue): only the rates matter.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_mm_bench.hip -o tools/stream_mm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define GL_AS1(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_AS3(p) ((__attribute__((address_space(3))) void*)(p))
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
  else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
  else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if constexpr (N == 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
  else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if constexpr (N == 21) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
  else if constexpr (N == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
  else if constexpr (N == 23) asm volatile("s_waitcnt vmcnt(23)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if constexpr (N == 25) asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
  else if constexpr (N == 26) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
  else if constexpr (N == 27) asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
  else if constexpr (N == 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
  else if constexpr (N == 29) asm volatile("s_waitcnt vmcnt(29)" ::: "memory");
  else if constexpr (N == 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
  else if constexpr (N == 31) asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
  else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  else if constexpr (N == 33) asm volatile("s_waitcnt vmcnt(33)" ::: "memory");
  else if constexpr (N == 34) asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
  else if constexpr (N == 35) asm volatile("s_waitcnt vmcnt(35)" ::: "memory");
  else if constexpr (N == 36) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
  else if constexpr (N == 37) asm volatile("s_waitcnt vmcnt(37)" ::: "memory");
  else if constexpr (N == 38) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
  else if constexpr (N == 39) asm volatile("s_waitcnt vmcnt(39)" ::: "memory");
  else if constexpr (N == 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
  else if constexpr (N == 41) asm volatile("s_waitcnt vmcnt(41)" ::: "memory");
  else if constexpr (N == 42) asm volatile("s_waitcnt vmcnt(42)" ::: "memory");
  else if constexpr (N == 43) asm volatile("s_waitcnt vmcnt(43)" ::: "memory");
  else if constexpr (N == 44) asm volatile("s_waitcnt vmcnt(44)" ::: "memory");
  else if constexpr (N == 45) asm volatile("s_waitcnt vmcnt(45)" ::: "memory");
  else if constexpr (N == 46) asm volatile("s_waitcnt vmcnt(46)" ::: "memory");
  else if constexpr (N == 47) asm volatile("s_waitcnt vmcnt(47)" ::: "memory");
  else if constexpr (N == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
  else static_assert(N < 0, "imm");
}

// RT: weight row blocks (16 rows) per wave -> BN = 64 * RT rows per chunk, W stage = BN * 128 B; x stage = 128 rows * 128 B = 16 KB
template <int RT, int DW, int DX, int LEVEL, int ORDER = 0>
__global__ __launch_bounds__(512) void mm_kernel(const char* __restrict__ W, long bytes_per_wg, const char* __restrict__ X, long x_bytes, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BN = 64 * RT, WST = BN * 128, XST = 128 * 128, WPER = WST / 4096, XPER = 4, MTW = 4;
  char* wring = smem;
  char* xring = smem + (DW + 1) * WST;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loads_w = wave < 4;
  const int w4 = wave & 3;
  const int mh = wave >> 2, wn = wave & 3;          // 2 x 4 tiling: x rows [64 mh, 64 mh + 64), W row blocks wn * RT ... + RT
  const int l15 = lane & 15, lq = lane >> 4;
  const char* wb = W + (long)blockIdx.x * bytes_per_wg;
  const long steps = bytes_per_wg / WST;
  // a DMA instruction lands 8 rows x 128 B; lane i brings piece (i % 8) ^ (row % 8) of row i / 8, so that piece p of row r sits in slot p ^ (r % 8)
  const int drow = lane >> 3, dpiece = (lane & 7) ^ (drow & 7);
  auto issue_w = [&](long s) {
    char* st = wring + (s % (DW + 1)) * WST;
#pragma unroll
    for (int j = 0; j < WPER; ++j) {
      const int q = w4 * WPER + j;                   // rows [8 q, 8 q + 8) of the chunk
      const char* src = wb + s * WST + (long)(q * 8 + drow) * 128 + dpiece * 16;
      __builtin_amdgcn_global_load_lds(GL_AS1(src), LDS_AS3(st + q * 1024), 16, 0, 2);
    }
  };
  auto issue_x = [&](long s) {
    char* st = xring + (s % (DX + 1)) * XST;
#pragma unroll
    for (int j = 0; j < XPER; ++j) {
      const int q = w4 * XPER + j;
      const long off = ((s * XST + (long)(q * 8 + drow) * 128 + dpiece * 16)) % x_bytes;
      __builtin_amdgcn_global_load_lds(GL_AS1(X + off), LDS_AS3(st + q * 1024), 16, 0, 0);
    }
  };
  f32x4 acc[RT][MTW];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int p = 0; p < MTW; ++p) acc[t][p] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (loads_w) { for (long s = 0; s < DW && s < steps; ++s) issue_w(s); }
  else { for (long s = 0; s < DX && s < steps; ++s) issue_x(s); }
  for (long s = 0; s < steps; ++s) {
    if constexpr (ORDER == 0) {
      if (loads_w) {
        if (s + DW < steps) { issue_w(s + DW); wait_vm<DW * WPER>(); } else wait_vm<0>();
      } else {
        if (s + DX < steps) { issue_x(s + DX); wait_vm<DX * XPER>(); } else wait_vm<0>();
      }
      __builtin_amdgcn_s_barrier();
    } else {
      // the slot of chunk s + DW is the one chunk s - 1 was read from: refill it only behind the barrier every wave passes after those reads
      if (loads_w) { if (s + DW <= steps) wait_vm<(DW - 1) * WPER>(); else wait_vm<0>(); }
      else { if (s + DX <= steps) wait_vm<(DX - 1) * XPER>(); else wait_vm<0>(); }
      __builtin_amdgcn_s_barrier();
      if (loads_w) { if (s + DW < steps) issue_w(s + DW); }
      else { if (s + DX < steps) issue_x(s + DX); }
    }
    if constexpr (LEVEL >= 1) {
      const char* wt = wring + (s % (DW + 1)) * WST;
      const char* xt = xring + (s % (DX + 1)) * XST;
      bf16x8 af[2][RT], bq[2][MTW];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const int row = (wn * RT + t) * 16 + l15;
          af[ks][t] = __builtin_bit_cast(bf16x8, *(const uint4*)(wt + row * 128 + (((4 * ks + lq) ^ (row & 7)) << 4)));
        }
#pragma unroll
        for (int p = 0; p < MTW; ++p) {
          const int row = 16 * (mh * MTW + p) + l15;
          bq[ks][p] = __builtin_bit_cast(bf16x8, *(const uint4*)(xt + row * 128 + (((4 * ks + lq) ^ (row & 7)) << 4)));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (LEVEL >= 2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int p = 0; p < MTW; ++p)
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][t], bq[ks][p], acc[t][p], 0, 0, 0);
      } else {
        // keep the reads alive without matrix work
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[t][0][0] += (float)af[ks][t][0];
#pragma unroll
          for (int p = 0; p < MTW; ++p) acc[0][p][1] += (float)bq[ks][p][0];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float tot = 0.f;
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int p = 0; p < MTW; ++p) tot += acc[t][p][0] + acc[t][p][1] + acc[t][p][2] + acc[t][p][3];
  if (tot == 12345.678f) sink[0] = tot;
}

// Variant B: waves 0-3 ONLY load W (LDS-DMA, ring of R slots of BN x 128 B, D chunks in flight, refilled two steps behind their
// last read: no race); waves 4-7 ONLY compute: 2 x 2 tiling of (16 MT) x BN, each takes its x fragments straight from L2 into
// VGPRs (one chunk ahead) -- x never touches LDS -- and its W fragments from the ring.  MT: x row blocks (8: M = 128, 4: M = 64).
template <int MT, int BN, int D, int R>
__global__ __launch_bounds__(512) void mmb_kernel(const char* __restrict__ W, long bytes_per_wg, const char* __restrict__ X, long x_ld, long x_bytes, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WST = BN * 128, WPER = WST / 4096;
  constexpr int MH = MT >= 8 ? 2 : 1, WN = 4 / MH;        // compute waves: MH along x rows, WN along W rows
  constexpr int MTW = MT / MH, RT = BN / 16 / WN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave < 4;
  const int w4 = wave & 3;
  const int l15 = lane & 15, lq = lane >> 4;
  const char* wb = W + (long)blockIdx.x * bytes_per_wg;
  const long steps = bytes_per_wg / WST;
  const int drow = lane >> 3, dpiece = (lane & 7) ^ (drow & 7);
  f32x4 acc[RT][MTW];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int p = 0; p < MTW; ++p) acc[t][p] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (loader) {
    auto issue_w = [&](long s) {
      char* st = smem + (s % R) * WST;
#pragma unroll
      for (int j = 0; j < WPER; ++j) {
        const int q = w4 * WPER + j;
        const char* src = wb + s * WST + (long)(q * 8 + drow) * 128 + dpiece * 16;
        __builtin_amdgcn_global_load_lds(GL_AS1(src), LDS_AS3(st + q * 1024), 16, 0, 2);
      }
    };
    for (long s = 0; s < D && s < steps; ++s) issue_w(s);
    for (long s = 0; s < steps; ++s) {
      if (s + D < steps) { issue_w(s + D); wait_vm<D * WPER>(); } else wait_vm<0>();      // slot (s + D) % R was read at step s + D - R <= s - 2
      __builtin_amdgcn_s_barrier();
    }
  } else {
    const int mh = w4 / WN, wn = w4 % WN;
    // x[16 MT][K] row-major in L2: fragment (block p, half ks) of chunk s = rows 16 (mh MTW + p) + l15, columns 64 s + 32 ks + 8 lq
    auto xaddr = [&](long s, int p, int ks) {
      const long off = ((long)(16 * (mh * MTW + p) + l15) * x_ld + s * 128 + ks * 64 + lq * 16) % x_bytes;
      return (const uint4*)(X + off);
    };
    uint4 xa[2][MTW], xb[2][MTW];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < MTW; ++p) xa[ks][p] = *xaddr(0, p, ks);
    auto step = [&](long s, uint4 (&cur)[2][MTW], uint4 (&nxt)[2][MTW]) {
      if (s + 1 < steps) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int p = 0; p < MTW; ++p) nxt[ks][p] = *xaddr(s + 1, p, ks);
        wait_vm<2 * MTW>();
      } else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      const char* wt = smem + (s % R) * WST;
      bf16x8 af[2][RT];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const int row = (wn * RT + t) * 16 + l15;
          af[ks][t] = __builtin_bit_cast(bf16x8, *(const uint4*)(wt + row * 128 + (((4 * ks + lq) ^ (row & 7)) << 4)));
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p = 0; p < MTW; ++p)
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][t], __builtin_bit_cast(bf16x8, cur[ks][p]), acc[t][p], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (MT == 8 && BN == 128) {
      // three chunks of x ahead: four register sets in rotation (the L2 round trip is ~1.2 us, a chunk is due every ~0.6 us)
      uint4 xs[4][2][MTW];
      auto ld = [&](long c, int set) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int p = 0; p < MTW; ++p) xs[set][ks][p] = *xaddr(c < steps ? c : steps - 1, p, ks);
      };
      auto go = [&](long c, int set) {
        ld(c + 3, (set + 3) & 3);
        wait_vm<3 * 2 * MTW>();
        __builtin_amdgcn_s_barrier();
        const char* wt = smem + (c % R) * WST;
        bf16x8 af[2][RT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int t = 0; t < RT; ++t) {
            const int row = (wn * RT + t) * 16 + l15;
            af[ks][t] = __builtin_bit_cast(bf16x8, *(const uint4*)(wt + row * 128 + (((4 * ks + lq) ^ (row & 7)) << 4)));
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int p = 0; p < MTW; ++p)
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][t], __builtin_bit_cast(bf16x8, xs[set][ks][p]), acc[t][p], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      ld(0, 0); ld(1, 1); ld(2, 2);
      long c = 0;
      for (; c + 3 < steps; c += 4) { go(c, 0); go(c + 1, 1); go(c + 2, 2); go(c + 3, 3); }
      for (int r = 0; c < steps; ++c, ++r) go(c, r);
    } else {
    long s = 0;
    for (; s + 1 < steps; s += 2) { step(s, xa, xb); step(s + 1, xb, xa); }
    if (s < steps) step(s, xa, xb);
    }
  }
  float tot = 0.f;
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int p = 0; p < MTW; ++p) tot += acc[t][p][0] + acc[t][p][1] + acc[t][p][2] + acc[t][p][3];
  if (tot == 12345.678f) sink[0] = tot;
}

template <int MT, int BN, int D, int R>
static void runb(const char* name, const char* W, size_t total, const char* X, long x_bytes, float* sink, int n_wg, hipStream_t st) {
  constexpr int WST = BN * 128;
  const size_t lds = (size_t)R * WST;
  if (lds > 160 * 1024) { printf("%s: %zu KB of LDS\n", name, lds / 1024); return; }
  CK(hipFuncSetAttribute((const void*)mmb_kernel<MT, BN, D, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long per = (long)(total / n_wg) / WST * WST;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((mmb_kernel<MT, BN, D, R>), dim3(n_wg), dim3(512), lds, st, W, per, X, (long)8192, x_bytes, sink);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double steps = (double)per / WST;
  printf("B: %-31s M %3d BN %3d, W %d x %2d KB in flight of %d slots, LDS %3zu KB, %3d WGs: %7.1f us  W %5.2f TB/s  %.3f us per chunk  (MFMA %.0f TFLOP/s)\n", name, 16 * MT, BN,
         D, WST / 1024, R, lds / 1024, n_wg, best * 1e3, (double)per * n_wg / best / 1e9, best * 1e3 / steps, 2.0 * 16 * MT * BN * 64 * steps * n_wg / best / 1e9);
}

// Variant C: waves 0-1 load W, waves 2-3 load x (both by LDS-DMA, rings refilled two steps behind their last read), waves 4-7 only
// compute: 2 x 2 (M = 128) or 1 x 4 (M = 64) tiling -- 64 x 64 / 64 x 32 outputs per wave, the fewest LDS bytes per weight byte.
template <int MT, int BN, int DW, int RW, int DX, int RX, int PIPE = 0>
__global__ __launch_bounds__(512) void mmc_kernel(const char* __restrict__ W, long bytes_per_wg, const char* __restrict__ X, long x_bytes, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WST = BN * 128, XST = 16 * MT * 128, WPER = WST / 2048, XPER = XST / 2048;
  constexpr int MH = MT >= 8 ? 2 : 1, WN = 4 / MH;
  constexpr int MTW = MT / MH, RT = BN / 16 / WN;
  char* wring = smem;
  char* xring = smem + RW * WST;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const char* wb = W + (long)blockIdx.x * bytes_per_wg;
  const long steps = bytes_per_wg / WST;
  const int drow = lane >> 3, dpiece = (lane & 7) ^ (drow & 7);
  f32x4 acc[RT][MTW];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int p = 0; p < MTW; ++p) acc[t][p] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (wave < 2) {
    auto issue_w = [&](long s) {
      char* st = wring + (s % RW) * WST;
#pragma unroll
      for (int j = 0; j < WPER; ++j) {
        const int q = wave * WPER + j;
        const char* src = wb + s * WST + (long)(q * 8 + drow) * 128 + dpiece * 16;
        __builtin_amdgcn_global_load_lds(GL_AS1(src), LDS_AS3(st + q * 1024), 16, 0, 2);
      }
    };
    for (long s = 0; s < DW && s < steps; ++s) issue_w(s);
    for (long s = 0; s < steps; ++s) {
      if (s + DW < steps) { issue_w(s + DW); wait_vm<DW * WPER>(); } else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
    }
  } else if (wave < 4) {
    auto issue_x = [&](long s) {
      char* st = xring + (s % RX) * XST;
#pragma unroll
      for (int j = 0; j < XPER; ++j) {
        const int q = (wave - 2) * XPER + j;
        const long off = (s * XST + (long)(q * 8 + drow) * 128 + dpiece * 16) % x_bytes;
        __builtin_amdgcn_global_load_lds(GL_AS1(X + off), LDS_AS3(st + q * 1024), 16, 0, 0);
      }
    };
    for (long s = 0; s < DX && s < steps; ++s) issue_x(s);
    for (long s = 0; s < steps; ++s) {
      if (s + DX < steps) { issue_x(s + DX); wait_vm<DX * XPER>(); } else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
    }
  } else {
    const int cw = wave - 4, mh = cw / WN, wn = cw % WN;
    if constexpr (PIPE) {
      // software pipeline: the fragment reads of chunk s run under the MFMAs of chunk s - 1 (two register sets); the reads of a chunk
      // are complete before the next barrier, so a slot is free one barrier after its chunk landed, as in the plain loop
      bf16x8 fa[2][2][RT], fb[2][2][MTW];
      auto reads = [&](long c, int set) {
        const char* wt = wring + (c % RW) * WST;
        const char* xt = xring + (c % RX) * XST;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int t = 0; t < RT; ++t) {
            const int row = (wn * RT + t) * 16 + l15;
            fa[set][ks][t] = __builtin_bit_cast(bf16x8, *(const uint4*)(wt + row * 128 + (((4 * ks + lq) ^ (row & 7)) << 4)));
          }
#pragma unroll
          for (int p = 0; p < MTW; ++p) {
            const int row = 16 * (mh * MTW + p) + l15;
            fb[set][ks][p] = __builtin_bit_cast(bf16x8, *(const uint4*)(xt + row * 128 + (((4 * ks + lq) ^ (row & 7)) << 4)));
          }
        }
      };
      auto mfmas = [&](int set) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int p = 0; p < MTW; ++p)
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[set][ks][t], fb[set][ks][p], acc[t][p], 0, 0, 0);
      };
      long c = 0;
      __builtin_amdgcn_s_barrier();
      reads(0, 0);
      for (c = 1; c + 1 < steps; c += 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        reads(c, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        reads(c + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (c < steps) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        reads(c, 1);
        mfmas(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfmas(1);
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfmas(0);
      }
    } else
    for (long s = 0; s < steps; ++s) {
      __builtin_amdgcn_s_barrier();
      const char* wt = wring + (s % RW) * WST;
      const char* xt = xring + (s % RX) * XST;
      bf16x8 af[2][RT], bq[2][MTW];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const int row = (wn * RT + t) * 16 + l15;
          af[ks][t] = __builtin_bit_cast(bf16x8, *(const uint4*)(wt + row * 128 + (((4 * ks + lq) ^ (row & 7)) << 4)));
        }
#pragma unroll
        for (int p = 0; p < MTW; ++p) {
          const int row = 16 * (mh * MTW + p) + l15;
          bq[ks][p] = __builtin_bit_cast(bf16x8, *(const uint4*)(xt + row * 128 + (((4 * ks + lq) ^ (row & 7)) << 4)));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p = 0; p < MTW; ++p)
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][t], bq[ks][p], acc[t][p], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float tot = 0.f;
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int p = 0; p < MTW; ++p) tot += acc[t][p][0] + acc[t][p][1] + acc[t][p][2] + acc[t][p][3];
  if (tot == 12345.678f) sink[0] = tot;
}

template <int MT, int BN, int DW, int RW, int DX, int RX, int PIPE = 0>
static void runc(const char* name, const char* W, size_t total, const char* X, long x_bytes, float* sink, int n_wg, hipStream_t st) {
  constexpr int WST = BN * 128, XST = 16 * MT * 128;
  const size_t lds = (size_t)RW * WST + (size_t)RX * XST;
  if (lds > 160 * 1024) { printf("%s: %zu KB of LDS\n", name, lds / 1024); return; }
  CK(hipFuncSetAttribute((const void*)mmc_kernel<MT, BN, DW, RW, DX, RX, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long per = (long)(total / n_wg) / WST * WST;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((mmc_kernel<MT, BN, DW, RW, DX, RX, PIPE>), dim3(n_wg), dim3(512), lds, st, W, per, X, x_bytes, sink);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double steps = (double)per / WST;
  printf("C%d: %-31s M %3d BN %3d, W %d of %d x %2d KB, x %d of %d x %2d KB, LDS %3zu KB, %3d WGs: %7.1f us  W %5.2f TB/s  %.3f us per chunk  (MFMA %.0f TFLOP/s)\n", PIPE, name, 16 * MT, BN,
         DW, RW, WST / 1024, DX, RX, XST / 1024, lds / 1024, n_wg, best * 1e3, (double)per * n_wg / best / 1e9, best * 1e3 / steps, 2.0 * 16 * MT * BN * 64 * steps * n_wg / best / 1e9);
}

template <int RT, int DW, int DX, int LEVEL, int ORDER = 0>
static void run(const char* name, const char* W, size_t total, const char* X, long x_bytes, float* sink, int n_wg, hipStream_t st) {
  constexpr int BN = 64 * RT, WST = BN * 128, XST = 128 * 128;
  const size_t lds = (size_t)(DW + 1) * WST + (size_t)(DX + 1) * XST;
  if (lds > 160 * 1024) { printf("%s: %zu KB of LDS\n", name, lds / 1024); return; }
  CK(hipFuncSetAttribute((const void*)mm_kernel<RT, DW, DX, LEVEL, ORDER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long per = (long)(total / n_wg) / WST * WST;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((mm_kernel<RT, DW, DX, LEVEL, ORDER>), dim3(n_wg), dim3(512), lds, st, W, per, X, x_bytes, sink);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double steps = (double)per / WST;
  printf("%-34s BN %3d, W %d x %2d KB + x %d x 16 KB in flight, LDS %3zu KB, level %d order %d: %7.1f us  W %5.2f TB/s  %.3f us per chunk  (MFMA %.0f TFLOP/s)\n", name, BN, DW,
         WST / 1024, DX, lds / 1024, LEVEL, ORDER, best * 1e3, (double)per * n_wg / best / 1e9, best * 1e3 / steps, LEVEL >= 2 ? 2.0 * 128 * BN * 64 * steps * n_wg / best / 1e9 : 0.0);
}

int main() {
  const size_t total = (size_t)1 << 30;
  char* W; CK(hipMalloc(&W, total)); CK(hipMemset(W, 1, total));
  const long x_bytes = 1 << 20;     // x[128][4096] bf16: L2-resident
  char* X; CK(hipMalloc(&X, x_bytes)); CK(hipMemset(X, 2, x_bytes));
  float* sink; CK(hipMalloc(&sink, 64));
  hipStream_t st; CK(hipStreamCreate(&st));
  const int n = 256;
  run<2, 4, 3, 0>("BN 128 streams", W, total, X, x_bytes, sink, n, st);
  run<2, 4, 3, 1>("BN 128 + fragment reads", W, total, X, x_bytes, sink, n, st);
  run<2, 4, 3, 2>("BN 128 + MFMA", W, total, X, x_bytes, sink, n, st);
  run<2, 6, 2, 2>("BN 128 + MFMA, deeper W", W, total, X, x_bytes, sink, n, st);
  run<4, 2, 2, 0>("BN 256 streams", W, total, X, x_bytes, sink, n, st);
  run<4, 2, 2, 1>("BN 256 + fragment reads", W, total, X, x_bytes, sink, n, st);
  run<4, 2, 2, 2>("BN 256 + MFMA", W, total, X, x_bytes, sink, n, st);
  run<4, 3, 1, 2>("BN 256 + MFMA, 96 KB of W", W, total, X, x_bytes, sink, n, st);
  run<4, 3, 1, 2, 1>("BN 256 + MFMA, safe order", W, total, X, x_bytes, sink, n, st);
  run<4, 3, 1, 2, 1>("same, 224 workgroups", W, total, X, x_bytes, sink, 224, st);
  run<4, 3, 1, 2, 0>("racy order, 224 workgroups", W, total, X, x_bytes, sink, 224, st);
  run<2, 6, 2, 2, 1>("BN 128 + MFMA, safe, deep", W, total, X, x_bytes, sink, n, st);
  run<3, 3, 2, 2>("BN 192 + MFMA", W, total, X, x_bytes, sink, n, st);
  run<3, 4, 1, 2>("BN 192 + MFMA, 96 KB of W", W, total, X, x_bytes, sink, n, st);
  runc<8, 128, 4, 6, 2, 4, 1>("M 128 / BN 128 pipelined", W, total, X, x_bytes, sink, n, st);
  runc<8, 128, 5, 7, 1, 3, 1>("M 128 / BN 128 pipelined, W deeper", W, total, X, x_bytes, sink, n, st);
  runc<8, 128, 4, 6, 2, 4, 1>("M 128 / BN 128 pipelined, 224", W, total, X, x_bytes, sink, 224, st);
  runc<4, 128, 6, 8, 2, 4, 1>("M 64 / BN 128 pipelined", W, total, X, x_bytes, sink, n, st);
  runc<2, 128, 6, 8, 2, 4, 1>("M 32 / BN 128 pipelined", W, total, X, x_bytes, sink, n, st);
  runc<8, 128, 4, 6, 2, 4>("M 128 / BN 128", W, total, X, x_bytes, sink, n, st);
  runc<8, 128, 5, 7, 1, 3>("M 128 / BN 128, W deeper", W, total, X, x_bytes, sink, n, st);
  runc<8, 128, 4, 6, 2, 4>("M 128 / BN 128, 224 WGs", W, total, X, x_bytes, sink, 224, st);
  runc<8, 256, 2, 4, 0, 2>("M 128 / BN 256 (x not ahead)", W, total, X, x_bytes, sink, n, st);
  runc<4, 128, 6, 8, 2, 4>("M 64 / BN 128", W, total, X, x_bytes, sink, n, st);
  runc<4, 256, 2, 4, 2, 4>("M 64 / BN 256", W, total, X, x_bytes, sink, n, st);
  runc<4, 256, 3, 4, 2, 4>("M 64 / BN 256 racy 3 of 4", W, total, X, x_bytes, sink, n, st);
  runb<8, 128, 6, 8>("M 128 / BN 128, 6 of 8", W, total, X, x_bytes, sink, n, st);
  runb<8, 128, 8, 10>("M 128 / BN 128, 8 of 10", W, total, X, x_bytes, sink, n, st);
  runb<8, 128, 4, 6>("M 128 / BN 128, 4 of 6", W, total, X, x_bytes, sink, n, st);
  runb<8, 128, 8, 10>("M 128 / BN 128, 224 WGs", W, total, X, x_bytes, sink, 224, st);
  runb<8, 256, 3, 5>("M 128 / BN 256, 3 of 5", W, total, X, x_bytes, sink, n, st);
  runb<4, 128, 8, 10>("M 64 / BN 128, 8 of 10", W, total, X, x_bytes, sink, n, st);
  runb<4, 256, 3, 5>("M 64 / BN 256, 3 of 5", W, total, X, x_bytes, sink, n, st);
  return 0;
}
