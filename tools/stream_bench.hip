// How fast can one workgroup per CU stream a weight matrix into LDS by LDS-DMA (global_load_lds, nt), and how much does the shape of a
// request matter?  The decode GEMMs read W[N][K] (row-major) tile by tile: a wave instruction = 64 lanes x 16 B = 1 KB, cut into
// PIECE-byte runs of consecutive rows (PIECE = 128: eight rows x one 64-column chunk, what lia_gemm_skinny2 / lia_chain do; 256, 512,
// 1024: longer runs per row).  No compute, no x operand: the ceiling of the weight stream alone.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_bench.hip -o tools/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define GL_AS1(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_AS3(p) ((__attribute__((address_space(3))) void*)(p))

template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  else if constexpr (N == 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
  else if constexpr (N == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
  else static_assert(N < 0, "imm");
}

// Each workgroup (WAVES waves) streams ROWS rows x K columns (bf16) of its own: per step every wave issues PER instructions (1 KB each),
// the workgroup's step = WAVES * PER KB; DEPTH steps are kept in flight (ring of DEPTH + 1 LDS stages).
template <int PIECE, int WAVES, int PER, int DEPTH>
__global__ __launch_bounds__(64 * WAVES) void stream_kernel(const char* __restrict__ W, long row_bytes, int rows, long bytes_per_wg, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = WAVES * PER * 1024;
  constexpr int LPP = PIECE / 16;              // lanes per piece
  constexpr int PPI = 64 / LPP;                // pieces (rows) per instruction
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = W + (long)blockIdx.x * rows * row_bytes;
  // instruction q of a step (q = wave * PER + j) covers rows [q * PPI, q * PPI + PPI) mod rows at column offset col
  const long steps = bytes_per_wg / STAGE;
  const int rows_per_step = WAVES * PER * PPI;
  long col = 0; int row0 = 0;
  auto issue = [&](long s) {
    char* st = smem + (s % (DEPTH + 1)) * STAGE;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int q = wave * PER + j;
      const int row = row0 + q * PPI + lane / LPP;
      const char* src = base + (long)row * row_bytes + col + (lane % LPP) * 16;
      if (nt) __builtin_amdgcn_global_load_lds(GL_AS1(src), LDS_AS3(st + q * 1024), 16, 0, 2);
      else __builtin_amdgcn_global_load_lds(GL_AS1(src), LDS_AS3(st + q * 1024), 16, 0, 0);
    }
    row0 += rows_per_step;
    if (row0 >= rows) { row0 = 0; col += PIECE; }
  };
  for (long s = 0; s < DEPTH && s < steps; ++s) issue(s);
  for (long s = 0; s < steps; ++s) {
    if (s + DEPTH < steps) { issue(s + DEPTH); wait_vm<DEPTH * PER>(); }
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
  }
}

template <int PIECE, int WAVES, int PER, int DEPTH>
static void run(const char* name, const char* W, size_t total_bytes, int n_wg, int nt, hipStream_t st) {
  constexpr int STAGE = WAVES * PER * 1024;
  const size_t lds = (size_t)(DEPTH + 1) * STAGE;
  CK(hipFuncSetAttribute((const void*)stream_kernel<PIECE, WAVES, PER, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int rows = WAVES * PER * (64 / (PIECE / 16));      // one step = all rows once (so a row is PIECE bytes per step)
  const long row_bytes = 8192;                               // K = 4096 bf16
  long per_wg = (long)rows * row_bytes;                      // each workgroup streams its rows x 8 KB
  // repeat the tile so that every workgroup moves ~ total / n_wg bytes
  const long want = (long)(total_bytes / n_wg);
  const int reps = (int)(want / per_wg) > 0 ? (int)(want / per_wg) : 1;
  // (a workgroup's tile is rows x row_bytes; it sweeps it `reps` times at different base offsets by enlarging rows)
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int big_rows = rows * reps;                          // rows per workgroup: rows x reps distinct rows
  const long bytes_per_wg = (long)big_rows * row_bytes;
  if ((size_t)bytes_per_wg * n_wg > total_bytes) { printf("%s: skip\n", name); return; }
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((stream_kernel<PIECE, WAVES, PER, DEPTH>), dim3(n_wg), dim3(64 * WAVES), lds, st, W, row_bytes, big_rows, bytes_per_wg, nt);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("%-40s piece %4d B  %d waves x %d instr/step, %d steps in flight (%3d KB), %3d WGs, nt=%d: %7.1f us  %6.2f TB/s  (%.1f GB/s per CU)\n", name, PIECE, WAVES, PER, DEPTH,
         DEPTH * STAGE / 1024, n_wg, nt, best * 1e3, (double)bytes_per_wg * n_wg / best / 1e9, (double)bytes_per_wg / best / 1e6);
}

int main() {
  const size_t total = (size_t)1 << 30;       // 1 GiB: well beyond the Infinity Cache
  char* W; CK(hipMalloc(&W, total)); CK(hipMemset(W, 1, total));
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int n_wg : {256, 224}) {
    for (int nt : {1, 0}) {
      run<128, 8, 2, 4>("chain today: 16 KB steps, 4 in flight", W, total, n_wg, nt, st);
      run<128, 8, 2, 6>("16 KB steps, 6 in flight", W, total, n_wg, nt, st);
      run<128, 4, 4, 6>("4 loader waves, 16 KB steps, 6 in flight", W, total, n_wg, nt, st);
      run<256, 8, 2, 6>("256-B pieces", W, total, n_wg, nt, st);
      run<512, 8, 2, 6>("512-B pieces", W, total, n_wg, nt, st);
      run<1024, 8, 2, 6>("1 KB pieces (whole instruction contiguous)", W, total, n_wg, nt, st);
      run<128, 8, 4, 4>("32 KB steps, 4 in flight (128 KB)", W, total, n_wg, nt, st);
      run<1024, 8, 4, 4>("1 KB pieces, 32 KB steps, 4 in flight", W, total, n_wg, nt, st);
    }
  }
  return 0;
}
