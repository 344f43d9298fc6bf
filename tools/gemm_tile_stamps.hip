// Where a 256 x 256 tile of the prefill GEMM spends its time, and how long a CU waits between two tiles (s_memrealtime stamps per
// workgroup, HW_ID / XCC_ID to find the workgroups that shared a CU).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLIA_GEMM_STAMPS -I isca-2025-lia_amd/csrc tools/gemm_tile_stamps.hip -o tools/gemm_tile_stamps
//   tools/gemm_tile_stamps [M N K residual]
#include "../isca-2025-lia_amd/csrc/lia_gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void fill_kernel(uint16_t* p, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) { uint32_t h = (uint32_t)(i * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    float f = ((int)(h & 0xffff) - 32768) * (0.02f / 32768.f); p[i] = (uint16_t)(__float_as_uint(f) >> 16); }
}
static void run(int M, int N, int K, int with_res) {
  uint16_t *x, *w, *y, *bias, *res;
  CK(hipMalloc(&x, (size_t)M * K * 2)); CK(hipMalloc(&w, (size_t)N * K * 2)); CK(hipMalloc(&y, (size_t)M * N * 2)); CK(hipMalloc(&bias, N * 2)); CK(hipMalloc(&res, (size_t)M * N * 2));
  fill_kernel<<<2048, 256>>>(x, (size_t)M * K, 3); fill_kernel<<<2048, 256>>>(w, (size_t)N * K, 5); fill_kernel<<<64, 256>>>(bias, N, 7); fill_kernel<<<2048, 256>>>(res, (size_t)M * N, 9);
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  LiaEpilogue ep{bias, with_res ? res : nullptr, N, 0};
  LiaOutMap om; memset(&om, 0, sizeof(om)); om.base[0] = y; om.ld[0] = N; om.seg_n = N; om.T = 1;
  void* sp; CK(hipGetSymbolAddress(&sp, HIP_SYMBOL(g_t4_stamps)));
  float ms = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemsetAsync(sp, 0, sizeof(unsigned long long) * 65536 * 8, st));
    int regime = 0;
    if (lia_gemm_launch(x, K, w, K, M, N, K, &ep, &om, nullptr, 0, nullptr, 0, st, e0, e1, &regime, nullptr, nullptr)) { printf("launch failed\n"); exit(1); }
    CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const int ntiles = ((M + 255) / 256) * ((N + 255) / 256);
  std::vector<unsigned long long> h((size_t)65536 * 8);
  CK(hipMemcpy(h.data(), sp, h.size() * 8, hipMemcpyDeviceToHost));
  double d01 = 0, d12 = 0, d23 = 0, d34 = 0; int n = 0;
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> by_cu;
  for (int g = 0; g < std::min(ntiles, 65536); ++g) {
    const unsigned long long* q = &h[(size_t)g * 8];
    if (!q[0]) continue;
    ++n; d01 += q[1] - q[0]; d12 += q[2] - q[1]; d23 += q[3] - q[2]; d34 += q[4] - q[3];
    const unsigned hw = (unsigned)q[5], xcc = (unsigned)(q[5] >> 32) & 0xf;
    const unsigned long long cu = ((unsigned long long)xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
    by_cu[cu].push_back({q[0], q[4]});
  }
  double gap = 0; int ngap = 0; double gap_max = 0;
  for (auto& kv : by_cu) {
    auto& v = kv.second; std::sort(v.begin(), v.end());
    for (size_t i = 1; i < v.size(); ++i) { const double g = (double)v[i].first - (double)v[i - 1].second; gap += g; ++ngap; gap_max = std::max(gap_max, g); }
  }
  const double u = 0.01;   // 100 MHz ticks -> us
  printf("M %d N %d K %d residual %d: kernel %.1f us, %d tiles on %zu CUs (%.1f per CU): per tile  prologue (first K-tile lands) %.2f us | K loop %.2f us (%.3f us per 64-deep step) | "
         "epilogue issue %.2f us | store drain %.2f us | gap to the next tile's start on the same CU %.2f us (max %.1f)\n",
         M, N, K, with_res, 1e3 * ms, n, by_cu.size(), (double)n / by_cu.size(), u * d01 / n, u * d12 / n, u * d12 / n / (K / 64), u * d23 / n, u * d34 / n, ngap ? u * gap / ngap : 0.0, u * gap_max);
  CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(bias)); CK(hipFree(res));
}
int main(int argc, char** argv) {
  if (argc >= 4) { run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), argc > 4 ? atoi(argv[4]) : 0); return 0; }
  run(16384, 21504, 7168, 0);      // OPT-30B q|k|v
  run(16384, 7168, 7168, 1);       // out
  run(16384, 28672, 7168, 0);      // fc1
  run(16384, 7168, 28672, 1);      // fc2
  run(32768, 4096, 4096, 1);       // Llama-3-8B o (a quarter of the rows)
  run(32768, 4096, 14336, 1);      // down
  return 0;
}
