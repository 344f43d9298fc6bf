// Stand-alone timing of the GEMM kernels on cold weights (rotating buffers > Infinity Cache).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I isca-2025-lia_amd/csrc tools/gemm_bench.hip -o tools/gemm_bench
#include "../isca-2025-lia_amd/csrc/lia_gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(uint16_t* p, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t h = (uint32_t)(i * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    float f = ((int)(h & 0xffff) - 32768) * (0.02f / 32768.f);
    p[i] = (uint16_t)(__float_as_uint(f) >> 16);
  }
}

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 64;
  int force_split = argc > 2 ? atoi(argv[2]) : 0;

  struct Shape { const char* name; int N, K; };
  std::vector<Shape> shapes = {{"qkv", 21504, 7168}, {"out", 7168, 7168}, {"fc1", 28672, 7168}, {"fc2", 7168, 28672}, {"lm_head", 50272, 7168}};
  if (getenv("LLAMA")) shapes = {{"qkv", 6144, 4096}, {"o", 4096, 4096}, {"gate_up", 28672, 4096}, {"down", 4096, 14336}, {"lm_head", 128256, 4096}};
  if (getenv("SHAPE")) { int n_, k_; sscanf(getenv("SHAPE"), "%d,%d", &n_, &k_); shapes = {{"custom", n_, k_}, {"dummy", 16, 128}}; }
  if (getenv("KSWEEP")) shapes = {{"k1792", 7168, 1792}, {"k3584", 7168, 3584}, {"k7168", 7168, 7168}, {"k14336", 7168, 14336}, {"k28672", 7168, 28672}, {"dummy", 16, 128}};
  const int NBUF = 4;
  size_t maxw = (size_t)128256 * (4096 + 1024);
  uint16_t* w[NBUF];
  for (int i = 0; i < NBUF; ++i) { CK(hipMalloc(&w[i], maxw * 2)); fill_kernel<<<2048, 256>>>(w[i], maxw, 17 + i); }
  uint16_t *x, *y, *bias, *res; float* ws;
  CK(hipMalloc(&x, (size_t)M * (28672 + 1024) * 2)); fill_kernel<<<2048, 256>>>(x, (size_t)M * (28672 + 1024), 3);
  CK(hipMalloc(&y, (size_t)M * 128256 * 2)); CK(hipMalloc(&bias, 128256 * 2)); CK(hipMalloc(&res, (size_t)M * 128256 * 2));
  fill_kernel<<<256, 256>>>(bias, 128256, 5); fill_kernel<<<2048, 256>>>(res, (size_t)M * 128256, 7);
  if (M > 256) shapes.pop_back();  // no lm_head in prefill (last position only)
  size_t ws_bytes = M <= 256 ? (size_t)8 * M * 128256 * 4 : (size_t)4 * (M < 2048 ? M : 1) * 28672 * 4; CK(hipMalloc(&ws, ws_bytes));
  if (getenv("ZERO")) { for (int i = 0; i < NBUF; ++i) CK(hipMemset(w[i], 0, maxw * 2)); CK(hipMemset(x, 0, (size_t)M * 28672 * 2)); }
  LiaGemmOpts* const tickets = nullptr;       // (default options; the r04 in-launch combine and its tickets are gone)
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipDeviceSynchronize());
#ifdef LIA_GEMM_STAMPS
  if (getenv("S2STAMPS")) {
    // skinny kernel: where a workgroup's time goes (100 MHz ticks): start -> first chunk landed -> K loop done -> stores issued -> drained
    void* sp; CK(hipGetSymbolAddress(&sp, HIP_SYMBOL(g_s2_stamps)));
    for (auto& sh : shapes) {
      if (!strcmp(sh.name, "dummy")) continue;
      const bool glu = getenv("GLU") && !strcmp(sh.name, "gate_up");   // the Llama layer's own call: no bias / residual, SiLU*up paired by the epilogue
      LiaEpilogue ep{glu ? nullptr : bias, glu ? nullptr : res, sh.N, 0};
      LiaOutMap om; memset(&om, 0, sizeof(om)); om.base[0] = y; om.ld[0] = sh.N; om.seg_n = sh.N; om.T = 1;
      LiaPost post{}; post.kind = LIA_POST_SILU_MUL; post.out = res; post.ldo = sh.N / 2; post.gu_block = LIA_GU_BLOCK;
      int post_done = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(sp, 0, 8192 * 8 * 8, st));
        lia_gemm_launch(x, sh.K, w[rep % NBUF], sh.K, M, sh.N, sh.K, &ep, &om, ws, ws_bytes, tickets, force_split, st, e0, e1, nullptr, glu ? &post : nullptr, glu ? &post_done : nullptr);
      }
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> h(8192 * 8);
      CK(hipMemcpy(h.data(), sp, h.size() * 8, hipMemcpyDeviceToHost));
      int nwg = 0; unsigned long long first = ~0ull, last = 0, last_start = 0, first_end = ~0ull;
      double d01 = 0, d12 = 0, d23 = 0, d34 = 0;
      for (int g = 0; g < 8192; ++g) {
        const unsigned long long* q = h.data() + 8 * g;
        if (!q[0]) continue;
        ++nwg; first = std::min(first, q[0]); last = std::max(last, q[4]); last_start = std::max(last_start, q[0]); first_end = std::min(first_end, q[4]);
        d01 += (double)(q[1] - q[0]); d12 += (double)(q[2] - q[1]); d23 += (double)(q[3] - q[2]); d34 += (double)(q[4] - q[3]);
      }
      if (getenv("S2DUMP")) {
        // K-loop time and end time per workgroup, grouped by blockIdx.x % 8 (the XCD the dispatcher deals it to) and by octiles
        double sum[8] = {0}, mx[8] = {0}; int cnt[8] = {0};
        std::vector<double> loops, ends;
        for (int g = 0; g < 8192; ++g) {
          const unsigned long long* q = h.data() + 8 * g;
          if (!q[0]) continue;
          const double l = (double)(q[2] - q[1]) * 0.01, e = (double)(q[4] - first) * 0.01;
          sum[g % 8] += l; mx[g % 8] = std::max(mx[g % 8], e); ++cnt[g % 8];
          loops.push_back(l); ends.push_back(e);
        }
        printf("   per g %% 8: K loop avg / last end:");
        for (int i = 0; i < 8; ++i) printf("  %d: %.1f / %.1f", i, cnt[i] ? sum[i] / cnt[i] : 0.0, mx[i]);
        std::sort(loops.begin(), loops.end()); std::sort(ends.begin(), ends.end());
        printf("\n   K loop us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f;  end us: p10 %.1f p50 %.1f p90 %.1f max %.1f\n", loops.front(), loops[loops.size() / 10],
               loops[loops.size() / 2], loops[loops.size() * 9 / 10], loops.back(), ends[ends.size() / 10], ends[ends.size() / 2], ends[ends.size() * 9 / 10], ends.back());
        // first 32 workgroups in launch order: start offset, first chunk, loop
        printf("   g: start / first chunk / loop / end:");
        for (int g = 0; g < 8192 && g < 24; ++g) { const unsigned long long* q = h.data() + 8 * g; if (q[0]) printf("  %d: %.1f/%.1f/%.1f/%.1f", g, (q[0] - first) * 0.01, (q[1] - q[0]) * 0.01, (q[2] - q[1]) * 0.01, (q[4] - first) * 0.01); }
        printf("\n");
      }
      printf("%-8s M=%d N=%d K=%d: event bracket %.1f us; %d workgroups, span %.2f us (starts spread over %.2f us, first end at %.2f us); per workgroup: "
             "first chunk %.2f us, K loop %.2f us, stores issued %.2f us, drain %.2f us\n", sh.name, M, sh.N, sh.K, ms * 1e3, nwg, (last - first) * 0.01,
             (last_start - first) * 0.01, (first_end - first) * 0.01, d01 * 0.01 / nwg, d12 * 0.01 / nwg, d23 * 0.01 / nwg, d34 * 0.01 / nwg);
    }
    return 0;
  }
#endif
  for (auto& s : shapes) {
    const bool glu = getenv("GLU") && !strcmp(s.name, "gate_up");
    LiaEpilogue ep{(getenv("NOBIAS") || glu) ? nullptr : bias, (getenv("NORES") || glu) ? nullptr : res, s.N, 0};
    LiaOutMap om; memset(&om, 0, sizeof(om)); om.base[0] = y; om.ld[0] = s.N; om.seg_n = s.N; om.T = 1;
    LiaPost post_{}; post_.kind = LIA_POST_SILU_MUL; post_.out = res; post_.ldo = s.N / 2; post_.gu_block = LIA_GU_BLOCK;
    int post_done_ = 0;
    const LiaPost* post = glu ? &post_ : nullptr;
    int* post_done = glu ? &post_done_ : nullptr;
    const int iters = M > 256 ? 4 : 12;
    const int nbuf = getenv("WARM") ? 1 : NBUF;   // WARM=1: the same weight buffer every launch (it stays in the 256 MB Infinity Cache when it fits) -- what an L2 / MALL prefetch of the next GEMM's weights could buy at best
    const long ldp = getenv("LDPAD") ? atol(getenv("LDPAD")) : 0;   // row stride = K + LDPAD elements (aliasing experiment)
    for (int it = 0; it < 3; ++it) lia_gemm_launch(x, s.K + ldp, w[it % nbuf], s.K + ldp, M, s.N, s.K, &ep, &om, ws, ws_bytes, tickets, force_split, st, nullptr, nullptr, nullptr, post, post_done);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int it = 0; it < iters; ++it) lia_gemm_launch(x, s.K + ldp, w[it % nbuf], s.K + ldp, M, s.N, s.K, &ep, &om, ws, ws_bytes, tickets, force_split, st, nullptr, nullptr, nullptr, post, post_done);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    double bytes = 2.0 * ((double)s.N * s.K + (double)M * s.K + (double)M * s.N);
    double flops = 2.0 * M * (double)s.N * s.K;
    printf("%-8s M=%d N=%d K=%d: %8.1f us  %7.1f GB/s  %7.1f TFLOP/s\n", s.name, M, s.N, s.K, ms * 1e3, bytes / ms / 1e6, flops / ms / 1e9);
  }
  return 0;
}
