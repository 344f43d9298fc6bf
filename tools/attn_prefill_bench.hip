// A/B of the d = 128 prefill attention: the kernel in csrc/lia_attention.hip (third generation) against the second generation kept in
// tools/attn_prefill_gen2.inc ("candidate" below), same inputs, outputs compared bit for bit, both timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -I isca-2025-lia_amd/csrc -I include tools/attn_prefill_bench.hip -o tools/attn_prefill_bench   (-fno-honor-nans: as csrc/Makefile builds lia_attention.o)
//   tools/attn_prefill_bench            (OPT-30B's B 64 x T 256 x 56 heads, Llama-3-8B's B 32 x T 1024 x 32 / 8 heads, ragged T)
#include "../isca-2025-lia_amd/csrc/lia_attention.hip"
#include <vector>
#include <cstdio>
#include "attn_prefill_gen2.inc"
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint32_t rng_state = 12345;
static inline uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state; }
static inline bf16_t rnd_bf16(float scale) {
  // ~ uniform in [-scale, scale), rounded to bf16 by truncation of a float with random low bits (fine for a parity input)
  float f = ((int)(rnd() >> 8) - (1 << 23)) * (scale / (1 << 23));
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (bf16_t)(u >> 16);
}

static void run_case(const char* name, int B, int T, int heads, int kv_heads, int post_scale, int reps) {
  const int d = 128, H = heads * d, hd = kv_heads * d, Bc = B;
  const size_t nq = (size_t)B * T * H, nkv = (size_t)T * Bc * hd;
  std::vector<bf16_t> hq(nq), hk(nkv), hv(nkv);
  for (auto& x : hq) x = rnd_bf16(2.0f);
  for (auto& x : hk) x = rnd_bf16(2.0f);
  for (auto& x : hv) x = rnd_bf16(1.0f);
  bf16_t *q, *k, *v, *o0, *o1;
  CK(hipMalloc(&q, nq * 2)); CK(hipMalloc(&k, nkv * 2)); CK(hipMalloc(&v, nkv * 2)); CK(hipMalloc(&o0, nq * 2)); CK(hipMalloc(&o1, nq * 2));
  CK(hipMemcpy(q, hq.data(), nq * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(k, hk.data(), nkv * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(v, hv.data(), nkv * 2, hipMemcpyHostToDevice));
  CK(hipMemset(o0, 0xff, nq * 2)); CK(hipMemset(o1, 0xee, nq * 2));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms[2] = {0, 0};
  for (int which = 0; which < 2; ++which) {
    bf16_t* o = which ? o1 : o0;
    for (int it = 0; it < reps + 2; ++it) {
      if (it == 2) CK(hipEventRecord(e0, st));
      int rc = which ? attn_candidate_launch(q, H, k, v, o, H, B, T, heads, kv_heads, Bc, 0, post_scale, st)
                     : lia_attn_prefill_launch(q, H, k, v, o, H, B, T, heads, kv_heads, d, Bc, 0, post_scale, st);
      if (rc) { printf("launch rc %d\n", rc); exit(1); }
    }
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms[which], e0, e1));
    ms[which] /= reps;
  }
  std::vector<bf16_t> r0(nq), r1(nq);
  CK(hipMemcpy(r0.data(), o0, nq * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), o1, nq * 2, hipMemcpyDeviceToHost));
  size_t diff = 0, first = (size_t)-1;
  for (size_t i = 0; i < nq; ++i) if (r0[i] != r1[i]) { if (!diff) first = i; ++diff; }
  printf("%-28s B %3d T %4d heads %2d/%2d post_scale %d: product %8.1f us   candidate %8.1f us   ratio %.2f   differing outputs %zu of %zu%s\n",
         name, B, T, heads, kv_heads, post_scale, 1e3 * ms[0], 1e3 * ms[1], ms[0] / ms[1], diff, nq, diff ? "   <-- MISMATCH" : "");
  if (diff) printf("   first mismatch at %zu (b %zu t %zu col %zu): %04x vs %04x\n", first, first / ((size_t)T * H), (first / H) % T, first % H, r0[first], r1[first]);
  CK(hipFree(q)); CK(hipFree(k)); CK(hipFree(v)); CK(hipFree(o0)); CK(hipFree(o1));
  CK(hipStreamDestroy(st)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  if (getenv("BIG")) { run_case("Llama-3-8B prefill", 128, 1024, 32, 8, 1, reps); return 0; }
  run_case("OPT-30B prefill", 64, 256, 56, 56, 0, reps);
  run_case("Llama-3-8B prefill (B/4)", 32, 1024, 32, 8, 1, reps);
  run_case("OPT-30B, 2 minibatches", 32, 256, 56, 56, 0, reps);
  run_case("OPT-30B T 2016 (mb 8 of B 64)", 8, 2016, 56, 56, 0, reps > 5 ? 5 : reps);        // llm/scripts/lia_offline.sh:15: --input-tokens 2016 --num-minibatch 8
  run_case("OPT-30B T 1792 B 1", 1, 1792, 56, 56, 0, reps > 5 ? 5 : reps);
  if (getenv("QUICK")) return 0;
  const int ts[] = {1, 2, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 191, 192, 193, 255, 257, 300, 511, 513, 700, 1023, 2047, 2048};
  for (int t : ts) { run_case("ragged", 3, t, 5, 5, 0, 2); run_case("ragged gqa", 2, t, 8, 2, 1, 2); }
  return 0;
}
