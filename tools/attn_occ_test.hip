// occupancy sensitivity of the product prefill attention: launch with extra dynamic LDS so that only one workgroup fits per CU
#include "../isca-2025-lia_amd/csrc/lia_attention.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
  const int B = 64, T = 256, heads = 56, d = 128, H = heads * d;
  const size_t nq = (size_t)B * T * H;
  bf16_t *q, *k, *v, *o;
  CK(hipMalloc(&q, nq * 2)); CK(hipMalloc(&k, nq * 2)); CK(hipMalloc(&v, nq * 2)); CK(hipMalloc(&o, nq * 2));
  CK(hipMemset(q, 0x3c, nq * 2)); CK(hipMemset(k, 0x3c, nq * 2)); CK(hipMemset(v, 0x3c, nq * 2));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)lia_attn_prefill128_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  for (int extra : {0, 32 * 1024}) {       // 64 KB static + 32 KB dynamic = 96 KB: one workgroup per CU
    const int nqb = (T + 127) / 128, n_groups = B * heads;
    dim3 grid((unsigned)(((n_groups + 7) / 8) * 8 * nqb));          // r06: the XCD-aware 1-D grid of lia_attn_prefill_launch
    const long hd = (long)heads * 128;
    for (int it = 0; it < 22; ++it) {
      if (it == 2) CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(lia_attn_prefill128_kernel<0>, grid, dim3(256), extra, st, q, (long)H, k, v, o, (long)H, T, heads, heads, (long)B * hd, hd, 0, 0.0883883f, nqb, n_groups);
    }
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("extra dynamic LDS %d KB: %.1f us per launch\n", extra / 1024, 1e3 * ms / 20);
  }
  return 0;
}
