// tools/issue_model.hip: how one SIMD of gfx950 shares its time between the matrix pipe, the vector ALU, scalar instructions and
// s_nop when one or two waves run on it -- the model behind the prefill attention's per-tile time (LABNOTES r06).
// Every wave runs `iters` iterations of: NM x v_mfma_f32_32x32x16_bf16 (two independent accumulators), NV x v_fma_f32 (8 independent
// chains), NS x s_add_u32 / NN x s_nop 0, in the interleaved order a software-pipelined loop would have.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value tools/issue_model.hip -o tools/issue_model && tools/issue_model
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NM, int NV, int NS, int NN, int OP = 0> __global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 acc0 = {0}, acc1 = {0};
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)(threadIdx.x + i); fb[i] = (__bf16)(float)i; }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = (float)threadIdx.x + i;
  unsigned s = 0;
  constexpr int G = NM > 0 ? NM : 1;          // groups per iteration: one MFMA (if any) then its share of the other instructions
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (NM > 0) {
        if (g & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(fa), "v"(fb));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
      }
#pragma unroll
      for (int i = 0; i < NV / G; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(a), "v"(b));
        if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i & 7]));
        if (OP == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i & 7]) : "v"(a));
      }
      // (one asm block: between separate blocks that write an SGPR hipcc puts an s_nop of its own)
      if (NS / G > 0) asm volatile(".rept %1\n s_add_u32 %0, %0, 1\n .endr" : "+s"(s) : "i"(NS / G) : "scc");
#pragma unroll
      for (int i = 0; i < NN / G; ++i) asm volatile("s_nop 0");
    }
  }
  float r = (float)s;
  for (int i = 0; i < 8; ++i) r += x[i];
  for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int NM, int NV, int NS, int NN, int OP = 0> void run(float* out) {
  const int iters = 4000;
  for (int w = 1; w <= 2; ++w) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, NV, NS, NN, OP>), dim3(256 * w), dim3(256), 0, 0, out, 50, 1.0f, 0.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV, NS, NN, OP>), dim3(256 * w), dim3(256), 0, 0, out, iters, 1.0f, 0.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s mfma %2d  valu %3d  salu %3d  s_nop %3d | %d wave(s)/SIMD: %7.1f ns per iteration of one wave's loop%s\n", OP == 0 ? "v_fma_f32" : OP == 1 ? "v_exp_f32" : "v_and_b32", NM, NV, NS, NN, w, ms * 1e6 / iters,
           w == 2 ? "  (both waves finish in this time)" : "");
  }
}
int main() {
  float* out; hipMalloc(&out, 256 * 2 * 256 * 4);
  run<8, 0, 0, 0>(out);      // the matrix pipe alone: 8 x 32 x 32 x 16
  run<0, 128, 0, 0>(out);    // the vector ALU alone
  run<0, 0, 128, 0>(out);    // scalar instructions alone
  run<0, 0, 0, 128>(out);    // s_nop 0 alone
  run<8, 32, 0, 0>(out);     // 4 vector instructions per MFMA
  run<8, 64, 0, 0>(out);     // 8
  run<8, 128, 0, 0>(out);    // 16: about sweep 1 of the attention (16 MFMAs, 203-233 vector instructions)
  run<8, 128, 32, 32>(out);  // + scalar + s_nop: about the whole instruction stream of a tile
  run<8, 64, 32, 32>(out);
  run<0, 128, 32, 32>(out);
  run<8, 16, 0, 0>(out);
  run<8, 48, 0, 0>(out);
  run<0, 64, 0, 0, 1>(out);  // v_exp_f32 alone, under MFMAs
  run<8, 32, 0, 0, 1>(out);
  run<8, 64, 0, 0, 1>(out);
  run<0, 128, 0, 0, 2>(out); // an integer instruction alone, under MFMAs
  run<8, 64, 0, 0, 2>(out);
  run<8, 128, 0, 0, 2>(out);
  return 0;
}
