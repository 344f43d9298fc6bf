// tools/pk_rate.hip: does a packed fp32 instruction (v_pk_fma_f32: two fmas per lane) cost one VALU slot or two on gfx950?
// hipcc (clang 22) unpacks v_pk_{fma,mul,add}_f32 that sit in the shadow of an MFMA into two single instructions (SIPreEmitPeephole);
// in a loop bound by VALU issue with two waves per SIMD that is only free if a packed instruction takes twice a single one's time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pk_rate.hip -o tools/pk_rate && tools/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x2 x[8];
  for (int i = 0; i < 8; ++i) x[i] = f32x2{(float)threadIdx.x + i, (float)i};
  f32x2 va = {a, a}, vb = {b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(va), "v"(vb));
        if (MODE == 1) {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].x) : "v"(a), "v"(b));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].y) : "v"(a), "v"(b));
        }
        if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(va));
        if (MODE == 3) {
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i].x) : "v"(a));
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i].y) : "v"(a));
        }
        if (MODE == 4) {
          asm volatile("v_exp_f32 %0, %0" : "+v"(x[i].x));
          asm volatile("v_exp_f32 %0, %0" : "+v"(x[i].y));
        }
        if (MODE == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[i].x) : "v"(x[i].x), "v"(x[i].y));
      }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int wgs_per_cu, float* out) {
  const int iters = 20000, grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 100, 1.0f, 0.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per wave: iters * 32 "pair operations"; waves per SIMD = wgs_per_cu (4 waves of a workgroup on 4 SIMDs)
  const double pair_ops_per_simd = (double)iters * 32 * wgs_per_cu;
  printf("%-34s %d waves/SIMD: %8.3f ms  -> %.2f ns per pair operation per SIMD (a 4-cycle slot at 2.4 GHz = 1.67 ns)\n", name, wgs_per_cu, ms,
         ms * 1e6 / pair_ops_per_simd);
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  for (int w = 1; w <= 2; ++w) {
    run<0>("v_pk_fma_f32 (1 instr / pair)", w, out);
    run<1>("2 x v_fma_f32", w, out);
    run<2>("v_pk_mul_f32 (1 instr / pair)", w, out);
    run<3>("2 x v_mul_f32", w, out);
    run<4>("2 x v_exp_f32", w, out);
    run<5>("v_cvt_pk_bf16_f32 (1 instr / pair)", w, out);
  }
  return 0;
}
