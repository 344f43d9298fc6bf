// Shared device/host helpers for the LIA hot-path kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // bf16 bit pattern; all tensors cross the C ABI as these

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LIA_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((uint32_t)v << 16); }

// One bf16 rounding point (round-to-nearest-even; hipcc lowers the cast to v_cvt_pk_bf16_f32,
// which keeps NaN a NaN -- MI355X_MICROARCH.md "Correctness boundaries").
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// The reference materialises a bf16 tensor after the matmul, after "+ bias", and after
// "residual + ." (decoder.py:79-105,229,310; attentions.py:393-394,418).  The fused epilogues keep
// every one of those roundings.
struct LiaEpilogue {
  const bf16_t* bias;      // [N] or nullptr
  const bf16_t* residual;  // [M, ldr] or nullptr
  long ldr;
  int relu;
};

__device__ __forceinline__ float lia_epilogue_apply(float acc, float bias, bool has_bias, int relu, float res,
                                                    bool has_res) {
  float t = rbf(acc);
  if (has_bias) t = rbf(t + bias);
  if (relu) t = fmaxf(t, 0.f);
  if (has_res) t = rbf(res + t);
  return t;
}

// Where a GEMM output row lands.  Up to three equal-width column segments (fused q|k|v projection),
// each with its own base / leading dimension; a segment in "cache" mode scatters token row
// m = b*T + t to the seq-major KV-cache row (pos0 + t)*Bc + b0 + b  (attentions.py:457-458,475-476,
// 490-491: key.permute(1,0,2,3) written into the [S,B,h,d] cache).
struct LiaOutMap {
  bf16_t* base[3];
  long ld[3];
  int cache_mode[3];
  int seg_n;   // columns per segment (N if a single segment)
  int T;       // tokens per batch row in this call
  int Bc;      // batch size of the cache (row pitch in batch rows)
  int b0;      // first batch row of this minibatch inside the cache
  int pos0;    // first sequence position written
};

__device__ __forceinline__ bf16_t* lia_out_ptr(const LiaOutMap& o, int m, int n) {
  int s = n / o.seg_n;
  int nn = n - s * o.seg_n;
  long row = m;
  if (o.cache_mode[s]) {
    int b = m / o.T, t = m - b * o.T;
    row = (long)(o.pos0 + t) * o.Bc + o.b0 + b;
  }
  return o.base[s] + row * o.ld[s] + nn;
}
