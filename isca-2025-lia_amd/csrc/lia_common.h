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
  int glu;                 // tiled kernels only: the N columns are blocks of 32 gate | 32 up columns (LIA_GU_BLOCK); the epilogue
                           // writes silu(gate) * up -- N/2 columns -- to the output map instead of the N raw ones
};
#define LIA_GU_BLOCK 32

__device__ __forceinline__ float lia_epilogue_apply(float acc, float bias, bool has_bias, int relu, float res,
                                                    bool has_res) {
  float t = rbf(acc);
  if (has_bias) t = rbf(t + bias);
  if (relu) t = fmaxf(t, 0.f);
  if (has_res) t = rbf(res + t);
  return t;
}

// Where a GEMM output row lands.  Up to LIA_OUT_SEGS equal-width column segments (fused q|k|v projection),
// each with its own base / leading dimension; a segment in "cache" mode scatters token row
// m = b*T + t to the seq-major KV-cache row (pos0 + t)*Bc + b0 + b  (attentions.py:457-458,475-476,
// 490-491: key.permute(1,0,2,3) written into the [S,B,h,d] cache).
#define LIA_OUT_SEGS 10
struct LiaOutMap {
  bf16_t* base[LIA_OUT_SEGS];    // (a grouped-query q|k|v projection: heads/kv_heads segments of q, then k, then v)
  long ld[LIA_OUT_SEGS];
  int cache_mode[LIA_OUT_SEGS];
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

// ---------------------------------------------------------------------------------------------
// Work that follows a decode GEMM and can ride in its split-K combine instead of a kernel of its own
// (lia_gemm_launch's `post`; the caller runs the stand-alone kernel when the GEMM was not split).  The arithmetic of
// each kind is the device function the stand-alone kernel uses too, so the two routes give the same bits.
// ---------------------------------------------------------------------------------------------
enum { LIA_POST_NONE = 0, LIA_POST_LAYERNORM = 1, LIA_POST_RMSNORM = 2, LIA_POST_SILU_MUL = 3, LIA_POST_ROPE = 4 };
struct LiaPost {
  int kind;
  // LAYERNORM / RMSNORM of the finished output row (N = hidden): out[m][:] = norm(y[m][:]; g, b, eps)
  const bf16_t* g;
  const bf16_t* b;
  float eps;
  bf16_t* out;          // norm: [M][ldo];  SILU_MUL: act[M][ldo] = silu(gate) * up  (y itself is not written)
  long ldo;
  int gu_block;         // SILU_MUL: 0 = columns [gate (N/2) | up (N/2)]; LIA_GU_BLOCK = blocks of 32 gate | 32 up columns
  // ROPE: the first rot_heads heads (width hd) of every output row are rotated at position pos0 + m % T, the rest are plain
  const bf16_t* cos_t;
  const bf16_t* sin_t;
  int rot_heads, hd, pos0, T;
};

// Per-context switches and counters of the GEMM launcher (lia_gemm_launch's `opts`; owned by lia_ctx, set through
// lia_ctx_set_option / read through lia_ctx_get_counter -- the library keeps no process-wide setting, two contexts may differ).
struct LiaGemmOpts {
  int fuse_combine;         // 1 (default): the split-K combine also runs the op behind the GEMM; 0: every op a kernel of its own (A/B tests)
  long fused_combines[5];   // fused combines launched per LIA_POST_* kind (tests assert the route was taken)
};

// One output row per workgroup (16 waves' worth of threads: the row ops of a decode step are pure latency, and a combine that
// reads 8 slabs x 28 KB per row wants every load of the row in flight at once).  The row is cut into 8-value pieces (packed
// bf16); VIRTUAL thread v of LIA_ROW_THREADS holds pieces v, v + LIA_ROW_THREADS, ...  A workgroup of LIA_ROW_THREADS / VT real
// threads runs VT virtual threads per thread (real thread t = virtual threads t + h * LIA_ROW_THREADS / VT, h < VT): the
// stand-alone row kernels and the split-K combines use VT = 1 (a 512-thread workgroup would run VT = 2: every sum is taken over
// the same values in the same order, so the two give the same bits).
#define LIA_ROW_WAVES 16
#define LIA_ROW_THREADS (64 * LIA_ROW_WAVES)

// 16-byte store; SC1: write-through, visible to other workgroups of the SAME launch without a release fence once the storing
// wave has drained it (cdna_hip_programming.md Guideline 16, R1)
template <bool SC1> __device__ __forceinline__ void lia_store16(void* p, const uint4& v) {
  if constexpr (SC1) {
    const u32x4 d{v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
  } else {
    *(uint4*)p = v;
  }
}
template <bool SC1> __device__ __forceinline__ void lia_store8(void* p, const uint2& v) {
  if constexpr (SC1) {
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_;
    const u32x2_ d{v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
  } else {
    *(uint2*)p = v;
  }
}

// sum over the virtual threads of a row workgroup, the same value in every thread (red: LIA_ROW_WAVES floats of LDS, used once
// per call); the virtual waves' totals are added in wave order
template <int VT> __device__ __forceinline__ float block_sum_row_vt(const float (&v)[VT], float* red) {
  constexpr int RW = LIA_ROW_WAVES / VT;     // real waves
#pragma unroll
  for (int h = 0; h < VT; ++h) {
    const float s = wave_sum(v[h]);
    if ((threadIdx.x & 63) == 0) red[h * RW + (threadIdx.x >> 6)] = s;
  }
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int w = 1; w < LIA_ROW_WAVES; ++w) t += red[w];
  return t;
}
__device__ __forceinline__ float block_sum_row(float v, float* red) {
  const float a[1] = {v};
  return block_sum_row_vt<1>(a, red);
}

// LayerNorm as torch.nn.functional.layer_norm on bf16: statistics in fp32 (two passes over the registers), one rounding.
template <int NV, int VT, bool SC1>
__device__ __forceinline__ void row_layernorm_vt(const uint4 (&v)[VT][NV], const uint4 (&gv)[VT][NV], const uint4 (&bv)[VT][NV], int nv, int H,
                                                 float eps, bf16_t* __restrict__ yr, float* red /* 2 x LIA_ROW_WAVES floats */) {
  constexpr int RTH = LIA_ROW_THREADS / VT;
  const int tid = threadIdx.x;
  float s[VT];
#pragma unroll
  for (int h = 0; h < VT; ++h) {
    s[h] = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (tid + RTH * h + LIA_ROW_THREADS * k < nv) {
        const uint32_t w[4] = {v[h][k].x, v[h][k].y, v[h][k].z, v[h][k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) s[h] += bf2f(w[j] & 0xffff) + bf2f(w[j] >> 16);
      }
    }
  }
  const float mean = block_sum_row_vt<VT>(s, red) / (float)H;
  float q[VT];
#pragma unroll
  for (int h = 0; h < VT; ++h) {
    q[h] = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (tid + RTH * h + LIA_ROW_THREADS * k < nv) {
        const uint32_t w[4] = {v[h][k].x, v[h][k].y, v[h][k].z, v[h][k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a = bf2f(w[j] & 0xffff) - mean, c = bf2f(w[j] >> 16) - mean;
          q[h] += a * a + c * c;
        }
      }
    }
  }
  const float rstd = 1.0f / sqrtf(block_sum_row_vt<VT>(q, red + LIA_ROW_WAVES) / (float)H + eps);
#pragma unroll
  for (int h = 0; h < VT; ++h) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = tid + RTH * h + LIA_ROW_THREADS * k;
      if (i < nv) {
        const uint32_t w[4] = {v[h][k].x, v[h][k].y, v[h][k].z, v[h][k].w}, gw[4] = {gv[h][k].x, gv[h][k].y, gv[h][k].z, gv[h][k].w},
                       bw[4] = {bv[h][k].x, bv[h][k].y, bv[h][k].z, bv[h][k].w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float lo = (bf2f(w[j] & 0xffff) - mean) * rstd * bf2f(gw[j] & 0xffff) + bf2f(bw[j] & 0xffff);
          float hi = (bf2f(w[j] >> 16) - mean) * rstd * bf2f(gw[j] >> 16) + bf2f(bw[j] >> 16);
          o[j] = pack_bf16x2(lo, hi);
        }
        lia_store16<SC1>(yr + 8 * i, uint4{o[0], o[1], o[2], o[3]});
      }
    }
  }
}
template <int NV>
__device__ __forceinline__ void row_layernorm_block(const uint4 (&v)[NV], const uint4 (&gv)[NV], const uint4 (&bv)[NV], int nv, int H,
                                                 float eps, bf16_t* __restrict__ yr, float* red /* 2 x LIA_ROW_WAVES floats */) {
  row_layernorm_vt<NV, 1, false>(reinterpret_cast<const uint4(&)[1][NV]>(v), reinterpret_cast<const uint4(&)[1][NV]>(gv),
                                 reinterpret_cast<const uint4(&)[1][NV]>(bv), nv, H, eps, yr, red);
}

// LlamaRMSNorm.forward: y = bf16( w * bf16( x * rsqrt(mean(x^2) + eps) ) )
template <int NV, int VT, bool SC1>
__device__ __forceinline__ void row_rmsnorm_vt(const uint4 (&v)[VT][NV], const uint4 (&gv)[VT][NV], int nv, int H, float eps,
                                               bf16_t* __restrict__ yr, float* red /* LIA_ROW_WAVES floats */) {
  constexpr int RTH = LIA_ROW_THREADS / VT;
  const int tid = threadIdx.x;
  float ss[VT];
#pragma unroll
  for (int h = 0; h < VT; ++h) {
    ss[h] = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (tid + RTH * h + LIA_ROW_THREADS * k < nv) {
        const uint32_t u[4] = {v[h][k].x, v[h][k].y, v[h][k].z, v[h][k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { float a = bf2f(u[j] & 0xffff), c = bf2f(u[j] >> 16); ss[h] += a * a + c * c; }
      }
    }
  }
  const float rstd = 1.0f / sqrtf(block_sum_row_vt<VT>(ss, red) / (float)H + eps);
#pragma unroll
  for (int h = 0; h < VT; ++h) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = tid + RTH * h + LIA_ROW_THREADS * k;
      if (i < nv) {
        const uint32_t u[4] = {v[h][k].x, v[h][k].y, v[h][k].z, v[h][k].w}, gw[4] = {gv[h][k].x, gv[h][k].y, gv[h][k].z, gv[h][k].w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = pack_bf16x2(bf2f(gw[j] & 0xffff) * rbf(bf2f(u[j] & 0xffff) * rstd), bf2f(gw[j] >> 16) * rbf(bf2f(u[j] >> 16) * rstd));
        lia_store16<SC1>(yr + 8 * i, uint4{o[0], o[1], o[2], o[3]});
      }
    }
  }
}
template <int NV>
__device__ __forceinline__ void row_rmsnorm_block(const uint4 (&v)[NV], const uint4 (&gv)[NV], int nv, int H, float eps,
                                               bf16_t* __restrict__ yr, float* red /* LIA_ROW_WAVES floats */) {
  row_rmsnorm_vt<NV, 1, false>(reinterpret_cast<const uint4(&)[1][NV]>(v), reinterpret_cast<const uint4(&)[1][NV]>(gv), nv, H, eps, yr, red);
}

// LlamaMLP act_fn(gate) * up on two packed bf16 pairs: bf16( bf16(silu(g)) * u ); silu in fp32 through the hardware exp2 /
// reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each) instead of libm expf + an IEEE division -- the result is rounded to bf16 right
// after, so a last-bit fp32 difference reaches the output about once in 2^15 elements (HF's own silu is no libm-exact
// reference either: Sleef on the CPU, a fast-math kernel on GPUs)
__device__ __forceinline__ uint32_t lia_silu_mul_pair(uint32_t gw, uint32_t uw) {
  const float g0 = bf2f(gw & 0xffff), g1 = bf2f(gw >> 16);
  const float s0 = rbf(g0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g0)));
  const float s1 = rbf(g1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g1)));
  return pack_bf16x2(s0 * bf2f(uw & 0xffff), s1 * bf2f(uw >> 16));
}

// apply_rotary_pos_emb on one packed pair of (x[i], x[i+1]) and its partner (x[i+half], x[i+half+1]):
// out = bf16( bf16(x*cos) + bf16(rotate_half(x)*sin) ).  c0/s0: table entries at i, c1/s1: at i + half.
__device__ __forceinline__ void lia_rope_pair(uint32_t aw, uint32_t bw, uint32_t c0, uint32_t c1, uint32_t s0, uint32_t s1,
                                              uint32_t& oa, uint32_t& ob) {
  const float a_lo = bf2f(aw & 0xffff), a_hi = bf2f(aw >> 16), b_lo = bf2f(bw & 0xffff), b_hi = bf2f(bw >> 16);
  oa = pack_bf16x2(rbf(a_lo * bf2f(c0 & 0xffff)) + rbf(-b_lo * bf2f(s0 & 0xffff)), rbf(a_hi * bf2f(c0 >> 16)) + rbf(-b_hi * bf2f(s0 >> 16)));
  ob = pack_bf16x2(rbf(b_lo * bf2f(c1 & 0xffff)) + rbf(a_lo * bf2f(s1 & 0xffff)), rbf(b_hi * bf2f(c1 >> 16)) + rbf(a_hi * bf2f(s1 >> 16)));
}
