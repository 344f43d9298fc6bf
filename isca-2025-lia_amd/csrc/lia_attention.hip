// GPU attention for policies 0 / 3 (attentions.py:443-536), with every bf16 rounding point of the
// reference's op sequence kept:
//     q  = bf16(q * d^-0.5)            (:456)
//     s  = bf16(q . k)                 torch.bmm output (:499)
//     prefill: causal mask -> -inf     (:444-449, 500-509: fp32 add of -3.4028e38, clamp, cast to bf16)
//     p  = bf16(softmax(s))            fp32 inside (:512);  decode (T == 1): no mask
//     o  = bf16(p . v)                 torch.bmm output (:529)
// Because p is rounded AFTER normalisation by the full-row sum, a one-pass online softmax cannot
// reproduce it; the prefill kernel therefore makes two sweeps over the keys (statistics, then P.V).
// Attention is 0.3 % of the prefill flops at T = 256, so the second Q.K^T is
// irrelevant to the layer time while keeping bit-level agreement with the oracle.
//
// K and V are always read from the seq-major cache layout [S][Bc][h][d] (attentions.py:457-476) --
// the q|k|v GEMM epilogue has already scattered the fresh rows there -- so one kernel serves the
// resident layers (device cache) and the streamed prefill (staging slab that is then copied to the
// host cache).
#include <cstdlib>
#include "lia_common.h"
#include <type_traits>

// ---------------------------------------------------------------------------------------------
// prefill: LDS-tiled, MFMA 32x32x16.  Workgroup = 4 waves = 128 query rows of one (batch, head);
// wave w owns query rows q0 + 32w .. +31.  Key tiles of 32 are staged in LDS and shared.
//   S^T = K . Qs^T   (A = K rows from LDS, B = Qs rows held in registers): the lane owns one query
//   column, so the softmax row statistics are in-lane + one cross-half shuffle.
//   O^T = V^T . P^T  (A = V^T from a transposed LDS image, B = the P^T accumulator registers re-used
//   directly as the next MFMA's operand, cdna_hip_programming.md section 3).
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void lia_attn_prefill_kernel(const bf16_t* __restrict__ q, long ldq,
                                                                const bf16_t* __restrict__ kc,
                                                                const bf16_t* __restrict__ vc, bf16_t* __restrict__ out,
                                                                long ldo, int T, int heads, int kv_heads, int Bc, int b0,
                                                                float scaling, int post_scale) {
  constexpr int CH = D / 8;              // 16-byte chunks per K row
  constexpr int RPB = 16 / CH;           // K rows per 256-byte LDS bank row
  constexpr int KSTEPS = D / 16;         // MFMA k-steps over d for S^T
  constexpr int DB = D / 32;             // 32-row blocks of O^T
  constexpr int VT_STRIDE = 72;          // bytes per V^T row: 32 keys * 2 B + 8 pad (8-byte aligned)
  __shared__ __attribute__((aligned(16))) char k_lds[32 * D * 2];
  __shared__ __attribute__((aligned(16))) char vt_lds[D * VT_STRIDE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int hh = blockIdx.y, b = blockIdx.z;
  // every other (batch row, head) takes its query blocks and its waves' row blocks in reverse order (r05, see the d = 128 kernel:
  // otherwise XCD gridDim.x - 1 and SIMD 3 collect all the long rows of the causal triangle); the results do not depend on it
  const int rev = (hh + b) & 1;
  const int q_wg = (rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x) * 128;
  const int q_wave = q_wg + (rev ? 3 - wave : wave) * 32;
  // grouped-query attention (Llama family): query head hh reads K/V head hh / (heads / kv_heads); OPT: kv_heads == heads
  const int kh = hh / (heads / kv_heads);
  const long kv_row = (long)Bc * kv_heads * D;  // elements between consecutive sequence positions
  const bf16_t* kbase = kc + ((long)(b0 + b) * kv_heads + kh) * D;
  const bf16_t* vbase = vc + ((long)(b0 + b) * kv_heads + kh) * D;
  // OPT rounds q * d^-0.5 before the product (attentions.py:456); HF Llama rounds the product, then scales it
  // (eager_attention_forward: matmul(q, k^T) * scaling)
  const float qscale = post_scale ? 1.0f : scaling;

  // Qs fragments: B operand, lane holds Qs[q_wave + r][16 s + 8 h + j]
  bf16x8 qf[KSTEPS];
  {
    const int qrow = min(q_wave + r, T - 1);
    const bf16_t* qp = q + ((long)b * T + qrow) * ldq + (long)hh * D;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
#ifdef P3X_NO_QLOAD
      uint4 v = uint4{(unsigned)lane, (unsigned)s, 0x3c003c00u, 0x3c003c00u};
#else
      uint4 v = *(const uint4*)(qp + 16 * s + 8 * h);
#endif
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(bf2f(w[j] & 0xffff) * qscale, bf2f(w[j] >> 16) * qscale);
      qf[s] = __builtin_bit_cast(bf16x8, uint4{o[0], o[1], o[2], o[3]});
    }
  }

  const int n_tiles_wg = min((q_wg + 128 + 31) / 32, (T + 31) / 32);  // causal: keys <= last query of the WG
  const int my_q = q_wave + r;                                        // this lane's query (column of S^T)

  auto stage_k = [&](int kt) {
    // 32 keys x D: thread -> (key = tid / CH', chunk); 256 threads move 4 KB per round
    constexpr int ROUNDS = (32 * CH + 255) / 256;
#pragma unroll
    for (int rr = 0; rr < ROUNDS; ++rr) {
      int idx = rr * 256 + tid;
      if (idx < 32 * CH) {
        int key = idx / CH, c = idx % CH;
        int gk = min(kt * 32 + key, T - 1);
        uint4 v = *(const uint4*)(kbase + (long)gk * kv_row + 8 * c);
        *(uint4*)(k_lds + key * (D * 2) + ((c ^ ((key / RPB) % CH)) << 4)) = v;
      }
    }
  };
  auto stage_vt = [&](int kt) {
    constexpr int ROUNDS = (32 * CH + 255) / 256;
#pragma unroll
    for (int rr = 0; rr < ROUNDS; ++rr) {
      int idx = rr * 256 + tid;
      if (idx < 32 * CH) {
        int key = idx / CH, c = idx % CH;
        int gk = min(kt * 32 + key, T - 1);
        uint4 v = *(const uint4*)(vbase + (long)gk * kv_row + 8 * c);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          *(bf16_t*)(vt_lds + (8 * c + 2 * j) * VT_STRIDE + key * 2) = (bf16_t)(w[j] & 0xffff);
          *(bf16_t*)(vt_lds + (8 * c + 2 * j + 1) * VT_STRIDE + key * 2) = (bf16_t)(w[j] >> 16);
        }
      }
    }
  };
  // S^T tile for key tile kt: lane gets S^T[key = kt*32 + (i&3) + 8*(i>>2) + 4h][query = my_q], rounded
  // to bf16 and causally masked
  auto scores = [&](int kt, f32x16& sacc) {
    sacc = f32x16{0};
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      uint4 kv = *(const uint4*)(k_lds + r * (D * 2) + (((2 * s + h) ^ ((r / RPB) % CH)) << 4));
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kv), qf[s], sacc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int key = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      float sv = rbf(sacc[i]);
      if (post_scale) sv = rbf(sv * scaling);
      sacc[i] = (key <= my_q && key < T) ? sv : -INFINITY;
    }
  };

  // ---- sweep 1: row max and sum of exp over all keys <= query ----
  float m = -INFINITY, l = 0.f;
  for (int kt = 0; kt < n_tiles_wg; ++kt) {
    __syncthreads();
    stage_k(kt);
    __syncthreads();
    if (kt * 32 <= q_wave + 31) {  // wave-uniform: tile not entirely above the diagonal
      f32x16 s;
      scores(kt, s);
      float tm = s[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) tm = fmaxf(tm, s[i]);
      tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
      float mn = fmaxf(m, tm);
      if (mn > -INFINITY) {
        float ts = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) ts += __expf(s[i] - mn);
        ts += __shfl_xor(ts, 32, 64);
        l = l * __expf(m - mn) + ts;
        m = mn;
      }
    }
  }
  const float inv_l = 1.0f / l;

  // ---- sweep 2: P = bf16(exp(s - m) / l), O^T += V^T . P^T ----
  f32x16 oacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) oacc[d] = f32x16{0};
  for (int kt = 0; kt < n_tiles_wg; ++kt) {
    __syncthreads();
    stage_k(kt);
    stage_vt(kt);
    __syncthreads();
    if (kt * 32 <= q_wave + 31) {
      f32x16 s;
      scores(kt, s);
      uint32_t pk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        pk[i] = pack_bf16x2(__expf(s[2 * i] - m) / l, __expf(s[2 * i + 1] - m) / l);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 pf = __builtin_bit_cast(bf16x8, uint4{pk[4 * ks], pk[4 * ks + 1], pk[4 * ks + 2], pk[4 * ks + 3]});
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const char* vrow = vt_lds + (32 * d + r) * VT_STRIDE + (16 * ks + 4 * h) * 2;
          uint2 lo = *(const uint2*)(vrow), hi = *(const uint2*)(vrow + 16);
          bf16x8 vf = __builtin_bit_cast(bf16x8, uint4{lo.x, lo.y, hi.x, hi.y});
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[d], 0, 0, 0);
        }
      }
    }
  }
  (void)inv_l;

  if (my_q < T) {
    bf16_t* op = out + ((long)b * T + my_q) * ldo + (long)hh * D;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 o;
        o.x = pack_bf16x2(oacc[d][4 * g], oacc[d][4 * g + 1]);
        o.y = pack_bf16x2(oacc[d][4 * g + 2], oacc[d][4 * g + 3]);
        *(uint2*)(op + 32 * d + 8 * g + 4 * h) = o;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// prefill, d = 128, third generation (r05).  The arithmetic -- every product, rounding, max / sum update and its order -- is that
// of lia_attn_prefill_kernel<128> and of the second generation (tools/attn_prefill_gen2.inc keeps that one;
// tools/attn_prefill_bench compares the two bit for bit on OPT-30B's, Llama-3-8B's and ragged shapes).  What changed is how the work
// is laid out, from per-wave cycle stamps and SQ counters of the second generation (LABNOTES.md r05: VALU 50 % busy, waves 40 % of
// their life in waits, SIMD 3 / XCD 7 holding every heavy piece of the causal triangle):
//   * K and V tiles (64 keys) go from global memory straight to LDS (global_load_lds, 16 B per lane, the images' chunk permutations
//     applied on the source address), double-buffered, ONE barrier per tile; the stage after sweep 1's last tile is sweep 2's first;
//   * a two-block tile runs as A: q.k chain of block 0 (fragments of block 1 fetched under it), B: chain of block 1 under the
//     softmax arithmetic of block 0, C: P.V of block 0 under the arithmetic of block 1, D: P.V of block 1 -- interleaved by hand
//     (sched_barrier between the pieces): hipcc otherwise serialises chain, arithmetic, chain, arithmetic;
//   * the arithmetic works on pairs (v_pk_add / v_pk_mul / v_pk_fma_f32, one v_cvt_pk_bf16_f32 per pair), masks are selects
//     (the second generation's `a || (b && c)` became a branch per element), POST_SCALE is a template parameter;
//   * every other (batch row, head) takes its query blocks and the four row blocks of a workgroup in reverse order.
// OPT-30B's B 64 x T 256 x 56 heads: 433 -> 333 us; Llama-3-8B's B 128 x T 1024 x 32 / 8 heads: 5.51 -> 3.23 ms.
// V^T operand: ds_read_b64_tr_b16 from the row-major image off(row, ch) = 256 row + 16 (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)))
// (lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4-key x 16-dim block and receives one dim of the four keys).
// ---------------------------------------------------------------------------------------------
typedef short lia_v4s __attribute__((ext_vector_type(4)));
#define GL_AS1(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_AS3(p) ((__attribute__((address_space(3))) void*)(p))

#define P3_KOFF(i) (((i) & 3) + 8 * ((i) >> 2))
typedef float f32x2 __attribute__((ext_vector_type(2)));

// one v_cvt_pk_bf16_f32 for the pair (compiler-visible: inline asm would hide the MFMA -> VALU read hazard from the hazard recognizer)
typedef __bf16 p3_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t p3_cvt_pk(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, p3_bf16x2));
}
// two scores at the reference's rounding points: bf16 of q.k; with POST_SCALE the product by d^-0.5 and its rounding (values of rbf())
template <int POST_SCALE> __device__ __forceinline__ f32x2 p3_round2(float a, float b, float scaling) {
  uint32_t p = p3_cvt_pk(a, b);
  f32x2 r = {__uint_as_float(p << 16), __uint_as_float(p & 0xffff0000u)};
  if (POST_SCALE) {
    const f32x2 t = r * scaling;
    p = p3_cvt_pk(t.x, t.y);
    r = f32x2{__uint_as_float(p << 16), __uint_as_float(p & 0xffff0000u)};
  }
  return r;
}
// causal / length mask of elements 2 c, 2 c + 1 of the lane: key 32 kt + 4 h + P3_KOFF(i) is visible iff P3_KOFF(i) <= lim
__device__ __forceinline__ f32x2 p3_mask2(f32x2 r, int lim, int c) {
  return f32x2{(lim >= P3_KOFF(2 * c)) ? r.x : -INFINITY, (lim >= P3_KOFF(2 * c + 1)) ? r.y : -INFINITY};
}
// __expf(r - m) of a pair: the subtraction, the product by log2(e) and v_exp_f32, as __expf lowers it
__device__ __forceinline__ f32x2 p3_exp2(f32x2 r, float m) {
  const f32x2 t = (r - m) * 0x1.715476p+0f;
  return f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
}
// bf16(e / l) of a pair, packed: hoisted reciprocal + two residual corrections = the correctly rounded quotient (second generation)
__device__ __forceinline__ uint32_t p3_prob2(f32x2 r, float m, float l, float rl) {
  const f32x2 e = p3_exp2(r, m);
  const f32x2 nl = (f32x2)(-l), rr = (f32x2)rl;
  f32x2 q = e * rl;
  q = __builtin_elementwise_fma(__builtin_elementwise_fma(nl, q, e), rr, q);
  q = __builtin_elementwise_fma(__builtin_elementwise_fma(nl, q, e), rr, q);
  return p3_cvt_pk(q.x, q.y);
}

// r06: the same operations on TWO pairs at a time.  A pair's chain -- subtract, scale, exp, e * (1 / l) and the four residual
// fmas -- is ten dependent instructions, and behind every packed one hipcc must put an s_nop before its consumer: 75-99 s_nop
// per sweep-2 tile, each a 1.8 ns issue slot of the wave (tools/issue_model).  As 4-wide vector operations every stage
// legalises into two independent packed instructions side by side, so a stage's consumer is one instruction away and needs no
// s_nop: 79-85 -> 8-10 per tile (+11 vector instructions).  Same operations on the same values: bit-identical.  Sweep 2 only:
// in sweep 1 the 4-wide subtraction comes out as 32 single instructions instead of 16 packed ones and the gain is gone.
template <int POST_SCALE, int MASKED>
__device__ __forceinline__ f32x4 p3_round4(float a0, float a1, float a2, float a3, float scaling, int lim, int c) {
  uint32_t p0 = p3_cvt_pk(a0, a1), p1 = p3_cvt_pk(a2, a3);
  f32x4 r = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u), __uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
  if (POST_SCALE) {
    const f32x4 t = r * scaling;
    p0 = p3_cvt_pk(t.x, t.y);
    p1 = p3_cvt_pk(t.z, t.w);
    r = f32x4{__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u), __uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
  }
  if (MASKED) {      // pairs c and c + 1 (p3_mask2)
    r.x = (lim >= P3_KOFF(2 * c)) ? r.x : -INFINITY;
    r.y = (lim >= P3_KOFF(2 * c + 1)) ? r.y : -INFINITY;
    r.z = (lim >= P3_KOFF(2 * c + 2)) ? r.z : -INFINITY;
    r.w = (lim >= P3_KOFF(2 * c + 3)) ? r.w : -INFINITY;
  }
  return r;
}
__device__ __forceinline__ f32x4 p3_exp4(f32x4 r, float m) {
  const f32x4 t = (r - m) * 0x1.715476p+0f;
  return f32x4{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y), __builtin_amdgcn_exp2f(t.z), __builtin_amdgcn_exp2f(t.w)};
}
__device__ __forceinline__ uint2 p3_prob4(f32x4 r, float m, float l, float rl) {
  const f32x4 e = p3_exp4(r, m);
  const f32x4 nl = (f32x4)(-l), rr = (f32x4)rl;
  f32x4 q = e * rl;
  q = __builtin_elementwise_fma(__builtin_elementwise_fma(nl, q, e), rr, q);
  q = __builtin_elementwise_fma(__builtin_elementwise_fma(nl, q, e), rr, q);
  return uint2{p3_cvt_pk(q.x, q.y), p3_cvt_pk(q.z, q.w)};
}

template <int POST_SCALE>
__global__ __launch_bounds__(256, 2) void lia_attn_prefill128_kernel(const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ kc,
                                                                  const bf16_t* __restrict__ vc, bf16_t* __restrict__ out, long ldo, int T,
                                                                  int heads, int kv_heads, long kv_row, long kv_batch, int b0, float scaling,
                                                                  int nqb, int n_groups) {
  constexpr int D = 128;
  __shared__ __attribute__((aligned(16))) char lds[2][2][64 * 256];   // [buffer][0 = K, 1 = V][key row of 256 bytes]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // r06: XCD-aware order (n_groups > 0).  The workgroups that read the same K / V -- every query block of the G query heads of one
  // (batch row, KV head) -- form a group of nqb * G members; group g runs on XCD g % 8 (workgroup ids are dealt round-robin over
  // the XCDs: id % 8), its members in consecutive slots of that XCD.  A K / V tile is then fetched into ONE L2 and found there by
  // the other members (and by the second sweep) instead of crossing the fabric once per XCD: the per-tile wait is an L2 hit, which
  // the one-tile-deep LDS-DMA prefetch covers; a miss (2-3 us under load) it does not (a tile's arithmetic is ~1.2 us).
  // OPT-30B T 2016: 1872 -> 1512 us per 8-row minibatch, bit-identical (tools/attn_prefill_bench, -DLIA_ATTN_LEGACY_GRID for r05's).
  // n_groups <= 0: r05's order -- consecutive ids = the query blocks of one head -- kept for launches whose groups do not spread
  // evenly over the 8 XCDs (the launcher decides).
  int bx, hh, b;
  if (n_groups > 0) {
    const int Gq = heads / kv_heads;
    const int nmem = nqb * Gq;
    const int slot = (int)(blockIdx.x >> 3);
    const int grp = (slot / nmem) * 8 + (int)(blockIdx.x & 7);
    if (grp >= n_groups) return;                      // (the grid is padded to a multiple of 8 groups)
    const int mem = slot % nmem;
    bx = mem % nqb;
    b = grp / kv_heads;
    hh = (grp % kv_heads) * Gq + mem / nqb;
  } else {
    bx = (int)(blockIdx.x % nqb);
    hh = (int)((blockIdx.x / nqb) % heads);
    b = (int)(blockIdx.x / ((unsigned)nqb * heads));
  }
  // every other (batch row, head) takes its query blocks -- and, below, the four row blocks of a workgroup -- in reverse order:
  // wave w of a workgroup goes to SIMD w, and late rows have more keys to visit, so without it SIMD 3 would get every heavy piece
  // and SIMD 0 every light one (with r05's grid: XCD 7 / XCD 0 as well)
  const int rev = (hh + b) & 1;
  // r06: a workgroup takes TWO query blocks of its head, block bx and its mirror image nqb_all - 1 - bx (the launcher passes
  // nqb = ceil(nqb_all / 2) then; nqb == nqb_all: one block per workgroup as before).  A light block and a heavy one: every
  // workgroup of a sequence costs the same (causal: 1 + ... keys), half as many workgroups are launched and drained -- the
  // launches alone were 75 us of the 313 us of OPT-30B's T = 256 launch (the kernel without loads, arithmetic and stores,
  // LABNOTES r06) -- and the second block's K / V tiles were just read by the first.
  const int nqb_all = (T + 127) >> 7;
  const int npass = (nqb != nqb_all && nqb_all - 1 - bx != bx) ? 2 : 1;
  const int wsub = rev ? 3 - wave : wave;
  const int kh = hh / (heads / kv_heads);
  const bf16_t* kbase = kc + (long)(b0 + b) * kv_batch + (long)kh * D;
  const bf16_t* vbase = vc + (long)(b0 + b) * kv_batch + (long)kh * D;
  const float qscale = POST_SCALE ? 1.0f : scaling;

  // ---- staging by LDS-DMA: instruction j of wave w moves key rows 16 j + 4 w + (lane >> 4) of the tile, one 16-byte chunk per lane,
  // to consecutive LDS bytes (1 KB per instruction); the images' chunk permutations are applied on the SOURCE side:
  //   K: chunk c of row R sits in slot c ^ (R & 15);  V: in slot c ^ (((R & 3) << 2) | ((R >> 2) & 3))   (R & 15 = 4 w + (lane >> 4))
  // r06: the address of a row is a UNIFORM base (first row of the instruction's 16, scalar 64-bit arithmetic) plus a 32-bit
  // per-lane byte offset (row within the 16, clamped to the sequence's last row, + chunk): the global saddr form.  r05 computed
  // min(key, T - 1) * kv_row per lane in 64 bits -- ~36 of a tile's ~230 vector instructions in a VALU-bound loop (read in the ISA).
  const int srow = 4 * wave + (lane >> 4), sslot = lane & 15;
  // Second form (r06): ONE scalar base per tile (its first row) and the lane's row within the 64 clamped against one scalar limit,
  // min(srow + 16 j, T - 1 - 64 t) rows -- the per-instruction scalar arithmetic of the first form (a 64-bit product, two minima and
  // two adds, x 4) was 77-84 scalar instructions per tile, and scalar instructions cost 1.8 ns each WITHOUT overlapping between the
  // two waves of a SIMD (tools/issue_model): now 23-30.
  const unsigned kvrow_b = (unsigned)kv_row * 2u;                  // bytes per key row (the launcher checks 64 rows fit 32 bits)
  unsigned rowoff_b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rowoff_b[j] = (unsigned)(srow + 16 * j) * kvrow_b;
  const unsigned kch_b = 16u * (unsigned)(sslot ^ srow);
  const unsigned vch_b = 16u * (unsigned)(sslot ^ (((srow & 3) << 2) | ((srow >> 2) & 3)));
  // (measurement switches of tools/attn_prefill_bench, never defined in the library build: -DP3X_NO_LOADS leaves the K / V stages
  // out, -DP3X_NO_MATH the tiles' arithmetic, -DP3X_NO_WAIT the wait for a stage (wrong results) -- where a tile step's time goes, LABNOTES r06)
#ifdef P3X_NO_WAIT
#define P3X_WAIT() do {} while (0)
#else
#define P3X_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
#ifdef P3X_NO_LOADS
#define P3X_NJ 0
#else
#define P3X_NJ 4
#endif
#define P3_ISSUE(t, buf, WITH_V)                                                                                                   \
  do {                                                                                                                             \
    int t0_ = (t) * 64;                                                 /* uniform; <= T - 1: the tile holds a valid row */        \
    asm volatile("" : "+s"(t0_));   /* (opaque: or loop strength reduction turns the bases into per-lane 64-bit pointers) */       \
    const unsigned lim_ = (unsigned)min(T - 1 - t0_, 63) * kvrow_b;     /* min(t0 + row, T - 1) = t0 + min(row, T - 1 - t0) */     \
    const long base_ = (long)t0_ * (long)kvrow_b;                                                                                  \
    _Pragma("unroll") for (int j_ = 0; j_ < P3X_NJ; ++j_) {                                                                        \
      const unsigned ro_ = min(rowoff_b[j_], lim_);                                                                                \
      __builtin_amdgcn_global_load_lds(GL_AS1((const char*)kbase + base_ + (ro_ + kch_b)),                                        \
                                       LDS_AS3(&lds[buf][0][(16 * j_ + 4 * wave) * 256]), 16, 0, 0);                               \
      if (WITH_V)                                                                                                                  \
        __builtin_amdgcn_global_load_lds(GL_AS1((const char*)vbase + base_ + (ro_ + vch_b)),                                      \
                                         LDS_AS3(&lds[buf][1][(16 * j_ + 4 * wave) * 256]), 16, 0, 0);                             \
    }                                                                                                                              \
  } while (0)

  for (int pass = 0; pass < npass; ++pass) {
  // (the lane-derived LDS offsets below are the same in both passes; behind this opaque copy hipcc recomputes them per pass
  // instead of keeping ~20 more registers alive through both sweeps: as a plain loop the kernel needed 256 registers + scratch
  // and ran 19 % slower than without the loop)
  int lane_p_ = lane;
  asm volatile("" : "+v"(lane_p_));
  const int lane = lane_p_;
  const int r = lane & 31, h = lane >> 5;
  // (the blocks in reverse order for every other (batch row, head), as the waves: see `rev`)
  const int bxp = (nqb != nqb_all) ? (((pass ^ rev) & 1) ? nqb_all - 1 - bx : bx) : (rev ? nqb - 1 - bx : bx);
  const int q_wg = bxp * 128;
  const int q_wave = q_wg + wsub * 32;
  // (the first block's output staging lives in stage buffer 0, where the second block's first K tile goes)
  if (pass) __syncthreads();
  // r06: the first K tile is requested BEFORE the query rows -- the LDS-DMA needs no register, and the workgroup's two first
  // round trips to memory (q rows into registers, K tile 0 into LDS) then overlap instead of following each other (~1.5 us of a
  // ~16 us workgroup at T = 256: the prologue was ~5 us of it, LABNOTES r05)
  P3_ISSUE(0, 0, false);

  bf16x8 qf[8];
  {
    const int qrow = min(q_wave + r, T - 1);
    const bf16_t* qp = q + ((long)b * T + qrow) * ldq + (long)hh * D;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      uint4 v = *(const uint4*)(qp + 16 * s + 8 * h);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(bf2f(w[j] & 0xffff) * qscale, bf2f(w[j] >> 16) * qscale);
      qf[s] = __builtin_bit_cast(bf16x8, uint4{o[0], o[1], o[2], o[3]});
    }
  }

  const int n32 = min((q_wg + 128 + 31) / 32, (T + 31) / 32);   // 32-key blocks the workgroup needs (causal)
  const int n64 = (n32 + 1) / 2;
  const int my_q = q_wave + r;
  const int nblk = min(q_wave + 31, T - 1) / 32 + 1;            // 32-key blocks THIS wave needs: 0 .. nblk - 1
  const int vis = min(my_q, T - 1) - 4 * h;                     // key 32 kt + 4 h + koff is visible iff koff <= vis - 32 kt

  // K^T fragment of 16-dim slice ss: key row 32 sb + r, chunk (2 ss + h) ^ (r & 15)
  int k_rd[8];
#pragma unroll
  for (int ss = 0; ss < 8; ++ss) k_rd[ss] = r * 256 + (((2 * ss + h) ^ (r & 15)) << 4);
  // V^T transposed read (ds_read_b64_tr_b16), as in the second generation
  const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
  int v_rd[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int key = 4 * h + 8 * u + tq;
      const int ch = 4 * d + 2 * g1 + (tp >> 1);
      v_rd[u][d] = 256 * key + 16 * (ch ^ (((key & 3) << 2) | ((key >> 2) & 3))) + 8 * (tp & 1);
    }

#define P3_KFRAGS(kf, kb, sb)                                                                                              \
  do { _Pragma("unroll") for (int ss_ = 0; ss_ < 8; ++ss_) kf[ss_] = *(const uint4*)((kb) + (sb) * 8192 + k_rd[ss_]); } while (0)
#define P3_QK(kf, sacc)                                                                                                    \
  do {                                                                                                                     \
    sacc = f32x16{0};                                                                                                      \
    _Pragma("unroll") for (int ss_ = 0; ss_ < 8; ++ss_)                                                                    \
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[ss_]), qf[ss_], sacc, 0, 0, 0);         \
  } while (0)

  // ---- sweep 1: row max and sum of exp over the visible keys ----
  // statistics of a block held as eight rounded pairs: the first generation's operations in its order (its `if (mn > -inf)` is
  // always taken: key 0 is visible to every query, so the running maximum is finite from block 0 on)
#define P3_STATS(rp)                                                                                                       \
  do {                                                                                                                     \
    float tm_ = fmaxf(rp[0].x, rp[0].y);                                                                                   \
    _Pragma("unroll") for (int c_ = 1; c_ < 8; ++c_) tm_ = fmaxf(tm_, fmaxf(rp[c_].x, rp[c_].y));                          \
    tm_ = fmaxf(tm_, __shfl_xor(tm_, 32, 64));                                                                             \
    const float mn_ = fmaxf(m, tm_);                                                                                       \
    float ts_ = 0.f;                                                                                                       \
    _Pragma("unroll") for (int c_ = 0; c_ < 8; ++c_) {                                                                     \
      const f32x2 e_ = p3_exp2(rp[c_], mn_);                                                                               \
      ts_ += e_.x;                                                                                                         \
      ts_ += e_.y;                                                                                                         \
    }                                                                                                                      \
    ts_ += __shfl_xor(ts_, 32, 64);                                                                                        \
    l = l * __expf(m - mn_) + ts_;                                                                                         \
    m = mn_;                                                                                                               \
  } while (0)
#define P3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
  // chain 0 of a two-block tile with the K^T fragments of block 1 fetched under it
#define P3_CHAIN0()                                                                                                        \
  P3_KFRAGS(kf0, kb, 0);                                                                                                   \
  _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                          \
    s0 = P3_MFMA(__builtin_bit_cast(bf16x8, kf0[c]), qf[c], s0);                                                           \
    kf1[c] = *(const uint4*)(kb + 8192 + k_rd[c]);                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
  }                                                                                                                        \
  __builtin_amdgcn_sched_barrier(0)
  // two visible blocks: chain 1 runs under the statistics of block 0 (never masked here) -- MFMAs 0..3 beside the rounding and the
  // maximum, 4..7 beside the sum of exp; then block 1, always through the mask (it may hold the diagonal or the end of the sequence)
#ifdef P3X_NO_MATH
#define P3X_MATH 0
#else
#define P3X_MATH 1
#endif
#define P3_S1_TILE2(MASKED)                                                                                                \
  if (P3X_MATH) do {                                                                                                                     \
    uint4 kf0[8], kf1[8];                                                                                                  \
    f32x16 s0 = f32x16{0}, s1 = f32x16{0};                                                                                 \
    f32x2 rp[8];                                                                                                           \
    float tm = -INFINITY, mn = 0.f, ts = 0.f;                                                                              \
    P3_CHAIN0();                                                                                                           \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                        \
      s1 = P3_MFMA(__builtin_bit_cast(bf16x8, kf1[c]), qf[c], s1);                                                         \
      if (c < 4) {                                                                                                         \
        rp[2 * c] = p3_round2<POST_SCALE>(s0[4 * c], s0[4 * c + 1], scaling);                                              \
        rp[2 * c + 1] = p3_round2<POST_SCALE>(s0[4 * c + 2], s0[4 * c + 3], scaling);                                      \
        tm = fmaxf(fmaxf(tm, fmaxf(rp[2 * c].x, rp[2 * c].y)), fmaxf(rp[2 * c + 1].x, rp[2 * c + 1].y));                   \
        if (c == 3) {                                                                                                      \
          tm = fmaxf(tm, __shfl_xor(tm, 32, 64));                                                                          \
          mn = fmaxf(m, tm);                                                                                               \
        }                                                                                                                  \
      } else {                                                                                                             \
        const f32x2 e0 = p3_exp2(rp[2 * (c - 4)], mn), e1 = p3_exp2(rp[2 * (c - 4) + 1], mn);                              \
        ts += e0.x; ts += e0.y; ts += e1.x; ts += e1.y;                                                                    \
      }                                                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
    }                                                                                                                      \
    ts += __shfl_xor(ts, 32, 64);                                                                                          \
    l = l * __expf(m - mn) + ts;                                                                                           \
    m = mn;                                                                                                                \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                        \
      rp[c] = p3_round2<POST_SCALE>(s1[2 * c], s1[2 * c + 1], scaling);                                                    \
      if (MASKED) rp[c] = p3_mask2(rp[c], lim1, c);                                                                        \
    }                                                                                                                      \
    P3_STATS(rp);                                                                                                          \
  } while (0)

  float m = -INFINITY, l = 0.f;
  // r06: both sweeps are FOUR loops instead of one loop with three cases.  A wave has two visible key blocks in tiles 0 .. nfull - 1
  // (nfull = nblk / 2), one in tile nfull when nblk is odd (the diagonal / the end of the sequence), none behind it -- but every
  // wave meets every barrier and issues its share of every stage; and in tiles 0 .. nfree - 1 the LAST key of the tile is visible
  // to the wave's FIRST query (64 t + 63 <= min(q_wave, T - 1)), so block 1's mask selects nothing there and is left out (two
  // v_cmp + two v_cndmask per pair: 12-14 % of a tile's vector instructions).  As ONE loop (`if (nb >= 2) ... else if (nb == 1)`)
  // hipcc also merged sweep 2's 64 accumulator registers at the latch: 64 v_mov out of the two-block body and 64 back on the
  // back-edge, 128 of ~400 vector instructions per iteration (found in the ISA).  Same operations on the same values in the same order.
  const int nfull = nblk >> 1;                                     // (<= n64: nblk <= n32 <= 2 n64)
  const int nfree = min((min(q_wave, T - 1) + 1) >> 6, nfull);
#define P3_S1_STEP()                                                                                                       \
  P3X_WAIT();                                                                                                              \
  __syncthreads();                                                                                                         \
  /* (the stage after sweep 1's last tile is sweep 2's first: its K and V ride under this tile's arithmetic) */           \
  if (t + 1 < n64) P3_ISSUE(t + 1, (t + 1) & 1, false);                                                                    \
  else P3_ISSUE(0, n64 & 1, true);                                                                                         \
  const char* kb = &lds[t & 1][0][0];                                                                                      \
  const int lim1 = vis - 32 * (2 * t + 1)
  {
    int t = 0;
    for (; t < nfree; ++t) {
      P3_S1_STEP();
      P3_S1_TILE2(0);
      (void)lim1;
    }
    for (; t < nfull; ++t) {
      P3_S1_STEP();
      P3_S1_TILE2(1);
    }
    if (t < n64) {
      P3_S1_STEP();
      if (nblk & 1) {
        uint4 kf0[8];
        f32x16 s0;
        f32x2 rp[8];
        P3_KFRAGS(kf0, kb, 0);
        P3_QK(kf0, s0);
#pragma unroll
        for (int c = 0; c < 8; ++c) rp[c] = p3_mask2(p3_round2<POST_SCALE>(s0[2 * c], s0[2 * c + 1], scaling), lim1 + 32, c);
        P3_STATS(rp);
      }
      ++t;
    }
    for (; t < n64; ++t) {
      P3_S1_STEP();
      (void)kb; (void)lim1;
    }
  }
#undef P3_S1_STEP

  // ---- sweep 2: P = bf16(exp(s - m) / l), O^T += V^T . P^T ----
  const float rl0 = __builtin_amdgcn_rcpf(l);
  const float rl = __builtin_fmaf(__builtin_fmaf(-l, rl0, 1.0f), rl0, rl0);
  f32x16 oacc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) oacc[d] = f32x16{0};
  // V^T piece c = (ks = c >> 2, d = c & 3) of block sb: keys 32 sb + 16 ks + 4 h + (0..3) -> elements 0..3, the same + 8 -> elements
  // 4..7 (the k order of an accumulator tile used as an operand); dims 32 d ..
#define P3_VFRAG(dst, vb, sb, c)                                                                                            \
  do {                                                                                                                     \
    const char* vt_ = (vb) + (32 * (sb) + 16 * ((c) >> 2)) * 256;                                                          \
    lia_v4s lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) lia_v4s*)(vt_ + v_rd[0][(c) & 3])); \
    lia_v4s hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) lia_v4s*)(vt_ + v_rd[1][(c) & 3])); \
    dst = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7));                           \
  } while (0)
#define P3_PF(pk, c) __builtin_bit_cast(bf16x8, uint4{pk[4 * ((c) >> 2)], pk[4 * ((c) >> 2) + 1], pk[4 * ((c) >> 2) + 2], pk[4 * ((c) >> 2) + 3]})
  // two visible blocks: chain 1 under the softmax of block 0, P.V of block 0 under the softmax of block 1, then P.V of block 1; the
  // V^T pieces are fetched one phase ahead.  Every oacc[d] takes its four products in the order (block 0, ks 0), (0, 1), (1, 0), (1, 1).
#define P3_S2_TILE2(MASKED)                                                                                                \
  if (P3X_MATH) do {                                                                                                                     \
    uint4 kf0[8], kf1[8];                                                                                                  \
    f32x16 s0 = f32x16{0}, s1 = f32x16{0};                                                                                 \
    uint32_t pk0[8], pk1[8];                                                                                               \
    bf16x8 vfa[8], vfb[8];                                                                                                 \
    P3_CHAIN0();                                                                                                           \
    _Pragma("unroll") for (int c2 = 0; c2 < 4; ++c2) {                                                                     \
      _Pragma("unroll") for (int c = 2 * c2; c < 2 * c2 + 2; ++c) {                                                        \
        s1 = P3_MFMA(__builtin_bit_cast(bf16x8, kf1[c]), qf[c], s1);                                                       \
        P3_VFRAG(vfa[c], vb, 0, c);                                                                                        \
      }                                                                                                                    \
      {                                                                                                                    \
        const uint2 p_ = p3_prob4(p3_round4<POST_SCALE, 0>(s0[4 * c2], s0[4 * c2 + 1], s0[4 * c2 + 2], s0[4 * c2 + 3], scaling, 0, 0), m, l, rl); \
        pk0[2 * c2] = p_.x;                                                                                                \
        pk0[2 * c2 + 1] = p_.y;                                                                                            \
      }                                                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
    }                                                                                                                      \
    _Pragma("unroll") for (int c2 = 0; c2 < 4; ++c2) {                                                                     \
      _Pragma("unroll") for (int c = 2 * c2; c < 2 * c2 + 2; ++c) {                                                        \
        oacc[c & 3] = P3_MFMA(vfa[c], P3_PF(pk0, c), oacc[c & 3]);                                                         \
        P3_VFRAG(vfb[c], vb, 1, c);                                                                                        \
      }                                                                                                                    \
      {                                                                                                                    \
        const uint2 p_ = p3_prob4(p3_round4<POST_SCALE, MASKED>(s1[4 * c2], s1[4 * c2 + 1], s1[4 * c2 + 2], s1[4 * c2 + 3], scaling, lim1, 2 * c2), m, l, rl); \
        pk1[2 * c2] = p_.x;                                                                                                \
        pk1[2 * c2 + 1] = p_.y;                                                                                            \
      }                                                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
    }                                                                                                                      \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) oacc[c & 3] = P3_MFMA(vfb[c], P3_PF(pk1, c), oacc[c & 3]);               \
  } while (0)

  // (four loops, as sweep 1)
#define P3_S2_STEP()                                                                                                       \
  P3X_WAIT();                                                                                                              \
  __syncthreads();                                                                                                         \
  if (t + 1 < n64) P3_ISSUE(t + 1, (n64 + t + 1) & 1, true);                                                               \
  const char* kb = &lds[(n64 + t) & 1][0][0];                                                                              \
  const char* vb = &lds[(n64 + t) & 1][1][0];                                                                              \
  const int lim1 = vis - 32 * (2 * t + 1)
  int t = 0;
  for (; t < nfree; ++t) {
    P3_S2_STEP();
    P3_S2_TILE2(0);
    (void)lim1;
  }
  for (; t < nfull; ++t) {
    P3_S2_STEP();
    P3_S2_TILE2(1);
  }
  if (t < n64) {
    P3_S2_STEP();
    if (nblk & 1) {
      uint4 kf0[8];
      f32x16 s0;
      uint32_t pk0[8];
      bf16x8 vfa[8];
      P3_KFRAGS(kf0, kb, 0);
      P3_QK(kf0, s0);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        P3_VFRAG(vfa[c], vb, 0, c);
        pk0[c] = p3_prob2(p3_mask2(p3_round2<POST_SCALE>(s0[2 * c], s0[2 * c + 1], scaling), lim1 + 32, c), m, l, rl);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) oacc[c & 3] = P3_MFMA(vfa[c], P3_PF(pk0, c), oacc[c & 3]);
    }
    (void)vb;
    ++t;
  }
  for (; t < n64; ++t) {
    P3_S2_STEP();
    (void)kb; (void)vb; (void)lim1;
  }
#undef P3_S2_STEP
#undef P3_S2_TILE2
#undef P3_S1_TILE2
#undef P3_CHAIN0
#undef P3_STATS
#undef P3_PF
#undef P3_MFMA
#undef P3_VFRAG
#undef P3_QK
#undef P3_KFRAGS
#undef P3_ISSUE

  // ---- output: through the LDS, so that a store instruction writes whole rows ----
  // A lane holds 4 consecutive dims of ONE query per accumulator register group: stored from the registers, an instruction
  // writes 16 bytes to each of 32 rows (r05: 16 such instructions per wave, 512 partial-line writes).  Measured with the stores
  // left out (-DP3X_NO_STORE): 66 of the 330 us of OPT-30B's T = 256 launch, 70 of Llama-3-8B's 901.  r06: each wave parks its
  // 32 rows x 256 bytes in its own 8 KB of stage buffer 0 -- free since the barrier of the last step: the last tile of sweep 2
  // lives in buffer (2 n64 - 1) & 1 = 1 and no stage was issued behind it -- 16-byte chunks XOR-swizzled by the row as the K
  // image, and reads them back row-major: 8 instructions of 4 rows x 256 contiguous bytes.  Wave-private, in order: no barrier.
  {
    char* ob = &lds[0][0][0] + wave * 8192;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 o;
        o.x = pack_bf16x2(oacc[d][4 * g], oacc[d][4 * g + 1]);
        o.y = pack_bf16x2(oacc[d][4 * g + 2], oacc[d][4 * g + 3]);
        *(uint2*)(ob + r * 256 + (((4 * d + g) ^ (r & 15)) << 4) + 8 * h) = o;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int orow = lane >> 4, oc = lane & 15;
#ifdef P3X_NO_STORE
    const bool st_on = scaling == 12345.f;
#else
    const bool st_on = true;
#endif
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rr = 4 * i + orow;
      const uint4 v = *(const uint4*)(ob + rr * 256 + ((oc ^ (rr & 15)) << 4));
      if (q_wave + rr < T && st_on) *(uint4*)(out + ((long)b * T + q_wave + rr) * ldo + (long)hh * D + 8 * oc) = v;
    }
  }
  }   // pass
}


// ---------------------------------------------------------------------------------------------
// decode (T == 1): one workgroup per (batch row, KV head); KV-bandwidth bound.  The G = heads / kv_heads query
// heads that share a KV head (grouped-query attention; G = 1 for OPT) are served together, so every K/V row is
// read from HBM once.  LPK = D/8 lanes share one key row (16 bytes each); scores are parked in LDS, then every
// thread accumulates its 8 output dims over its share of the keys and the shares are combined through LDS.
// ---------------------------------------------------------------------------------------------
// a 16-byte load of cache rows that are read once per step: non-temporal, so the stream does not push q, the scores' neighbours
// and the next kernel's operands out of L2 (tools/stream_bench: a once-read stream runs 6.7 TB/s with nt against 6.0 without)
__device__ __forceinline__ uint4 lia_ldg_stream(const bf16_t* p) {
  const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
  return uint4{v[0], v[1], v[2], v[3]};
}

template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

template <int D, int G>
__global__ __launch_bounds__(256) void lia_attn_decode_kernel(const bf16_t* __restrict__ q, long ldq,
                                                               const bf16_t* __restrict__ kc,
                                                               const bf16_t* __restrict__ vc, bf16_t* __restrict__ out,
                                                               long ldo, int S, int heads, int kv_heads, int Bc, int b0,
                                                               float scaling, int post_scale) {
  constexpr int LPK = D / 8;          // lanes per key
  constexpr int KPP = 256 / LPK;      // keys per pass of the workgroup
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Spad = (S + 3) & ~3;
  float* sc = (float*)smem;                                   // [G][Spad] scores -> probabilities
  float* red = sc + (size_t)G * Spad;                         // [4 waves][G][D] partial outputs
  float* scratch = red + (size_t)4 * G * D;                   // [G][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kh = blockIdx.x, b = blockIdx.y;
  const long kv_row = (long)Bc * kv_heads * D;
  const bf16_t* kbase = kc + ((long)(b0 + b) * kv_heads + kh) * D;
  const bf16_t* vbase = vc + ((long)(b0 + b) * kv_heads + kh) * D;
  const float qscale = post_scale ? 1.0f : scaling;
  const int sub = tid % LPK, kslot = tid / LPK;

  // the (scaled, bf16-rounded) query slice of each head of the group as packed bf16 pairs: q . k runs on v_dot2c_f32_bf16,
  // two products per instruction and no unpacking of the key (the products of two bf16 are exact in fp32 either way; the
  // sum is rounded to bf16 right after): Llama-3-8B's grouped decode attention 119.6 -> 114.5 us per layer.  (Requesting
  // the next pass's rows before multiplying this pass's was tried too: 122 us -- four workgroups per CU already overlap.)
  typedef __attribute__((ext_vector_type(2))) __bf16 lia_bf16x2;
  uint32_t qp[G][4];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    uint4 v = *(const uint4*)(q + (long)b * ldq + (long)(kh * G + g) * D + 8 * sub);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) qp[g][j] = pack_bf16x2(bf2f(w[j] & 0xffff) * qscale, bf2f(w[j] >> 16) * qscale);
  }
  float lmax[G];
#pragma unroll
  for (int g = 0; g < G; ++g) lmax[g] = -INFINITY;
  constexpr int U = 8;   // key rows in flight per thread: KV reads are the whole cost, keep several 16-byte loads outstanding (4 -> 8: +2 % on Llama-3-8B decode)
  // One pass = U * KPP = 128 keys, every row of it requested before the first is used.  Full passes carry no bounds checks; the
  // last pass (S = prompt + 1 + step sits just behind a multiple of 128 for every usual prompt length: 1 ... 32 real keys) requests
  // only the rows that exist -- r03 clamped its other ~100 slots onto row S - 1, a hundred requests for one row per workgroup
  // (L2 hits, but 1/3 more requests at S = 257).  r04, OPT-30B decode: 83.3 -> 80.0 us.
  auto k_pass = [&](const int j0, auto tail) {
    constexpr bool TAIL = decltype(tail)::value;
    uint4 kv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * KPP + kslot;
      if constexpr (TAIL) {
        kv[u] = uint4{0u, 0u, 0u, 0u};
        if (j < S) kv[u] = lia_ldg_stream(kbase + (long)j * kv_row + 8 * sub);
      } else {
        kv[u] = lia_ldg_stream(kbase + (long)j * kv_row + 8 * sub);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * KPP + kslot;
      const uint32_t w[4] = {kv[u].x, kv[u].y, kv[u].z, kv[u].w};
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(lia_bf16x2, qp[g][e]), __builtin_bit_cast(lia_bf16x2, w[e]), a, false);
        if constexpr (LPK == 16) {
          // the 16 lanes of a key are one DPP row: the sum over them with row rotations / quad permutes as VALU operand
          // modifiers instead of four ds_bpermute_b32 per value (PMC, Llama-3-8B B 128: the LDS pipe was ~77 % busy with them).
          // Rotating by 8 and 4 instead of xor 8, 4 picks a partner that holds the same partial (same lane index mod 8 resp.
          // mod 4), so every lane still ends with ((p_l + p_l^8) + (p_l^4 + p_l^12)) + ...: the bits of the xor butterfly.
          a += dpp_f32<0x128>(a);     // row_ror:8
          a += dpp_f32<0x124>(a);     // row_ror:4
          a += dpp_f32<0x4E>(a);      // quad_perm:[2,3,0,1]  (xor 2)
          a += dpp_f32<0xB1>(a);      // quad_perm:[1,0,3,2]  (xor 1)
        } else {
#pragma unroll
          for (int o = LPK / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        }
        if (!TAIL || j < S) {
          float sv = rbf(a);
          if (post_scale) sv = rbf(sv * scaling);
          if (sub == 0) sc[g * Spad + j] = sv;
          lmax[g] = fmaxf(lmax[g], sv);
        }
      }
    }
  };
  const int s_full = S - S % (U * KPP);
  for (int j0 = 0; j0 < s_full; j0 += U * KPP) k_pass(j0, std::false_type{});
  if (s_full < S) k_pass(s_full, std::true_type{});
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float m = wave_max(lmax[g]);
    if (lane == 0) scratch[g * 8 + wave] = m;
  }
  __syncthreads();
  float mrow[G], lrow[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    mrow[g] = fmaxf(fmaxf(scratch[g * 8], scratch[g * 8 + 1]), fmaxf(scratch[g * 8 + 2], scratch[g * 8 + 3]));
    float ls = 0.f;
    for (int j = tid; j < S; j += 256) {
      float e = __expf(sc[g * Spad + j] - mrow[g]);
      sc[g * Spad + j] = e;
      ls += e;
    }
    ls = wave_sum(ls);
    if (lane == 0) scratch[g * 8 + 4 + wave] = ls;
  }
  __syncthreads();
#pragma unroll
  for (int g = 0; g < G; ++g) lrow[g] = scratch[g * 8 + 4] + scratch[g * 8 + 5] + scratch[g * 8 + 6] + scratch[g * 8 + 7];
  // probabilities once per (head, key) -- bf16(e / l), the reference's rounding point -- instead of once per lane of the
  // key in the P.V loop below (LPK lanes would each repeat the division)
#pragma unroll
  for (int g = 0; g < G; ++g)
    for (int j = tid; j < S; j += 256) sc[g * Spad + j] = rbf(sc[g * Spad + j] / lrow[g]);
  __syncthreads();

  float o[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[g][e] = 0.f;
  auto v_pass = [&](const int j0, auto tail) {
    constexpr bool TAIL = decltype(tail)::value;
    uint4 vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (TAIL) {
        vv[u] = uint4{0u, 0u, 0u, 0u};
        if (j0 + u * KPP < S) vv[u] = lia_ldg_stream(vbase + (long)(j0 + u * KPP) * kv_row + 8 * sub);
      } else {
        vv[u] = lia_ldg_stream(vbase + (long)(j0 + u * KPP) * kv_row + 8 * sub);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * KPP;
      if (!TAIL || j < S) {
        const uint32_t w[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float p = sc[g * Spad + j];
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[g][2 * e] += p * bf2f(w[e] & 0xffff); o[g][2 * e + 1] += p * bf2f(w[e] >> 16); }
        }
      }
    }
  };
  for (int j0 = 0; j0 < s_full; j0 += U * KPP) v_pass(j0 + kslot, std::false_type{});
  if (s_full < S) v_pass(s_full + kslot, std::true_type{});
  // the key slots of one wave are folded with shuffles (a [KPP][G][D] LDS slab would cap the CU at three workgroups;
  // with [4][G][D] a B = 128, 8-kv-head launch is resident in one round), then the four waves through LDS
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = o[g][e];
#pragma unroll
      for (int off = LPK; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
      o[g][e] = v;
    }
  if (lane < LPK) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[((size_t)wave * G + g) * D + 8 * sub + e] = o[g][e];
  }
  __syncthreads();
  for (int idx = tid; idx < G * D; idx += 256) {
    const int g = idx / D, dd = idx - g * D;
    const float acc = (red[(size_t)g * D + dd] + red[((size_t)G + g) * D + dd]) +
                      (red[((size_t)2 * G + g) * D + dd] + red[((size_t)3 * G + g) * D + dd]);
    out[(long)b * ldo + (long)(kh * G + g) * D + dd] = f2bf(acc);
  }
}


extern "C" int lia_attn_prefill_launch(const bf16_t* q, long ldq, const bf16_t* kc, const bf16_t* vc, bf16_t* out, long ldo,
                                       int B, int T, int heads, int kv_heads, int d, int Bc, int b0, int post_scale,
                                       hipStream_t st) {
  if (B <= 0 || T <= 0) return 0;
  if (kv_heads <= 0 || heads % kv_heads) return -1;
  dim3 grid((T + 127) / 128, heads, B);
  const float scaling = 1.0f / sqrtf((float)d);
  switch (d) {
    case 128: {
      const long hd = (long)kv_heads * 128;
      // (a token-major [B][T][h][d] K/V -- strides (hd, T hd) -- was measured: same time, so the cache layout stays)
      const int nqb_all = (T + 127) / 128;
#ifdef LIA_ATTN_NO_PAIRS
      const int nqb = nqb_all;
#else
      const int nqb = (nqb_all + 1) / 2;               // two query blocks per workgroup: bx and its mirror image (see the kernel)
#endif
      // the staging addresses hold (row within a 64-key tile) x (bytes per key row) + chunk in 32 bits: 64 Bc hd 2 < 2^32, i.e. a
      // cache batch x hidden of < 3.3e7 elements per key row (OPT-175B at Bc = 2048 is 2.5e7)
      if ((long)Bc * hd * 2 * 64 + 256 >= (1L << 32)) return -1;
      int n_groups = B * kv_heads;
      // the XCD-aware order when the groups spread evenly over the 8 XCDs (a multiple of 8, or so many that the remainder does not
      // matter); otherwise r05's order, which deals single workgroups round-robin (2 rows x 2 KV heads would use 4 XCDs of 8)
      bool xcd_order = (n_groups % 8 == 0) || n_groups >= 64;
#ifdef LIA_ATTN_LEGACY_GRID
      xcd_order = false;
#endif
      const unsigned nwg = xcd_order ? (unsigned)(((n_groups + 7) / 8) * 8 * nqb * (heads / kv_heads)) : (unsigned)(nqb * heads * B);
      if (!xcd_order) n_groups = 0;
      const dim3 grid128(nwg);
      if (post_scale) hipLaunchKernelGGL(lia_attn_prefill128_kernel<1>, grid128, dim3(256), 0, st, q, ldq, kc, vc, out, ldo, T, heads, kv_heads, (long)Bc * hd, hd, b0, scaling, nqb, n_groups);
      else hipLaunchKernelGGL(lia_attn_prefill128_kernel<0>, grid128, dim3(256), 0, st, q, ldq, kc, vc, out, ldo, T, heads, kv_heads, (long)Bc * hd, hd, b0, scaling, nqb, n_groups);
      break;
    }
    case 64: hipLaunchKernelGGL(lia_attn_prefill_kernel<64>, grid, dim3(256), 0, st, q, ldq, kc, vc, out, ldo, T, heads, kv_heads, Bc, b0, scaling, post_scale); break;
    case 32: hipLaunchKernelGGL(lia_attn_prefill_kernel<32>, grid, dim3(256), 0, st, q, ldq, kc, vc, out, ldo, T, heads, kv_heads, Bc, b0, scaling, post_scale); break;
    default: return -1;
  }
  return 0;
}

template <int D>
static int launch_decode(const bf16_t* q, long ldq, const bf16_t* kc, const bf16_t* vc, bf16_t* out, long ldo, int B, int S, int heads,
                         int kv_heads, int Bc, int b0, float scaling, int post_scale, hipStream_t st) {
  const int G = heads / kv_heads;
  dim3 grid(kv_heads, B);
  const int kpp = 256 / (D / 8);
  const int Spad = (S + 3) & ~3;
  (void)kpp;
  size_t lds = ((size_t)G * Spad + (size_t)4 * G * D + (size_t)G * 8) * sizeof(float);
  if (lds > 160 * 1024) return -1;
#define LIA_DEC(GV)                                                                                                            \
  {                                                                                                                            \
    static bool attr = false;                                                                                                  \
    if (!attr) { (void)hipFuncSetAttribute((const void*)lia_attn_decode_kernel<D, GV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
    hipLaunchKernelGGL((lia_attn_decode_kernel<D, GV>), grid, dim3(256), lds, st, q, ldq, kc, vc, out, ldo, S, heads, kv_heads, Bc, b0, scaling, post_scale); \
  }
  switch (G) {
    case 1: LIA_DEC(1) break;
    case 2: LIA_DEC(2) break;
    case 4: LIA_DEC(4) break;
    case 8: LIA_DEC(8) break;
    default: return -1;
  }
#undef LIA_DEC
  return 0;
}

extern "C" int lia_attn_decode_launch(const bf16_t* q, long ldq, const bf16_t* kc, const bf16_t* vc, bf16_t* out, long ldo,
                                      int B, int S, int heads, int kv_heads, int d, int Bc, int b0, int post_scale,
                                      hipStream_t st) {
  if (B <= 0 || S <= 0) return 0;
  if (kv_heads <= 0 || heads % kv_heads) return -1;
  const float scaling = 1.0f / sqrtf((float)d);
  switch (d) {
    case 128: return launch_decode<128>(q, ldq, kc, vc, out, ldo, B, S, heads, kv_heads, Bc, b0, scaling, post_scale, st);
    case 64: return launch_decode<64>(q, ldq, kc, vc, out, ldo, B, S, heads, kv_heads, Bc, b0, scaling, post_scale, st);
    case 32: return launch_decode<32>(q, ldq, kc, vc, out, ldo, B, S, heads, kv_heads, Bc, b0, scaling, post_scale, st);
    default: return -1;
  }
}
