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
  const int q_wg = blockIdx.x * 128;
  const int q_wave = q_wg + wave * 32;
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
      uint4 v = *(const uint4*)(qp + 16 * s + 8 * h);
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
// prefill, d = 128, second generation.  The arithmetic -- every product, rounding, max / sum update and its order -- is
// that of lia_attn_prefill_kernel<128>, so the two kernels agree bit for bit; what changes is how the bytes move:
//   * 64-key tiles (two 32-key blocks per barrier pair instead of one);
//   * register-staged prefetch: the global loads of tile t+1 are issued before tile t is computed and written to LDS
//     after the next barrier (cdna_hip_programming.md T14), so HBM / L2 latency hides behind the MFMAs;
//   * V is staged row-major like K (two ds_write_b128 per thread instead of sixteen ds_write_b16) in the dual-use image
//     off(row, ch) = 256 row + 16 (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) of T10, and the V^T operand is read with
//     ds_read_b64_tr_b16: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4-key x 16-dim block and
//     receives one dim of the four keys.
// ---------------------------------------------------------------------------------------------
typedef short lia_v4s __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void lia_attn_prefill128_kernel(const bf16_t* __restrict__ q, long ldq,
                                                                   const bf16_t* __restrict__ kc,
                                                                   const bf16_t* __restrict__ vc, bf16_t* __restrict__ out,
                                                                   long ldo, int T, int heads, int kv_heads, long kv_row,
                                                                   long kv_batch, int b0, float scaling, int post_scale) {
  constexpr int D = 128;
  __shared__ __attribute__((aligned(16))) char k_lds[64 * 256];
  __shared__ __attribute__((aligned(16))) char v_lds[64 * 256];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int hh = blockIdx.y, b = blockIdx.z;
  const int q_wg = blockIdx.x * 128;
  const int q_wave = q_wg + wave * 32;
  const int kh = hh / (heads / kv_heads);
  // kv_row / kv_batch: elements between consecutive positions / batch rows of K and V.  Seq-major cache [S][Bc][h][d]:
  // (Bc h d, h d); token-major projection output [B][T][h][d]: (h d, T h d)
  const bf16_t* kbase = kc + (long)(b0 + b) * kv_batch + (long)kh * D;
  const bf16_t* vbase = vc + (long)(b0 + b) * kv_batch + (long)kh * D;
  const float qscale = post_scale ? 1.0f : scaling;

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

  // staging: thread -> key rr*16 + tid/16 of the tile, chunk tid%16 (a key row = 256 contiguous bytes over 16 lanes)
  const int skey = tid >> 4, sc = tid & 15;
  const int k_wr = skey * 256 + ((sc ^ skey) << 4);                                   // + rr * 4096 (key & 15 == skey)
  const int v_wr = skey * 256 + ((sc ^ (((skey & 3) << 2) | ((skey >> 2) & 3))) << 4);  // + rr * 4096 (16 rr keeps both fields)
  // K^T fragment read: key row 32 sb + r, chunk 2 s + h
  const int k_rd = r * 256;
  // V^T transposed read: group g1 = (lane >> 4) & 1 takes dims 16 g1 .. +15 of the 32-dim block; lane i = lane & 15 of the
  // group addresses key q = i >> 2, columns 4 p .. 4 p + 3 (p = i & 3)
  const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
  int v_rd[2][4];     // [u][d]: byte offset of (key 4 h + 8 u + q, dims 32 d + 16 g1 + 4 p) in the swizzled image
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int key = 4 * h + 8 * u + tq;
      const int ch = 4 * d + 2 * g1 + (tp >> 1);
      v_rd[u][d] = 256 * key + 16 * (ch ^ (((key & 3) << 2) | ((key >> 2) & 3))) + 8 * (tp & 1);
    }

  // q.k of one 32-key block: eight dependent MFMAs
#define P2_QK(sb, sacc)                                                                                               \
  do {                                                                                                                \
    sacc = f32x16{0};                                                                                                 \
    _Pragma("unroll") for (int ss_ = 0; ss_ < 8; ++ss_) {                                                             \
      uint4 kv_ = *(const uint4*)(k_lds + (sb) * 8192 + k_rd + (((2 * ss_ + h) ^ (r & 15)) << 4));                    \
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kv_), qf[ss_], sacc, 0, 0, 0);        \
    }                                                                                                                 \
  } while (0)
  // the reference's rounding points on the scores, then the causal mask (inside_: wave-uniform, nothing to mask)
#define P2_ROUND_MASK(kt32, sacc, inside_)                                                                            \
  do {                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                                  \
      int key_ = (kt32) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;                                                        \
      float sv_ = rbf(sacc[i]);                                                                                       \
      if (post_scale) sv_ = rbf(sv_ * scaling);                                                                       \
      sacc[i] = ((inside_) || (key_ <= my_q && key_ < T)) ? sv_ : -INFINITY;                                          \
    }                                                                                                                 \
  } while (0)
#define P2_SCORES(kt32, sb, sacc)                                                                                     \
  do {                                                                                                                \
    P2_QK(sb, sacc);                                                                                                  \
    const bool in_ = (kt32) * 32 + 31 <= q_wave && (kt32) * 32 + 31 < T;                                              \
    P2_ROUND_MASK(kt32, sacc, in_);                                                                                   \
  } while (0)
  // running row max / sum of exp with the block's scores (sweep 1)
#define P2_STATS(s)                                                                                                   \
  do {                                                                                                                \
    float tm = s[0];                                                                                                  \
    _Pragma("unroll") for (int i = 1; i < 16; ++i) tm = fmaxf(tm, s[i]);                                              \
    tm = fmaxf(tm, __shfl_xor(tm, 32, 64));                                                                           \
    float mn = fmaxf(m, tm);                                                                                          \
    if (mn > -INFINITY) {                                                                                             \
      float ts = 0.f;                                                                                                 \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) ts += __expf(s[i] - mn);                                         \
      ts += __shfl_xor(ts, 32, 64);                                                                                   \
      l = l * __expf(m - mn) + ts;                                                                                    \
      m = mn;                                                                                                         \
    }                                                                                                                 \
  } while (0)
  // A 64-key tile that lies entirely at or below the wave's diagonal (all but the last one or two tiles of a wave) runs both
  // blocks' q.k chains FIRST and only then the VALU work on block 0, which overlaps the matrix pipe still busy with block 1
  // (then P.V of block 0 under the VALU work of block 1): same operations, same order within a block and between the
  // blocks' updates of (m, l) and of the output accumulators -- the bits do not change.  PMC before: matrix pipe 10 % busy,
  // VALU 30 %, two waves per SIMD (222 of 512 registers each) stalled on one another's latencies; B 128 x T 1024 x 32 heads
  // 7.25 -> 5.65 ms, OPT-30B's B 64 x T 256 x 56 heads 0.478 -> 0.409 ms.  (The same interleave for the partly masked tiles,
  // with the mask flags as run-time values, needs 294 registers -- one wave per SIMD, 12.7 ms -- or spills at 256: 7.9 ms.)
#define P2_TILE_INSIDE(t) ((2 * (t) + 1) * 32 + 31 <= q_wave && (2 * (t) + 1) * 32 + 31 < T)
  // (A second such path for the diagonal tile -- block 0 visible, block 1 masked, flags still compile-time -- takes the kernel to
  // 314 registers: one wave per SIMD, 8.6 ms; capped at 256 it spills 20 bytes: 6.2 ms.  The diagonal tiles keep the plain path.)

  // (named registers and unconditional loads: an array filled under `if (t + 1 < n64)` is kept in scratch by hipcc)
  uint4 kreg0, kreg1, kreg2, kreg3, vreg0, vreg1, vreg2, vreg3;
#define P2_LD(base, t, rr) (*(const uint4*)((base) + (long)min((t) * 64 + (rr) * 16 + skey, T - 1) * kv_row + 8 * sc))
#define P2_LOAD_K(t) do { const int t_ = min((t), n64 - 1); kreg0 = P2_LD(kbase, t_, 0); kreg1 = P2_LD(kbase, t_, 1); kreg2 = P2_LD(kbase, t_, 2); kreg3 = P2_LD(kbase, t_, 3); } while (0);
#define P2_LOAD_V(t) do { const int t_ = min((t), n64 - 1); vreg0 = P2_LD(vbase, t_, 0); vreg1 = P2_LD(vbase, t_, 1); vreg2 = P2_LD(vbase, t_, 2); vreg3 = P2_LD(vbase, t_, 3); } while (0);
#define P2_STORE_K() do { *(uint4*)(k_lds + k_wr) = kreg0; *(uint4*)(k_lds + 4096 + k_wr) = kreg1; *(uint4*)(k_lds + 8192 + k_wr) = kreg2; *(uint4*)(k_lds + 12288 + k_wr) = kreg3; } while (0);
#define P2_STORE_V() do { *(uint4*)(v_lds + v_wr) = vreg0; *(uint4*)(v_lds + 4096 + v_wr) = vreg1; *(uint4*)(v_lds + 8192 + v_wr) = vreg2; *(uint4*)(v_lds + 12288 + v_wr) = vreg3; } while (0);

  // ---- sweep 1: row max and sum of exp over all keys <= query ----
  float m = -INFINITY, l = 0.f;
  P2_LOAD_K(0)
  for (int t = 0; t < n64; ++t) {
    __syncthreads();
    P2_STORE_K()
    __syncthreads();
    P2_LOAD_K(t + 1)      // (clamped: the last iteration re-reads its own tile)
    if (P2_TILE_INSIDE(t)) {
      f32x16 s0, s1;
      P2_QK(0, s0);
      P2_QK(1, s1);
      P2_ROUND_MASK(2 * t, s0, true);
      P2_STATS(s0);
      P2_ROUND_MASK(2 * t + 1, s1, true);
      P2_STATS(s1);
      continue;
    }
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const int kt = 2 * t + sb;
      if (kt < n32 && kt * 32 <= q_wave + 31) {  // wave-uniform: block exists and is not entirely above the diagonal
        f32x16 s;
        P2_SCORES(kt, sb, s);
        P2_STATS(s);
      }
    }
  }

  // ---- sweep 2: P = bf16(exp(s - m) / l), O^T += V^T . P^T ----
  // e / l for the 16 e of a block share l: the reciprocal and its Newton step are hoisted, the per-element part is
  // the quotient + two residual corrections of the IEEE sequence hipcc emits for `/` (v_div_scale is the identity here:
  // 1 <= l <= T, 0 <= e <= 1), i.e. the same correctly rounded quotient in 5 FMAs instead of ~11 instructions.
  // (lia_attn_prefill_kernel keeps the plain `/`; tools/attn_ab.py compares the two kernels bit for bit.)
  const float rl0 = __builtin_amdgcn_rcpf(l);
  const float rl = __builtin_fmaf(__builtin_fmaf(-l, rl0, 1.0f), rl0, rl0);
#define P2_DIV(e_, out_)                                                                                              \
  do {                                                                                                                \
    const float n_ = (e_);                                                                                            \
    float q_ = n_ * rl;                                                                                               \
    q_ = __builtin_fmaf(__builtin_fmaf(-l, q_, n_), rl, q_);                                                          \
    out_ = __builtin_fmaf(__builtin_fmaf(-l, q_, n_), rl, q_);                                                        \
  } while (0)
  f32x16 oacc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) oacc[d] = f32x16{0};
  P2_LOAD_K(0)
  P2_LOAD_V(0)
  for (int t = 0; t < n64; ++t) {
    __syncthreads();
    P2_STORE_K()
    P2_STORE_V()
    __syncthreads();
    P2_LOAD_K(t + 1)
    P2_LOAD_V(t + 1)
#define P2_PROBS(s, pk)                                                                                               \
  do {                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                   \
      float p0, p1;                                                                                                   \
      P2_DIV(__expf(s[2 * i] - m), p0);                                                                               \
      P2_DIV(__expf(s[2 * i + 1] - m), p1);                                                                           \
      pk[i] = pack_bf16x2(p0, p1);                                                                                    \
    }                                                                                                                 \
  } while (0)
    // keys 32 sb + 16 ks + 4 h + (0..3) -> elements 0..3, the same + 8 -> elements 4..7 (the k order of an
    // accumulator tile used as an operand)
#define P2_PV(sb, pk)                                                                                                 \
  do {                                                                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                \
      bf16x8 pf = __builtin_bit_cast(bf16x8, uint4{pk[4 * ks], pk[4 * ks + 1], pk[4 * ks + 2], pk[4 * ks + 3]});      \
      _Pragma("unroll") for (int d = 0; d < 4; ++d) {                                                                 \
        const char* vb = v_lds + (32 * (sb) + 16 * ks) * 256;                                                         \
        lia_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) lia_v4s*)(vb + v_rd[0][d])); \
        lia_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) lia_v4s*)(vb + v_rd[1][d])); \
        bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));              \
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[d], 0, 0, 0);                                  \
      }                                                                                                               \
    }                                                                                                                 \
  } while (0)
    if (P2_TILE_INSIDE(t)) {
      f32x16 s0, s1;
      uint32_t pk0[8], pk1[8];
      P2_QK(0, s0);
      P2_QK(1, s1);
      P2_ROUND_MASK(2 * t, s0, true);
      P2_PROBS(s0, pk0);
      P2_PV(0, pk0);
      P2_ROUND_MASK(2 * t + 1, s1, true);
      P2_PROBS(s1, pk1);
      P2_PV(1, pk1);
      continue;
    }
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const int kt = 2 * t + sb;
      if (kt < n32 && kt * 32 <= q_wave + 31) {
        f32x16 s;
        uint32_t pk[8];
        P2_SCORES(kt, sb, s);
        P2_PROBS(s, pk);
        P2_PV(sb, pk);
      }
    }
  }
#undef P2_PROBS
#undef P2_PV
#undef P2_QK
#undef P2_ROUND_MASK
#undef P2_STATS
#undef P2_TILE_INSIDE
#undef P2_SCORES
#undef P2_DIV
#undef P2_LOAD_K
#undef P2_LOAD_V
#undef P2_LD
#undef P2_STORE_K
#undef P2_STORE_V

  if (my_q < T) {
    bf16_t* op = out + ((long)b * T + my_q) * ldo + (long)hh * D;
#pragma unroll
    for (int d = 0; d < 4; ++d)
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
      hipLaunchKernelGGL(lia_attn_prefill128_kernel, grid, dim3(256), 0, st, q, ldq, kc, vc, out, ldo, T, heads, kv_heads, (long)Bc * hd, hd, b0, scaling, post_scale);
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
