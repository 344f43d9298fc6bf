// Row-wise / gather kernels of the LIA hot path (HBM-bound, 16-byte vector accesses everywhere):
// LayerNorm, token+position embedding, last-position gather, greedy argmax.
#include <cstdlib>
#include "lia_common.h"

// F.layer_norm on a bf16 tensor (decoder.py:107-119; final LN lia/modeling_opt.py:1563): statistics and
// affine in fp32, ONE rounding at the output.  One wave per row; the row (<= 24 KB) is read three
// times, the 2nd/3rd time from L1/L2.
__global__ __launch_bounds__(256) void lia_layernorm_kernel(const bf16_t* __restrict__ x, long ldx,
                                                             const bf16_t* __restrict__ g, const bf16_t* __restrict__ b,
                                                             bf16_t* __restrict__ y, long ldy, long rows, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * ldx;
  const int nv = H >> 3;
  float s = 0.f;
  for (int i = lane; i < nv; i += 64) {
    uint4 v = *(const uint4*)(xr + 8 * i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) s += bf2f(w[j] & 0xffff) + bf2f(w[j] >> 16);
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
  for (int i = lane; i < nv; i += 64) {
    uint4 v = *(const uint4*)(xr + 8 * i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = bf2f(w[j] & 0xffff) - mean, c = bf2f(w[j] >> 16) - mean;
      q += a * a + c * c;
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
  bf16_t* yr = y + row * ldy;
  for (int i = lane; i < nv; i += 64) {
    uint4 v = *(const uint4*)(xr + 8 * i);
    uint4 gv = *(const uint4*)(g + 8 * i);
    uint4 bv = *(const uint4*)(b + 8 * i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w}, gw[4] = {gv.x, gv.y, gv.z, gv.w}, bw[4] = {bv.x, bv.y, bv.z, bv.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float lo = (bf2f(w[j] & 0xffff) - mean) * rstd * bf2f(gw[j] & 0xffff) + bf2f(bw[j] & 0xffff);
      float hi = (bf2f(w[j] >> 16) - mean) * rstd * bf2f(gw[j] >> 16) + bf2f(bw[j] >> 16);
      o[j] = pack_bf16x2(lo, hi);
    }
    *(uint4*)(yr + 8 * i) = uint4{o[0], o[1], o[2], o[3]};
  }
}

// The same arithmetic (same per-lane summation order: bit-identical results) with the row held in registers: every lane
// requests its NV 16-byte pieces at once and the row crosses the memory system ONCE.  The three-pass kernel above issues
// 3 x NV dependent round trips per lane: 13 us for 64 decode rows (latency), 2.7 TB/s at 16384 prefill rows.
template <int NV>
__global__ __launch_bounds__(256) void lia_layernorm_reg_kernel(const bf16_t* __restrict__ x, long ldx,
                                                                 const bf16_t* __restrict__ g, const bf16_t* __restrict__ b,
                                                                 bf16_t* __restrict__ y, long ldy, long rows, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * ldx;
  const int nv = H >> 3;
  uint4 v[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = lane + 64 * k;
    v[k] = i < nv ? *(const uint4*)(xr + 8 * i) : uint4{0u, 0u, 0u, 0u};
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (lane + 64 * k < nv) {
      const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) s += bf2f(w[j] & 0xffff) + bf2f(w[j] >> 16);
    }
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (lane + 64 * k < nv) {
      const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = bf2f(w[j] & 0xffff) - mean, c = bf2f(w[j] >> 16) - mean;
        q += a * a + c * c;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
  bf16_t* yr = y + row * ldy;
  // gain and bias of piece k + 1 are requested (clamped index, no branch around the loads) before piece k is computed and stored:
  // loaded under `if (i < nv)` each pair was waited for on the spot -- NV dependent round trips per wave
  uint4 gv = *(const uint4*)(g + 8 * min(lane, nv - 1)), bv = *(const uint4*)(b + 8 * min(lane, nv - 1));
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = lane + 64 * k;
    const uint4 gc = gv, bc = bv;
    if (k + 1 < NV) {
      gv = *(const uint4*)(g + 8 * min(i + 64, nv - 1));
      bv = *(const uint4*)(b + 8 * min(i + 64, nv - 1));
    }
    if (i < nv) {
      const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w}, gw[4] = {gc.x, gc.y, gc.z, gc.w}, bw[4] = {bc.x, bc.y, bc.z, bc.w};
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float lo = (bf2f(w[j] & 0xffff) - mean) * rstd * bf2f(gw[j] & 0xffff) + bf2f(bw[j] & 0xffff);
        float hi = (bf2f(w[j] >> 16) - mean) * rstd * bf2f(gw[j] >> 16) + bf2f(bw[j] >> 16);
        o[j] = pack_bf16x2(lo, hi);
      }
      *(uint4*)(yr + 8 * i) = uint4{o[0], o[1], o[2], o[3]};
    }
  }
}

// Decode-sized inputs (a few hundred rows): one 1024-thread workgroup per row instead of one wave -- every load of the row in
// flight at once and four times the CUs (64 rows of OPT-30B: 10.6 -> 4.4 us, the kernel is pure latency), and the same
// device function (row_layernorm_block) the fused split-K combine of lia_gemm.hip runs, so both routes give the same bits.
template <int NV>
__global__ __launch_bounds__(LIA_ROW_THREADS) void lia_layernorm_row_kernel(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ g,
                                                                 const bf16_t* __restrict__ b, bf16_t* __restrict__ y, long ldy,
                                                                 int H, float eps) {
  __shared__ float red[2 * LIA_ROW_WAVES];
  const long row = blockIdx.x;
  const bf16_t* xr = x + row * ldx;
  const int nv = H >> 3;
  uint4 v[NV], gv[NV], bv[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = threadIdx.x + LIA_ROW_THREADS * k;
    const bool in = i < nv;
    v[k] = in ? *(const uint4*)(xr + 8 * i) : uint4{0u, 0u, 0u, 0u};
    gv[k] = in ? *(const uint4*)(g + 8 * i) : uint4{0u, 0u, 0u, 0u};
    bv[k] = in ? *(const uint4*)(b + 8 * i) : uint4{0u, 0u, 0u, 0u};
  }
  row_layernorm_block<NV>(v, gv, bv, nv, H, eps, y + row * ldy, red);
}

// up to this many rows the workgroup-per-row kernel is used (the cut-over measured with tools/norm_bench.py, r03)
static const long g_row_norm_max_rows = 1024L;

extern "C" void lia_layernorm_launch(const bf16_t* x, long ldx, const bf16_t* g, const bf16_t* b, bf16_t* y, long ldy,
                                     long rows, int H, float eps, hipStream_t st) {
  if (rows <= 0) return;
  if (rows <= g_row_norm_max_rows && (H & 7) == 0 && (H >> 3) <= LIA_ROW_THREADS * 2) {
    const dim3 grid((unsigned)rows), block(LIA_ROW_THREADS);
    if ((H >> 3) <= LIA_ROW_THREADS) hipLaunchKernelGGL(lia_layernorm_row_kernel<1>, grid, block, 0, st, x, ldx, g, b, y, ldy, H, eps);
    else hipLaunchKernelGGL(lia_layernorm_row_kernel<2>, grid, block, 0, st, x, ldx, g, b, y, ldy, H, eps);
    return;
  }
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const int nvl = ((H >> 3) + 63) / 64;            // 16-byte pieces per lane
  if (nvl <= 4) hipLaunchKernelGGL(lia_layernorm_reg_kernel<4>, grid, block, 0, st, x, ldx, g, b, y, ldy, rows, H, eps);          // H <= 2048
  else if (nvl <= 8) hipLaunchKernelGGL(lia_layernorm_reg_kernel<8>, grid, block, 0, st, x, ldx, g, b, y, ldy, rows, H, eps);     // H <= 4096
  else if (nvl <= 14) hipLaunchKernelGGL(lia_layernorm_reg_kernel<14>, grid, block, 0, st, x, ldx, g, b, y, ldy, rows, H, eps);   // H <= 7168 (OPT-30B)
  else if (nvl <= 24) hipLaunchKernelGGL(lia_layernorm_reg_kernel<24>, grid, block, 0, st, x, ldx, g, b, y, ldy, rows, H, eps);   // H <= 12288 (OPT-175B)
  else hipLaunchKernelGGL(lia_layernorm_kernel, grid, block, 0, st, x, ldx, g, b, y, ldy, rows, H, eps);
}

// hidden = embed_tokens[ids] + embed_positions[past_len + t + 2], one bf16 add
// (lia/modeling_opt.py:1108 token embedding, :357-378 learned positions with offset 2 and an all-ones
// mask, :1142 the sum).  One workgroup per token row.
__global__ __launch_bounds__(256) void lia_embed_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ tok,
                                                         const bf16_t* __restrict__ pos, bf16_t* __restrict__ y, int T,
                                                         int past_len, int H) {
  const long row = blockIdx.x;
  const int t = (int)(row % T);
  const bf16_t* te = tok + ids[row] * (long)H;
  const bf16_t* pe = pos + (long)(past_len + t + 2) * H;
  bf16_t* yo = y + row * (long)H;
  for (int i = threadIdx.x; i < (H >> 3); i += 256) {
    uint4 a = *(const uint4*)(te + 8 * i), p = *(const uint4*)(pe + 8 * i);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, pw[4] = {p.x, p.y, p.z, p.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack_bf16x2(bf2f(aw[j] & 0xffff) + bf2f(pw[j] & 0xffff), bf2f(aw[j] >> 16) + bf2f(pw[j] >> 16));
    *(uint4*)(yo + 8 * i) = uint4{o[0], o[1], o[2], o[3]};
  }
}

extern "C" void lia_embed_launch(const int64_t* ids, const bf16_t* tok, const bf16_t* pos, bf16_t* y, int B, int T,
                                 int past_len, int H, hipStream_t st) {
  if (B * T <= 0) return;
  hipLaunchKernelGGL(lia_embed_kernel, dim3(B * T), dim3(256), 0, st, ids, tok, pos, y, T, past_len, H);
}

// Greedy argmax over bf16 logits [B, vocab], first maximal index on ties (greedy_search.py:367,395).
// `suppress` (or -1): a token whose score counts as -inf -- what HF's MinNewTokensLengthLogitsProcessor does
// to EOS while min_new_tokens is not reached (run_generation.py:173,179-182 sets min_new_tokens = max_new_tokens).
__global__ __launch_bounds__(LIA_ROW_THREADS) void lia_argmax_kernel(const bf16_t* __restrict__ logits, int64_t* __restrict__ out,
                                                          int vocab, int suppress) {
  __shared__ float sv[LIA_ROW_WAVES];
  __shared__ int si[LIA_ROW_WAVES];
  const bf16_t* row = logits + (long)blockIdx.x * vocab;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if ((vocab & 7) == 0) {
    // rows are 16-byte aligned: 8 logits per load, indices ascending inside a lane so ties keep the first one
    for (int i = threadIdx.x * 8; i < vocab; i += LIA_ROW_THREADS * 8) {
      const uint4 v = *(const uint4*)(row + i);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = i + e;
        const float f = idx == suppress ? -INFINITY : bf2f((w[e >> 1] >> ((e & 1) * 16)) & 0xffff);
        if (f > best || (f == best && idx < bi)) { best = f; bi = idx; }
      }
    }
  } else {
    for (int i = threadIdx.x; i < vocab; i += LIA_ROW_THREADS) {
      float f = i == suppress ? -INFINITY : bf2f(row[i]);
      if (f > best || (f == best && i < bi)) { best = f; bi = i; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ob = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < LIA_ROW_WAVES; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    out[blockIdx.x] = bi == 0x7fffffff ? 0 : bi;
  }
}

extern "C" void lia_argmax_launch(const bf16_t* logits, int64_t* out, int B, int vocab, int suppress, hipStream_t st) {
  if (B <= 0) return;
  hipLaunchKernelGGL(lia_argmax_kernel, dim3(B), dim3(LIA_ROW_THREADS), 0, st, logits, out, vocab, suppress);   // one 1024-thread workgroup per row: a row of 128 K logits is latency (45 -> 15 us at B 128)
}

// Small device<->pinned-host transfers of the policy-2 round trip (q|k|v out, attention result in) done by a
// KERNEL over the mapped host pointer instead of hipMemcpyAsync: the copy engines are saturated by the
// 1.2 GB/layer weight stream, and a 1 MB SDMA copy queued behind it would stall the layer for ~20 ms
// (measured: copy-engine busy 85 % instead of 99.9 %).  16 B per lane, grid-stride.
__global__ __launch_bounds__(256) void lia_blit_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n16; i += stride) dst[i] = src[i];
}

extern "C" void lia_blit_launch(void* dst, const void* src, size_t bytes, hipStream_t st) {
  if (bytes == 0) return;
  size_t n16 = bytes / 16;   // callers pass multiples of 16 (rows of H bf16 values, H % 8 == 0)
  unsigned blocks = (unsigned)((n16 + 255) / 256);
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(lia_blit_kernel, dim3(blocks), dim3(256), 0, st, (uint4*)dst, (const uint4*)src, n16);
}
