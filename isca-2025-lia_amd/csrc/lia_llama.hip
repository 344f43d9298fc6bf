// Llama-family element-wise kernels (BASELINE.json config 4; a build-defined extension -- the reference has no LIA
// Llama path, decoder.py:121-169).  Rounding points follow HF transformers' eager bf16 Llama (modeling_llama.py).
#include "lia_common.h"

// LlamaRMSNorm.forward: y = bf16( w * bf16( x * rsqrt(mean(x^2) + eps) ) ); one wave per row, 16-byte accesses.
__global__ __launch_bounds__(256) void lia_rmsnorm_kernel(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ w,
                                                           bf16_t* __restrict__ y, long ldy, long rows, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * ldx;
  const int nv = H >> 3;
  float ss = 0.f;
  for (int i = lane; i < nv; i += 64) {
    uint4 v = *(const uint4*)(xr + 8 * i);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { float a = bf2f(u[j] & 0xffff), c = bf2f(u[j] >> 16); ss += a * a + c * c; }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)H + eps);
  bf16_t* yr = y + row * ldy;
  for (int i = lane; i < nv; i += 64) {
    uint4 v = *(const uint4*)(xr + 8 * i);
    uint4 g = *(const uint4*)(w + 8 * i);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w}, gw[4] = {g.x, g.y, g.z, g.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack_bf16x2(bf2f(gw[j] & 0xffff) * rbf(bf2f(u[j] & 0xffff) * rstd), bf2f(gw[j] >> 16) * rbf(bf2f(u[j] >> 16) * rstd));
    *(uint4*)(yr + 8 * i) = uint4{o[0], o[1], o[2], o[3]};
  }
}

// the same arithmetic with the row in registers (one pass over memory, same per-lane summation order: bit-identical);
// see lia_layernorm_reg_kernel
template <int NV>
__global__ __launch_bounds__(256) void lia_rmsnorm_reg_kernel(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ w,
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
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (lane + 64 * k < nv) {
      const uint32_t u[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { float a = bf2f(u[j] & 0xffff), c = bf2f(u[j] >> 16); ss += a * a + c * c; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)H + eps);
  bf16_t* yr = y + row * ldy;
  // the gain of piece k + 1 is requested (clamped index) before piece k is computed: see lia_layernorm_reg_kernel
  uint4 gn = *(const uint4*)(w + 8 * min(lane, nv - 1));
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = lane + 64 * k;
    const uint4 g = gn;
    if (k + 1 < NV) gn = *(const uint4*)(w + 8 * min(i + 64, nv - 1));
    if (i < nv) {
      const uint32_t u[4] = {v[k].x, v[k].y, v[k].z, v[k].w}, gw[4] = {g.x, g.y, g.z, g.w};
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = pack_bf16x2(bf2f(gw[j] & 0xffff) * rbf(bf2f(u[j] & 0xffff) * rstd), bf2f(gw[j] >> 16) * rbf(bf2f(u[j] >> 16) * rstd));
      *(uint4*)(yr + 8 * i) = uint4{o[0], o[1], o[2], o[3]};
    }
  }
}

// decode-sized inputs: one workgroup per row (see lia_layernorm_row_kernel), the arithmetic of row_rmsnorm_block
template <int NV>
__global__ __launch_bounds__(LIA_ROW_THREADS) void lia_rmsnorm_row_kernel(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ w,
                                                               bf16_t* __restrict__ y, long ldy, int H, float eps) {
  __shared__ float red[LIA_ROW_WAVES];
  const long row = blockIdx.x;
  const bf16_t* xr = x + row * ldx;
  const int nv = H >> 3;
  uint4 v[NV], gv[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = threadIdx.x + LIA_ROW_THREADS * k;
    const bool in = i < nv;
    v[k] = in ? *(const uint4*)(xr + 8 * i) : uint4{0u, 0u, 0u, 0u};
    gv[k] = in ? *(const uint4*)(w + 8 * i) : uint4{0u, 0u, 0u, 0u};
  }
  row_rmsnorm_block<NV>(v, gv, nv, H, eps, y + row * ldy, red);
}

extern "C" void lia_rmsnorm_launch(const bf16_t* x, long ldx, const bf16_t* w, bf16_t* y, long ldy, long rows, int H, float eps,
                                   hipStream_t st) {
  if (rows > 0 && rows <= 1024 && (H & 7) == 0 && (H >> 3) <= LIA_ROW_THREADS * 2) {
    const dim3 grid((unsigned)rows), block(LIA_ROW_THREADS);
    if ((H >> 3) <= LIA_ROW_THREADS) hipLaunchKernelGGL(lia_rmsnorm_row_kernel<1>, grid, block, 0, st, x, ldx, w, y, ldy, H, eps);
    else hipLaunchKernelGGL(lia_rmsnorm_row_kernel<2>, grid, block, 0, st, x, ldx, w, y, ldy, H, eps);
    return;
  }
  if (rows <= 0) return;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const int nvl = ((H >> 3) + 63) / 64;
  if (nvl <= 4) hipLaunchKernelGGL(lia_rmsnorm_reg_kernel<4>, grid, block, 0, st, x, ldx, w, y, ldy, rows, H, eps);
  else if (nvl <= 8) hipLaunchKernelGGL(lia_rmsnorm_reg_kernel<8>, grid, block, 0, st, x, ldx, w, y, ldy, rows, H, eps);       // H <= 4096 (Llama-3-8B)
  else if (nvl <= 16) hipLaunchKernelGGL(lia_rmsnorm_reg_kernel<16>, grid, block, 0, st, x, ldx, w, y, ldy, rows, H, eps);     // H <= 8192 (70B)
  else hipLaunchKernelGGL(lia_rmsnorm_kernel, grid, block, 0, st, x, ldx, w, y, ldy, rows, H, eps);
}

// apply_rotary_pos_emb in place: out = bf16( bf16(x*cos) + bf16(rotate_half(x)*sin) ), cos/sin tables [max_pos][d] bf16.
// Rows are token rows of `heads` heads each; position of row r = pos0 + (pos_mod ? r % pos_mod : r / pos_div)
// (q buffer: r = b*T + t -> pos_mod = T; cache slab: r = t*Bc + b -> pos_div = Bc).  One thread per (row, head, pair).
__global__ __launch_bounds__(256) void lia_rope_kernel(bf16_t* __restrict__ x, long row_stride, const bf16_t* __restrict__ cosb,
                                                        const bf16_t* __restrict__ sinb, long rows, int heads, int d, int pos0,
                                                        int pos_mod, int pos_div) {
  const int half = d >> 1;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = rows * heads * half;
  if (idx >= total) return;
  const int i = (int)(idx % half);
  const long rh = idx / half;
  const int h = (int)(rh % heads);
  const long r = rh / heads;
  const int pos = pos0 + (pos_mod ? (int)(r % pos_mod) : (int)(r / pos_div));
  bf16_t* p = x + r * row_stride + (long)h * d;
  const float a = bf2f(p[i]), b = bf2f(p[i + half]);
  const float c0 = bf2f(cosb[(long)pos * d + i]), c1 = bf2f(cosb[(long)pos * d + i + half]);
  const float s0 = bf2f(sinb[(long)pos * d + i]), s1 = bf2f(sinb[(long)pos * d + i + half]);
  p[i] = f2bf(rbf(a * c0) + rbf(-b * s0));
  p[i + half] = f2bf(rbf(b * c1) + rbf(a * s1));
}

// the same arithmetic, 16 bytes per access: one thread rotates 8 consecutive pairs (i .. i+7, i+half .. i+half+7) of one
// (row, head); the cos / sin table rows are read 16 bytes at a time too.  The scalar kernel above moves 2 bytes per access and
// took 1.1 ms for the q rows of a B 128 x T 1024 prefill (1 GB in, 1 GB out).  head_dim must be a multiple of 16.
__global__ __launch_bounds__(256) void lia_rope_vec_kernel(bf16_t* __restrict__ x, long row_stride, const bf16_t* __restrict__ cosb,
                                                            const bf16_t* __restrict__ sinb, long rows, int heads, int d, int pos0,
                                                            int pos_mod, int pos_div) {
  const int half = d >> 1, groups = half >> 3;               // groups of 8 pairs per head
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = rows * heads * groups;
  if (idx >= total) return;
  const int gi = (int)(idx % groups);
  const long rh = idx / groups;
  const int h = (int)(rh % heads);
  const long r = rh / heads;
  const int pos = pos0 + (pos_mod ? (int)(r % pos_mod) : (int)(r / pos_div));
  bf16_t* p = x + r * row_stride + (long)h * d + 8 * gi;
  const bf16_t* cp = cosb + (long)pos * d + 8 * gi;
  const bf16_t* sp = sinb + (long)pos * d + 8 * gi;
  const uint4 av = *(const uint4*)p, bv = *(const uint4*)(p + half);
  const uint4 c0v = *(const uint4*)cp, c1v = *(const uint4*)(cp + half), s0v = *(const uint4*)sp, s1v = *(const uint4*)(sp + half);
  const uint32_t aw[4] = {av.x, av.y, av.z, av.w}, bw[4] = {bv.x, bv.y, bv.z, bv.w};
  const uint32_t c0[4] = {c0v.x, c0v.y, c0v.z, c0v.w}, c1[4] = {c1v.x, c1v.y, c1v.z, c1v.w};
  const uint32_t s0[4] = {s0v.x, s0v.y, s0v.z, s0v.w}, s1[4] = {s1v.x, s1v.y, s1v.z, s1v.w};
  uint32_t oa[4], ob[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) lia_rope_pair(aw[j], bw[j], c0[j], c1[j], s0[j], s1[j], oa[j], ob[j]);
  *(uint4*)p = uint4{oa[0], oa[1], oa[2], oa[3]};
  *(uint4*)(p + half) = uint4{ob[0], ob[1], ob[2], ob[3]};
}

extern "C" void lia_rope_launch(bf16_t* x, long row_stride, const bf16_t* cosb, const bf16_t* sinb, long rows, int heads, int d,
                                int pos0, int pos_mod, int pos_div, hipStream_t st) {
  if (rows <= 0) return;
  if ((d % 16) == 0 && (row_stride % 8) == 0) {
    const long total = rows * heads * (d / 16);
    hipLaunchKernelGGL(lia_rope_vec_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, row_stride, cosb, sinb, rows,
                       heads, d, pos0, pos_mod, pos_div);
    return;
  }
  const long total = rows * heads * (d / 2);
  hipLaunchKernelGGL(lia_rope_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, row_stride, cosb, sinb, rows, heads,
                     d, pos0, pos_mod, pos_div);
}

// LlamaMLP: act_fn(gate) * up with gate|up side by side in one [M, 2F] buffer: m = bf16( bf16(silu(g)) * u ).
// gu_block = 0: columns [gate (F) | up (F)]; LIA_GU_BLOCK: blocks of 32 gate | 32 up columns (the interleaved weight layout).
__global__ __launch_bounds__(256) void lia_silu_mul_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ out, long M, int F, int gu_block) {
  const long n8 = M * (F >> 3);
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  for (; i < n8; i += stride) {
    const long m = i / (F >> 3);
    const int c = (int)(i - m * (F >> 3)) * 8;
    const int ng = gu_block ? (c / LIA_GU_BLOCK) * (2 * LIA_GU_BLOCK) + (c % LIA_GU_BLOCK) : c;
    const int nu = gu_block ? ng + LIA_GU_BLOCK : F + c;
    uint4 g = *(const uint4*)(gu + m * 2 * (long)F + ng);
    uint4 u = *(const uint4*)(gu + m * 2 * (long)F + nu);
    const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, uw[4] = {u.x, u.y, u.z, u.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = lia_silu_mul_pair(gw[j], uw[j]);
    *(uint4*)(out + m * (long)F + c) = uint4{o[0], o[1], o[2], o[3]};
  }
}

extern "C" void lia_silu_mul_launch(const bf16_t* gu, bf16_t* out, long M, int F, int gu_block, hipStream_t st) {
  const long n8 = M * (F >> 3);
  if (n8 <= 0) return;
  long blocks = (n8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(lia_silu_mul_kernel, dim3((unsigned)blocks), dim3(256), 0, st, gu, out, M, F, gu_block);
}

// token embedding only (Llama has no learned positions): y[row] = embed[ids[row]]
__global__ __launch_bounds__(256) void lia_embed_tokens_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ tok,
                                                                bf16_t* __restrict__ y, int H) {
  const long row = blockIdx.x;
  const bf16_t* te = tok + ids[row] * (long)H;
  bf16_t* yo = y + row * (long)H;
  for (int i = threadIdx.x; i < (H >> 3); i += 256) *(uint4*)(yo + 8 * i) = *(const uint4*)(te + 8 * i);
}

extern "C" void lia_embed_tokens_launch(const int64_t* ids, const bf16_t* tok, bf16_t* y, long rows, int H, hipStream_t st) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(lia_embed_tokens_kernel, dim3((unsigned)rows), dim3(256), 0, st, ids, tok, y, H);
}
