// Epilogue arithmetic shared by the GEMM kernels (lia_gemm.hip), their split-K combines and the persistent decode chain
// (lia_chain.hip): one definition, so every route applies the reference's rounding points with the same instructions.
#pragma once
#include "lia_common.h"

// ---------------------------------------------------------------------------------------------
// shared epilogue: 4 consecutive columns n..n+3 of row m
// ---------------------------------------------------------------------------------------------
// the four finished values (every reference rounding point applied: they are bf16-representable)
__device__ __forceinline__ f32x4 epilogue_quad(const f32x4& v, int m, int n, const LiaEpilogue& ep) {
  float b[4] = {0.f, 0.f, 0.f, 0.f}, r[4] = {0.f, 0.f, 0.f, 0.f};
  const bool hb = ep.bias != nullptr, hr = ep.residual != nullptr;
  if (hb) {
    uint2 bb = *(const uint2*)(ep.bias + n);
    b[0] = bf2f(bb.x & 0xffff); b[1] = bf2f(bb.x >> 16); b[2] = bf2f(bb.y & 0xffff); b[3] = bf2f(bb.y >> 16);
  }
  if (hr) {
    uint2 rr = *(const uint2*)(ep.residual + (long)m * ep.ldr + n);
    r[0] = bf2f(rr.x & 0xffff); r[1] = bf2f(rr.x >> 16); r[2] = bf2f(rr.y & 0xffff); r[3] = bf2f(rr.y >> 16);
  }
  return f32x4{lia_epilogue_apply(v[0], b[0], hb, ep.relu, r[0], hr), lia_epilogue_apply(v[1], b[1], hb, ep.relu, r[1], hr),
               lia_epilogue_apply(v[2], b[2], hb, ep.relu, r[2], hr), lia_epilogue_apply(v[3], b[3], hb, ep.relu, r[3], hr)};
}

// the same with the bias / residual values already in registers (packed bf16 x 4 each): for callers that request them together
// with everything else they load, ahead of the arithmetic
__device__ __forceinline__ f32x4 epilogue_quad_pre(const f32x4& v, const uint2& bb, const uint2& rr, bool hb, int relu, bool hr) {
  float b[4] = {0.f, 0.f, 0.f, 0.f}, r[4] = {0.f, 0.f, 0.f, 0.f};
  if (hb) { b[0] = bf2f(bb.x & 0xffff); b[1] = bf2f(bb.x >> 16); b[2] = bf2f(bb.y & 0xffff); b[3] = bf2f(bb.y >> 16); }
  if (hr) { r[0] = bf2f(rr.x & 0xffff); r[1] = bf2f(rr.x >> 16); r[2] = bf2f(rr.y & 0xffff); r[3] = bf2f(rr.y >> 16); }
  return f32x4{lia_epilogue_apply(v[0], b[0], hb, relu, r[0], hr), lia_epilogue_apply(v[1], b[1], hb, relu, r[1], hr),
               lia_epilogue_apply(v[2], b[2], hb, relu, r[2], hr), lia_epilogue_apply(v[3], b[3], hb, relu, r[3], hr)};
}

__device__ __forceinline__ void store_quad(const f32x4& v, int m, int n, const LiaEpilogue& ep, const LiaOutMap& om) {
  float b[4] = {0.f, 0.f, 0.f, 0.f}, r[4] = {0.f, 0.f, 0.f, 0.f};
  const bool hb = ep.bias != nullptr, hr = ep.residual != nullptr;
  if (hb) {
    uint2 bb = *(const uint2*)(ep.bias + n);
    b[0] = bf2f(bb.x & 0xffff); b[1] = bf2f(bb.x >> 16); b[2] = bf2f(bb.y & 0xffff); b[3] = bf2f(bb.y >> 16);
  }
  if (hr) {
    uint2 rr = *(const uint2*)(ep.residual + (long)m * ep.ldr + n);
    r[0] = bf2f(rr.x & 0xffff); r[1] = bf2f(rr.x >> 16); r[2] = bf2f(rr.y & 0xffff); r[3] = bf2f(rr.y >> 16);
  }
  float t0 = lia_epilogue_apply(v[0], b[0], hb, ep.relu, r[0], hr);
  float t1 = lia_epilogue_apply(v[1], b[1], hb, ep.relu, r[1], hr);
  float t2 = lia_epilogue_apply(v[2], b[2], hb, ep.relu, r[2], hr);
  float t3 = lia_epilogue_apply(v[3], b[3], hb, ep.relu, r[3], hr);
  uint2 o;
  o.x = pack_bf16x2(t0, t1);
  o.y = pack_bf16x2(t2, t3);
  *(uint2*)lia_out_ptr(om, m, n) = o;
}

__device__ __forceinline__ f32x4 splitk_sum(const float* __restrict__ partial, int S, int M, int N, int m, int n) {
  // slice 0, 1, ...: the order of the plain combine; four slices' loads in flight at a time
  f32x4 a = *(const f32x4*)(partial + (long)m * N + n);
  for (int s0 = 1; s0 < S; s0 += 4) {
    f32x4 t[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) t[g] = *(const f32x4*)(partial + ((long)min(s0 + g, S - 1) * M + m) * N + n);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (s0 + g < S) a += t[g];
  }
  return a;
}
