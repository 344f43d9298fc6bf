// Persistent decode chain for HBM-resident layers (gfx950): ONE launch runs a program of steps
//     GEMM (split-K partial slabs or a finished tile)  ->  grid barrier  ->  REDUCE (combine + bias / residual / norm / RoPE)  ->  ...
// with one 512-thread workgroup per CU that never gives its CU back.  It replaces, per decoder layer, the ~10 launches of the
// op sequence of decoder.py:172-335 / attentions.py:393-529 behind the attention (out-proj, LN2, fc1, fc2, the NEXT layer's LN1
// and q|k|v projection; Llama: o, norm, gate|up . SiLU, down, norm, q|k|v + RoPE) -- the per-op path (lia_gemm.hip + its combines)
// pays a pipeline fill, a slab round trip and a combine launch per GEMM: "6.7 TB/s asymptotic + 14 us fixed per launch".
//
// What the persistent form changes (MI355X_MICROARCH.md, persistent-kernel price list):
//   * the weight ring never drains: every CU streams its W tiles through a DW-stage LDS ring by LDS-DMA (global_load_lds, nt) and
//     keeps requesting the NEXT step's weights while it waits at a seam -- weights do not depend on activations
//     ("prefetch-credit"); the x operand has a ring of its own (DX stages) that restarts behind every seam;
//   * one work item per CU wherever the shape allows (bn rows x K / split columns, chosen per GEMM by lia_chain_plan_gemm): no
//     second wave of workgroups, no idle CUs at 0.6-0.9 waves;
//   * seams are an XCD-hierarchical grid barrier (counter per blockIdx % 8 group, leader to a top counter) with the agent-scope
//     hand-off of cdna_hip_programming.md Guideline 16 R1: every byte another workgroup reads is stored write-through (sc1) and
//     drained (s_waitcnt vmcnt(0)) before ONE lane arrives; after the wait ONE lane acquires (buffer_inv sc1), the workgroup
//     meets, then plain loads / LDS-DMA.  Results never depend on placement; blockIdx % 8 only decides who waits for whom;
//   * every spin is bounded (spin_limit polls): a barrier that times out sets the error word and the launch runs to its end.
//
// Status (r04): measured 4-9 % SLOWER per decode step than the per-op route it was meant to replace (16.85 vs 16.10 ms OPT-30B
// resident, 8.42 vs 7.72 ms Llama-3-8B, results/r04_ab_*): a seam -- ring drain + grid barrier + reduce step + cold ring -- costs
// what a kernel boundary + combine kernel costs, and the per-op kernels' tails overlap the next launch's head.  Opt-in
// (LIA_FUSED_DECODE=1 / lia_set_fused_decode(1)); kept as a second implementation of the same arithmetic.  LABNOTES.md r04.
//
// Arithmetic: the K loop is lia_gemm_skinny2_kernel's (same chunks in the same order into the same accumulators), the combines
// add the slabs slice 0, 1, ... and finish a value with lia_epilogue.h / lia_common.h's device functions -- so a chain launch
// and the per-op path with the same K slices (lia_gemm_launch's force_split) give the same bits (tests/test_gpu_chain.py).
#include <cstdio>
#include <cstring>
#include "lia_chain.h"
#include "lia_epilogue.h"

#define CH_GL_AS1(p) ((const __attribute__((address_space(1))) void*)(p))
#define CH_LDS_AS3(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ int ch_swz(int row) { return (row >> 1) & 7; }     // = tl_swz of lia_gemm.hip (LDS image of a 128-B row)

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate): waits until at most n of this wave's
// vector-memory operations are outstanding.  The steady-state counts of the three kernel geometries come first and are exact;
// anything else is rounded DOWN to the next value in the list -- fewer requests allowed out only waits longer, never too short.
#define CH_WV(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
__device__ __forceinline__ void ch_wait_vmcnt(int n) {
  if (n == 16) CH_WV(16);
  else if (n == 4) CH_WV(4);
  else if (n == 2) CH_WV(2);
  else if (n == 14) CH_WV(14);
  else if (n >= 24) CH_WV(24);
  else if (n >= 20) CH_WV(20);
  else if (n >= 16) CH_WV(16);
  else if (n >= 12) CH_WV(12);
  else if (n >= 8) CH_WV(8);
  else if (n >= 6) CH_WV(6);
  else if (n >= 4) CH_WV(4);
  else if (n == 3) CH_WV(3);
  else if (n == 1) CH_WV(1);
  else CH_WV(0);
}
#undef CH_WV

// Which requests may still be out when a chunk is consumed.  The eight waves split the loading: waves 0-3 request the W chunks,
// waves 4-7 the x chunks -- vmcnt is one in-order counter per wave, so a wave that requested both would have to see its W requests
// land in step with the (shallow) x ring and only ~2 chunks of weights would ever be in flight (measured: 17 GB/s per CU, the
// latency bound of 32 KB in flight); with its own counter the W ring keeps DW - 1 chunks out.  Per wave a FIFO with one byte per
// requested-and-unconsumed chunk of ITS ring (oldest in the low byte) = the wave's load counter (mod 256) right after that
// chunk's loads were issued; one 64-bit scalar (<= 8 stages, far fewer than 256 loads in flight): no arrays, no dynamic indexing.
struct ChFifo {
  unsigned long long w;
  int n;      // live entries
};
__device__ __forceinline__ void ch_fifo_push(ChFifo& f, int issued) {
  f.w |= (unsigned long long)(issued & 0xff) << (8 * f.n);
  ++f.n;
}
__device__ __forceinline__ int ch_fifo_behind(const ChFifo& f, int issued) { return (issued - (int)(f.w & 0xffull)) & 0xff; }   // loads issued after the front chunk's
__device__ __forceinline__ void ch_fifo_pop(ChFifo& f) { f.w >>= 8; --f.n; }

typedef __attribute__((address_space(1))) unsigned ch_gu32;
#define CH_RLX_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// a barrier gave up: mark this launch (device word, polled by the other waiters so that they give up at once) and the context
// (host-mapped word the library checks at its next synchronize: the step's results are garbage and the caller must know)
__device__ __forceinline__ void ch_fail(unsigned* err, unsigned* err_host, unsigned code) {
  __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (err_host) __hip_atomic_store(err_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------
// grid barrier number k (1, 2, ... within the launch).  sync: zeroed by the host before the launch, 128-byte lines:
//   [g] arrivals of group g = blockIdx % 8, [8 + g] generation released to group g, [16] groups arrived, [17] error word.
// (Measured alternatives: every workgroup publishing a word of its own and re-reading all of them -- no atomics, two memory hops
// on paper -- is SLOWER, 3.4-4.8 us against 2.6-3.0: 256 CUs polling the same eight lines is a hot spot.)
// Caller contract: every wave has waited (vmcnt(0)) for its write-through stores.  On return every thread may read, with plain
// loads, whatever any workgroup stored before its arrival.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ch_grid_barrier(unsigned* sync, int k, int spin_limit, unsigned* err_host) {
  __builtin_amdgcn_s_barrier();
  if (threadIdx.x == 0) {
    const int G = gridDim.x, g = blockIdx.x & 7;
    const unsigned gs = (unsigned)((G - g + 7) >> 3), ng = (unsigned)(G < 8 ? G : 8);
    unsigned* const err = sync + LIA_CHAIN_ERR_WORD;
    const unsigned prev = __hip_atomic_fetch_add(sync + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The acquire's cache invalidate (buffer_inv sc1: this CU's L1 forgets every line it holds) is issued BEFORE the wait and
    // completes under it (it takes ~1.7 us, MI355X_MICROARCH.md): from here to the workgroup barrier below no wave of this CU
    // loads a byte that another workgroup writes in this launch -- the polls are sc1 loads that bypass L1, the weight prefetch
    // in flight reads weights only -- so L1 cannot pick a stale line up again before the release is seen.
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (prev + 1 == (unsigned)k * gs) {
      __hip_atomic_fetch_add(sync + 16 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int spins = 0; CH_RLX_LOAD(sync + 16 * 32) < (unsigned)k * ng; ++spins)
        if (spins > spin_limit || ((spins & 63) == 63 && CH_RLX_LOAD(err) != 0u)) { ch_fail(err, err_host, 0x100u + (unsigned)k); break; }
      __hip_atomic_store(sync + (8 + g) * 32, (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      for (int spins = 0; CH_RLX_LOAD(sync + (8 + g) * 32) < (unsigned)k; ++spins) {
        if (spins > spin_limit || ((spins & 63) == 63 && CH_RLX_LOAD(err) != 0u)) { ch_fail(err, err_host, 0x200u + (unsigned)k); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the invalidate has completed when the workgroup goes on
  }
  __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------------------------
// REDUCE_NORM: workgroup = one output row.  Slabs added slice 0, 1, ... (the order of lia_splitk_reduce_norm_kernel), the row is
// finished (bias, residual: lia_epilogue.h), stored to om.base[0], and normalised from registers into post.out.  512 threads
// run the 1024 virtual threads of the stand-alone combine (lia_common.h): same sums, same order, same bits.
// ---------------------------------------------------------------------------------------------
template <int KIND, int NV>
__device__ __forceinline__ void ch_reduce_norm_row(const LiaChainOp& o, int m, float* red) {
  constexpr int VT = 2, RTH = LIA_ROW_THREADS / VT;
  const int tid = threadIdx.x;
  const int N = o.N, M = o.M, S = o.slices, nv = N >> 3;
  const LiaEpilogue ep = o.ep;
  const LiaPost post = o.post;
  uint4 v[VT][NV], gv[VT][NV], bv[VT][NV];
  f32x4 acc[VT][NV][2];
  const float* prow = o.slab + (long)m * N;
#pragma unroll
  for (int h = 0; h < VT; ++h)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = min(tid + RTH * h + LIA_ROW_THREADS * k, nv - 1);          // (clamped: idle lanes re-read the last piece and drop it)
      if (KIND != LIA_POST_NONE) gv[h][k] = *(const uint4*)(post.g + 8 * i);
      if (KIND == LIA_POST_LAYERNORM) bv[h][k] = *(const uint4*)(post.b + 8 * i);
      acc[h][k][0] = *(const f32x4*)(prow + 8 * i);
      acc[h][k][1] = *(const f32x4*)(prow + 8 * i + 4);
    }
  // the other slices GS at a time: their loads are in flight together (a slice at a time is S dependent L2 round trips, 4-8 us
  // for the 8 slabs of a down-proj row), the sum per element is still slice 0 + 1 + 2 + ...
  constexpr int GS = NV == 1 ? 4 : 2;
  for (int s0 = 1; s0 < S; s0 += GS) {
    f32x4 t[GS][VT][NV][2];
#pragma unroll
    for (int g = 0; g < GS; ++g) {
      const float* ps = prow + (long)min(s0 + g, S - 1) * M * N;          // (clamped: a slice beyond the last is loaded twice and dropped)
#pragma unroll
      for (int h = 0; h < VT; ++h)
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          const int i = min(tid + RTH * h + LIA_ROW_THREADS * k, nv - 1);
          t[g][h][k][0] = *(const f32x4*)(ps + 8 * i);
          t[g][h][k][1] = *(const f32x4*)(ps + 8 * i + 4);
        }
    }
#pragma unroll
    for (int g = 0; g < GS; ++g) {
      if (s0 + g < S) {
#pragma unroll
        for (int h = 0; h < VT; ++h)
#pragma unroll
          for (int k = 0; k < NV; ++k) { acc[h][k][0] += t[g][h][k][0]; acc[h][k][1] += t[g][h][k][1]; }
      }
    }
  }
#pragma unroll
  for (int h = 0; h < VT; ++h)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = tid + RTH * h + LIA_ROW_THREADS * k;
      v[h][k] = uint4{0u, 0u, 0u, 0u};
      if (i < nv) {
        const f32x4 lo = epilogue_quad(acc[h][k][0], m, 8 * i, ep);
        const f32x4 hi = epilogue_quad(acc[h][k][1], m, 8 * i + 4, ep);
        v[h][k] = uint4{pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
        lia_store16<true>(o.om.base[0] + (long)m * o.om.ld[0] + 8 * i, v[h][k]);
      }
    }
  if (KIND == LIA_POST_NONE) return;
  bf16_t* yr = post.out + (long)m * post.ldo;
  if (KIND == LIA_POST_LAYERNORM) row_layernorm_vt<NV, VT, true>(v, gv, bv, nv, N, post.eps, yr, red);
  else row_rmsnorm_vt<NV, VT, true>(v, gv, nv, N, post.eps, yr, red);
}

// ---------------------------------------------------------------------------------------------
// REDUCE_NORM spread over Q workgroups per row (Q = 2, 4 or 8: whatever G / M allows).  One workgroup per row reads S slabs of
// the whole row alone: 229 KB for an OPT-30B fc2 row, 12 us at the ~35 GB/s one CU takes in, while 3/4 of the chip waits at
// the next barrier.  Here workgroup (m, q) runs the row's virtual threads [q, q + 1) x 1024 / Q (lia_common.h: virtual thread v
// holds the 8-value pieces v, v + 1024, ...; 64 consecutive virtual threads = one virtual wave = one real wave here), and the
// row statistics cross workgroups as the 16 virtual-wave totals: each workgroup publishes its 16 / Q totals as 8-byte
// {tag, float} granules (one sc1 store each: the data is the flag, cdna_hip_programming.md Guideline 16 R2), reads all 16 of
// its row and adds them in wave order -- the sum block_sum_row_vt takes, so the same bits as the one-workgroup form and as
// the stand-alone combine.  tag is unique per (launch, step, statistic): nothing is ever zeroed.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float ch_row_exchange(float s, int q, int Q, unsigned long long* grow, unsigned tag, float* red, int spin_limit,
                                                 unsigned* err, unsigned* err_host) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, VW = LIA_ROW_WAVES / Q;
  const float ws = wave_sum(s);
  if (lane == 0 && wave < VW)
    __hip_atomic_store(grow + q * VW + wave, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(ws), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (wave == 0) {
    unsigned long long g = 0ull;
    for (int spins = 0;; ++spins) {
      bool ok = true;
      if (lane < LIA_ROW_WAVES) {
        g = __hip_atomic_load(grow + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = (unsigned)(g >> 32) == tag;
      }
      if (__all(ok)) break;
      if (spins > spin_limit || ((spins & 63) == 63 && CH_RLX_LOAD(err) != 0u)) {
        if (lane == 0) ch_fail(err, err_host, 0x300u + (tag & 0xffu));
        break;
      }
    }
    if (lane < LIA_ROW_WAVES) red[lane] = __uint_as_float((unsigned)g);
  }
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int w = 1; w < LIA_ROW_WAVES; ++w) t += red[w];
  __syncthreads();                                                   // `red` is written again by the next statistic / row
  return t;
}

template <int KIND, int NV>
__device__ __forceinline__ void ch_reduce_norm_part(const LiaChainOp& o, int m, int q, int Q, float* red, unsigned long long* gran, unsigned tag,
                                                    int spin_limit, unsigned* err, unsigned* err_host) {
  const int tid = threadIdx.x;
  const int nth = LIA_ROW_THREADS / Q;                     // virtual threads of this workgroup = its active real threads
  const bool active = tid < nth;
  const int vt = q * nth + (active ? tid : 0);             // this thread's virtual thread
  const int N = o.N, M = o.M, S = o.slices, nv = N >> 3;
  const LiaEpilogue ep = o.ep;
  const LiaPost post = o.post;
  uint4 v[NV], gv[NV], bv[NV], eb[NV], er[NV];
  f32x4 acc[NV][2];
  const float* prow = o.slab + (long)m * N;
  const bool hb = ep.bias != nullptr, hr = ep.residual != nullptr;
#pragma unroll
  for (int k = 0; k < NV; ++k) eb[k] = er[k] = uint4{0u, 0u, 0u, 0u};
  if (active) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = min(vt + LIA_ROW_THREADS * k, nv - 1);          // (clamped: idle lanes re-read the last piece and drop it)
      if (KIND != LIA_POST_NONE) gv[k] = *(const uint4*)(post.g + 8 * i);
      if (KIND == LIA_POST_LAYERNORM) bv[k] = *(const uint4*)(post.b + 8 * i);
      acc[k][0] = *(const f32x4*)(prow + 8 * i);
      acc[k][1] = *(const f32x4*)(prow + 8 * i + 4);
      if (hb) eb[k] = *(const uint4*)(ep.bias + 8 * i);
      if (hr) er[k] = *(const uint4*)(ep.residual + (long)m * ep.ldr + 8 * i);
    }
    constexpr int GS = NV == 1 ? 7 : 4;
    for (int s0 = 1; s0 < S; s0 += GS) {
      f32x4 t[GS][NV][2];
#pragma unroll
      for (int g = 0; g < GS; ++g) {
        const float* ps = prow + (long)min(s0 + g, S - 1) * M * N;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          const int i = min(vt + LIA_ROW_THREADS * k, nv - 1);
          t[g][k][0] = *(const f32x4*)(ps + 8 * i);
          t[g][k][1] = *(const f32x4*)(ps + 8 * i + 4);
        }
      }
#pragma unroll
      for (int g = 0; g < GS; ++g) {
        if (s0 + g < S) {
#pragma unroll
          for (int k = 0; k < NV; ++k) { acc[k][0] += t[g][k][0]; acc[k][1] += t[g][k][1]; }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = vt + LIA_ROW_THREADS * k;
    v[k] = uint4{0u, 0u, 0u, 0u};
    if (active && i < nv) {
      const f32x4 lo = epilogue_quad_pre(acc[k][0], uint2{eb[k].x, eb[k].y}, uint2{er[k].x, er[k].y}, hb, ep.relu, hr);
      const f32x4 hi = epilogue_quad_pre(acc[k][1], uint2{eb[k].z, eb[k].w}, uint2{er[k].z, er[k].w}, hb, ep.relu, hr);
      v[k] = uint4{pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
      lia_store16<true>(o.om.base[0] + (long)m * o.om.ld[0] + 8 * i, v[k]);
    }
  }
  if (KIND == LIA_POST_NONE) return;
  bf16_t* yr = post.out + (long)m * post.ldo;
  unsigned long long* grow0 = gran + (size_t)m * LIA_ROW_WAVES;                       // statistic 0 of row m
  unsigned long long* grow1 = gran + (size_t)(128 + m) * LIA_ROW_WAVES;               // statistic 1 (LayerNorm's second pass): an array of its own
  const int H = N;
  if (KIND == LIA_POST_RMSNORM) {
    // the expressions of row_rmsnorm_vt (lia_common.h), one virtual thread per thread
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (active && vt + LIA_ROW_THREADS * k < nv) {
        const uint32_t u[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { float a = bf2f(u[j] & 0xffff), c = bf2f(u[j] >> 16); ss += a * a + c * c; }
      }
    }
    const float rstd = 1.0f / sqrtf(ch_row_exchange(ss, q, Q, grow0, tag, red, spin_limit, err, err_host) / (float)H + post.eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = vt + LIA_ROW_THREADS * k;
      if (active && i < nv) {
        const uint32_t u[4] = {v[k].x, v[k].y, v[k].z, v[k].w}, gw[4] = {gv[k].x, gv[k].y, gv[k].z, gv[k].w};
        uint32_t ow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          ow[j] = pack_bf16x2(bf2f(gw[j] & 0xffff) * rbf(bf2f(u[j] & 0xffff) * rstd), bf2f(gw[j] >> 16) * rbf(bf2f(u[j] >> 16) * rstd));
        lia_store16<true>(yr + 8 * i, uint4{ow[0], ow[1], ow[2], ow[3]});
      }
    }
  } else {
    // the expressions of row_layernorm_vt (lia_common.h)
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (active && vt + LIA_ROW_THREADS * k < nv) {
        const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) sm += bf2f(w[j] & 0xffff) + bf2f(w[j] >> 16);
      }
    }
    const float mean = ch_row_exchange(sm, q, Q, grow0, tag, red, spin_limit, err, err_host) / (float)H;
    float qq = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (active && vt + LIA_ROW_THREADS * k < nv) {
        const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a = bf2f(w[j] & 0xffff) - mean, c = bf2f(w[j] >> 16) - mean;
          qq += a * a + c * c;
        }
      }
    }
    const float rstd = 1.0f / sqrtf(ch_row_exchange(qq, q, Q, grow1, tag + 1u, red, spin_limit, err, err_host) / (float)H + post.eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = vt + LIA_ROW_THREADS * k;
      if (active && i < nv) {
        const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w}, gw[4] = {gv[k].x, gv[k].y, gv[k].z, gv[k].w},
                       bw[4] = {bv[k].x, bv[k].y, bv[k].z, bv[k].w};
        uint32_t ow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float lo = (bf2f(w[j] & 0xffff) - mean) * rstd * bf2f(gw[j] & 0xffff) + bf2f(bw[j] & 0xffff);
          float hi = (bf2f(w[j] >> 16) - mean) * rstd * bf2f(gw[j] >> 16) + bf2f(bw[j] >> 16);
          ow[j] = pack_bf16x2(lo, hi);
        }
        lia_store16<true>(yr + 8 * i, uint4{ow[0], ow[1], ow[2], ow[3]});
      }
    }
  }
}

// lia_out_ptr with the cache position supplied by the launch (one program serves every decode step)
__device__ __forceinline__ bf16_t* ch_out_ptr(const LiaOutMap& o, int pos0, int m, int n) {
  const int s = n / o.seg_n;
  const int nn = n - s * o.seg_n;
  long row = m;
  if (o.cache_mode[s]) {
    const int b = m / o.T, t = m - b * o.T;
    row = (long)(pos0 + t) * o.Bc + o.b0 + b;
  }
  return o.base[s] + row * o.ld[s] + nn;
}

// REDUCE_MAP: element-parallel combine into the output map (q | k | v segments, KV-cache scatter), with RoPE when post says so:
// the arithmetic of lia_splitk_reduce_kernel / lia_splitk_reduce_rope_kernel with write-through stores.  A thread takes U
// quads per round and requests every slab value of all of them before the first add (a quad at a time is one dependent L2
// round trip per loop trip: 5 us for the 3.5 trips of an OPT-30B fc1); per element the slabs are still added slice 0, 1, ...
template <int U>
__device__ __forceinline__ void ch_slab_sums_issue(const float* slab, int S, int M, int N, const int (&m)[U], const int (&n)[U], f32x4 (&a)[U]) {
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = *(const f32x4*)(slab + (long)m[u] * N + n[u]);
}
template <int U>
__device__ __forceinline__ void ch_slab_sums_rest(const float* slab, int S, int M, int N, const int (&m)[U], const int (&n)[U], f32x4 (&a)[U]) {
  constexpr int GS = 4;
  for (int s0 = 1; s0 < S; s0 += GS) {
    f32x4 t[GS][U];
#pragma unroll
    for (int g = 0; g < GS; ++g) {
      const float* ps = slab + (long)min(s0 + g, S - 1) * M * N;
#pragma unroll
      for (int u = 0; u < U; ++u) t[g][u] = *(const f32x4*)(ps + (long)m[u] * N + n[u]);
    }
#pragma unroll
    for (int g = 0; g < GS; ++g) {
      if (s0 + g < S) {
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] += t[g][u];
      }
    }
  }
}

// the output map's per-segment tables in LDS: a lane's segment differs from its neighbour's, and fetching base / ld / cache_mode of
// segment s from the kernel-argument segment is a dependent vector-memory round trip per output quad (measured: 5-6 us per
// REDUCE_MAP step for three or four quads per thread)
struct ChOutLds {
  unsigned long long base[LIA_OUT_SEGS];
  long ld[LIA_OUT_SEGS];
  int cache_mode[LIA_OUT_SEGS];
};
__device__ __forceinline__ bf16_t* ch_out_ptr_lds(const ChOutLds* t, int seg_n, int T, int Bc, int b0, int pos0, int m, int n) {
  const int s = n / seg_n;
  const int nn = n - s * seg_n;
  long row = m;
  if (t->cache_mode[s]) {
    const int b = m / T, tt = m - b * T;
    row = (long)(pos0 + tt) * Bc + b0 + b;
  }
  return (bf16_t*)t->base[s] + row * t->ld[s] + nn;
}

__device__ __forceinline__ void ch_reduce_map(const LiaChainOp& o, int pos0, char* lds) {
  const int M = o.M, N = o.N, S = o.slices;
  const LiaEpilogue ep = o.ep;
  const LiaOutMap& om = o.om;
  LiaPost post = o.post;
  const int opos = o.use_pos0 ? pos0 : om.pos0;
  if (o.use_pos0) post.pos0 = pos0;
  ChOutLds* const tab = (ChOutLds*)lds;
  if (threadIdx.x < LIA_OUT_SEGS) {
    tab->base[threadIdx.x] = (unsigned long long)om.base[threadIdx.x];
    tab->ld[threadIdx.x] = om.ld[threadIdx.x];
    tab->cache_mode[threadIdx.x] = om.cache_mode[threadIdx.x];
  }
  __syncthreads();
  const int seg_n = om.seg_n, oT = om.T, oBc = om.Bc, ob0 = om.b0;
  const bool hb = ep.bias != nullptr, hr = ep.residual != nullptr;
  const int stride = (int)(gridDim.x * blockDim.x);
  const int first = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (post.kind == LIA_POST_ROPE) {
    const int hd = post.hd, half = hd >> 1, gph = half >> 2, heads = N / hd;
    const int total = M * heads * gph;
    constexpr int UU = 2;                                           // units (2 quads each) per round
    for (int i0 = first; i0 < total; i0 += UU * stride) {
      int mm[2 * UU], nn[2 * UU], hh[UU], gg[UU];
      bool live[UU];
      uint2 bb[2 * UU], rr[2 * UU], cs[UU][4];
#pragma unroll
      for (int u = 0; u < UU; ++u) {
        const int idx = i0 + u * stride;
        live[u] = idx < total;
        const int id = live[u] ? idx : total - 1;
        gg[u] = id % gph;
        const int mh = id / gph;
        hh[u] = mh % heads;
        mm[2 * u] = mm[2 * u + 1] = mh / heads;
        nn[2 * u] = hh[u] * hd + 4 * gg[u];
        nn[2 * u + 1] = nn[2 * u] + half;
      }
      f32x4 sum[2 * UU];
      ch_slab_sums_issue<2 * UU>(o.slab, S, M, N, mm, nn, sum);
#pragma unroll
      for (int u = 0; u < 2 * UU; ++u) {
        bb[u] = hb ? *(const uint2*)(ep.bias + nn[u]) : uint2{0u, 0u};
        rr[u] = hr ? *(const uint2*)(ep.residual + (long)mm[u] * ep.ldr + nn[u]) : uint2{0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < UU; ++u) {
        const int pos = post.pos0 + mm[2 * u] % post.T;
        const bf16_t* ct = post.cos_t + (long)pos * hd + 4 * gg[u];
        const bf16_t* st = post.sin_t + (long)pos * hd + 4 * gg[u];
        cs[u][0] = *(const uint2*)ct; cs[u][1] = *(const uint2*)(ct + half); cs[u][2] = *(const uint2*)st; cs[u][3] = *(const uint2*)(st + half);
      }
      ch_slab_sums_rest<2 * UU>(o.slab, S, M, N, mm, nn, sum);
#pragma unroll
      for (int u = 0; u < UU; ++u) {
        if (!live[u]) continue;
        const int m = mm[2 * u], n0 = nn[2 * u], n1 = nn[2 * u + 1];
        const f32x4 a = epilogue_quad_pre(sum[2 * u], bb[2 * u], rr[2 * u], hb, ep.relu, hr);
        const f32x4 b = epilogue_quad_pre(sum[2 * u + 1], bb[2 * u + 1], rr[2 * u + 1], hb, ep.relu, hr);
        uint2 oa{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3])}, ob{pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
        if (hh[u] < post.rot_heads) {
          uint2 ra, rb;
          lia_rope_pair(oa.x, ob.x, cs[u][0].x, cs[u][1].x, cs[u][2].x, cs[u][3].x, ra.x, rb.x);
          lia_rope_pair(oa.y, ob.y, cs[u][0].y, cs[u][1].y, cs[u][2].y, cs[u][3].y, ra.y, rb.y);
          oa = ra; ob = rb;
        }
        lia_store8<true>(ch_out_ptr_lds(tab, seg_n, oT, oBc, ob0, opos, m, n0), oa);
        lia_store8<true>(ch_out_ptr_lds(tab, seg_n, oT, oBc, ob0, opos, m, n1), ob);
      }
    }
    return;
  }
  const int nq = M * (N / 4), rowq = N / 4;
  constexpr int U = 4;
  for (int q0 = first; q0 < nq; q0 += U * stride) {
    int mm[U], nn[U];
    bool live[U];
    uint2 bb[U], rr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = q0 + u * stride;
      live[u] = q < nq;
      const int qq = live[u] ? q : nq - 1;
      mm[u] = qq / rowq;
      nn[u] = (qq - mm[u] * rowq) * 4;
    }
    f32x4 sum[U];
    ch_slab_sums_issue<U>(o.slab, S, M, N, mm, nn, sum);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bb[u] = hb ? *(const uint2*)(ep.bias + nn[u]) : uint2{0u, 0u};
      rr[u] = hr ? *(const uint2*)(ep.residual + (long)mm[u] * ep.ldr + nn[u]) : uint2{0u, 0u};
    }
    ch_slab_sums_rest<U>(o.slab, S, M, N, mm, nn, sum);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!live[u]) continue;
      const f32x4 r = epilogue_quad_pre(sum[u], bb[u], rr[u], hb, ep.relu, hr);
      lia_store8<true>(ch_out_ptr_lds(tab, seg_n, oT, oBc, ob0, opos, mm[u], nn[u]), uint2{pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])});
    }
  }
}

// ---------------------------------------------------------------------------------------------
// the kernel.  MT: 16-row blocks of x (M <= 16 MT).  The eight waves tile an item as (8 / MH) waves along the weight rows x MH
// along the x rows: a wave owns RT 16-row weight blocks x MT / MH 16-row x blocks, so an item has up to (8 / MH) RT 16 weight rows.
// Per chunk a wave reads RT + MT / MH fragments per 32-deep k-step from LDS (M = 128: 2 x 4 -> 12 reads for 16 MFMAs; one weight
// block x all eight x blocks per wave would be 18).  DW / DX: stages of the W / x rings.
// LDS: [DW x (rows x 128 B)][DX x (16 MT x 128 B)] (the REDUCE steps' few hundred bytes of scratch alias the idle x ring).
// ---------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void ch_wait_imm() {
  static_assert(N >= 0 && N <= 63, "vmcnt immediate");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else static_assert(N < 0, "add the immediate");
}

template <int MT, int RT, int MH, int DW, int DX>
__global__ __launch_bounds__(512) void lia_chain_kernel(const LiaChainProgram P, unsigned* sync, unsigned* err_host, int pos0, int spin_limit, unsigned long long* stamps,
                                                       unsigned long long* gran, unsigned epoch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WN = 8 / MH;                      // waves along the weight rows
  constexpr int MTW = MT / MH;                    // x blocks per wave
  constexpr int BNMAX = WN * RT * 16;             // weight rows of a full item
  constexpr int WSTAGE = BNMAX * 128;
  constexpr int XROWS = 16 * MT;
  constexpr int XSTAGE = XROWS * 128;
  constexpr int WJ = BNMAX / 32;                  // LDS-DMA instructions of one W-loading wave per chunk (8 rows each; the four cover 32 rows per round)
  constexpr int XJ = (XROWS + 31) / 32;           // ... of one x-loading wave
  constexpr int SJ = WJ > XJ ? WJ : XJ;
  static_assert(MT % MH == 0 && XROWS % 32 == 0 && DW >= 3 && DX >= 3 && DW <= 8 && DX <= 8, "geometry");
  // the program lives in the kernel-argument segment: constant address space, so every field is a scalar load however often
  // the kernel's own write-through stores and atomics clobber global memory
  const LiaChainOp* const prog = P.op;
  const int n_ops = P.n_ops;
  char* const wring = smem;
  char* const xring = wring + DW * WSTAGE;
  float* const red = (float*)xring;             // the row ops' scratch: used by the REDUCE steps only, when no x chunk is in flight
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loads_w = wave < 4;                  // waves 0-3 request W chunks, waves 4-7 x chunks
  const int w4 = wave & 3;
  const int l15 = lane & 15, lq = lane >> 4;
  const int srow = lane >> 3, sc = lane & 7;
  const int b = blockIdx.x, G = gridDim.x;
  const int rb0 = (wave % WN) * RT;               // this wave's first 16-row weight block of an item
  const int mb0 = (wave / WN) * MTW;              // ... and its first 16-row x block

  // diagnostic stamps (tools/chain_stamps.py; stamps == nullptr in production): per workgroup and step four 100 MHz clock reads --
  // step entered (behind its seam), first chunk landed / reduce loads issued, work done, stores drained
#define CH_STAMP(op_, k_) do { if (stamps && tid == 0) stamps[((size_t)b * LIA_CHAIN_MAX_OPS + (op_)) * 4 + (k_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  ChFifo fifo{0ull, 0};                           // this wave's requested-and-unconsumed chunks (of the ring it loads)
  int issued = 0;                                 // LDS-DMA loads this wave has issued so far
  int nw = 0, nx = 0;                             // chunks requested and not yet consumed, W ring / x ring (the same in every wave)
  int wslot_i = 0, xslot_i = 0, wslot_c = 0, xslot_c = 0;      // next slot to fill / to consume
  const bf16_t* src[SJ];                          // this thread's source rows: of the W item (waves 0-3) or of x (waves 4-7)

  // ---- W cursor: runs over this workgroup's (GEMM step, item, chunk) list, ahead of the consumer, across the seams ----
  int w_op = -1, w_item = 0, w_c = 0, w_cend = 0, w_bn = 0;
  bool w_valid = false;
  auto w_next_item = [&]() -> bool {
    if (w_op >= 0) {
      w_item += G;
      if (w_item < prog[w_op].n_items) return true;
    }
    for (++w_op; w_op < n_ops; ++w_op)
      if (prog[w_op].kind == LIA_CH_GEMM && b < prog[w_op].n_items) { w_item = b; return true; }
    return false;
  };
  auto w_setup = [&]() {
    const LiaChainOp& o = prog[w_op];
    const int split = o.split, tile = w_item / split, slice = w_item - tile * split;
    const int n0 = tile * o.bn;
    w_bn = o.bn;
    w_c = slice * o.cps;
    w_cend = min(o.nchunks, w_c + o.cps);
    if (loads_w) {
      const bf16_t* W = o.W;
      const long ldw = o.ldw;
      const int N = o.N;
#pragma unroll
      for (int j = 0; j < WJ; ++j) {
        const int row = 8 * w4 + 32 * j + srow;
        src[j] = W + (long)min(n0 + row, N - 1) * ldw + ((sc ^ ch_swz(row)) << 3);
      }
    }
  };
  // W loads of this wave for chunk w_c of the cursor's item: D = how many of its rounds hold rows of the item (WJ: all of them)
  auto load_w = [&](int d) {
    char* st = wring + wslot_i * WSTAGE;
    const long koff = (long)w_c * 64;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      if (j < d) {
        // weights are read once by one workgroup: non-temporal (aux = 2) keeps them from evicting x and the slabs in L2
        __builtin_amdgcn_global_load_lds(CH_GL_AS1(src[j] + koff), CH_LDS_AS3(st + j * 4096 + w4 * 1024), 16, 0, 2);
      }
    }
    issued += d;
    ch_fifo_push(fifo, issued);
  };
  auto w_rounds = [&]() -> int {                   // rounds of this W-loading wave that hold rows of the cursor's item
    const int d = (w_bn - 8 * w4 + 31) >> 5;
    return d < 0 ? 0 : (d > WJ ? WJ : d);
  };
  auto issue_w = [&]() {
    if (loads_w) load_w(w_rounds());
    else if (loads_w) ch_fifo_push(fifo, issued);
    wslot_i = wslot_i + 1 == DW ? 0 : wslot_i + 1;
    ++nw;
    if (++w_c >= w_cend) {
      w_valid = w_next_item();
      if (w_valid) w_setup();
    }
  };
  w_valid = w_next_item();
  if (w_valid) w_setup();
  for (int k = 0; k < DW - 1 && w_valid; ++k) issue_w();

  for (int opi = 0; opi < n_ops; ++opi) {
    if (opi > 0) {
      // seam: drain this wave's write-through stores (and, in order, the weight requests in front of them), meet, top the weight
      // ring up with the slot the last chunk freed, then arrive / wait / acquire
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (w_valid && nw < DW) issue_w();
      ch_grid_barrier(sync, opi, spin_limit, err_host);
    }
    const LiaChainOp& o = prog[opi];
    const int kind = o.kind;
    CH_STAMP(opi, 0);
    if (kind == LIA_CH_GEMM) {
      const int n_items = o.n_items, split = o.split, cps = o.cps, bn = o.bn, nchunks = o.nchunks, N = o.N, M = o.M;
      // ---- x cursor: this step's items only (the operand exists once the seam in front of the step is behind us) ----
      int x_item = b, x_c = 0, x_cend = 0;
      bool x_valid = b < n_items;
      auto x_setup = [&]() {
        const int tile = x_item / split, slice = x_item - tile * split;
        x_c = slice * cps;
        x_cend = min(nchunks, x_c + cps);
      };
      if (!loads_w) {                                              // the x rows are the same for every item of the step
        const bf16_t* x = o.x;
        const long ldx = o.ldx;
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
          const int row = 8 * w4 + 32 * j + srow;
          src[j] = x + (long)min(min(row, XROWS - 1), M - 1) * ldx + ((sc ^ ch_swz(row)) << 3);
        }
      }
      auto load_x = [&]() {
        char* st = xring + xslot_i * XSTAGE;
        const long koff = (long)x_c * 64;
#pragma unroll
        for (int j = 0; j < XJ; ++j)
          __builtin_amdgcn_global_load_lds(CH_GL_AS1(src[j] + koff), CH_LDS_AS3(st + j * 4096 + w4 * 1024), 16, 0, 0);
        issued += XJ;
        ch_fifo_push(fifo, issued);
      };
      auto issue_x = [&]() {
        if (!loads_w) load_x();
        else if (!loads_w) ch_fifo_push(fifo, issued);
        xslot_i = xslot_i + 1 == DX ? 0 : xslot_i + 1;
        ++nx;
        if (++x_c >= x_cend) {
          x_item += G;
          x_valid = x_item < n_items;
          if (x_valid) x_setup();
        }
      };
      if (x_valid) x_setup();
      for (int k = 0; k < DX - 1 && x_valid; ++k) issue_x();

      int nt = bn / 16 - rb0;                                      // how many of this wave's RT weight blocks exist in the step's items
      nt = nt < 0 ? 0 : (nt > RT ? RT : nt);
      for (int item = b; item < n_items; item += G) {
        const int tile = item / split, slice = item - tile * split;
        const int n0 = tile * bn, c0 = slice * cps, c1 = min(nchunks, c0 + cps);
        f32x4 acc[RT][MTW];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int p = 0; p < MTW; ++p) acc[t][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the chunk in the rings' consume slots: all fragment reads first, then the MFMAs (left to itself hipcc alternates
        // ds_read / s_waitcnt lgkmcnt(0) / MFMA: a dependent LDS round trip per MFMA)
        auto compute = [&]() {
          const char* wt = wring + wslot_c * WSTAGE;
          const char* xt = xring + xslot_c * XSTAGE;
          if (nt > 0) {
            bf16x8 af[2][RT], bq[2][MTW];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
              for (int t = 0; t < RT; ++t) {
                const int row = (rb0 + t) * 16 + l15;
                af[ks][t] = __builtin_bit_cast(bf16x8, *(const uint4*)(wt + row * 128 + (((4 * ks + lq) ^ ch_swz(row)) << 4)));
              }
#pragma unroll
              for (int p = 0; p < MTW; ++p) {
                const int row = 16 * (mb0 + p) + l15;
                bq[ks][p] = __builtin_bit_cast(bf16x8, *(const uint4*)(xt + row * 128 + (((4 * ks + lq) ^ ch_swz(row)) << 4)));
              }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (nt == RT) {                                        // (one decision per chunk, not one per MFMA)
#pragma unroll
              for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int p = 0; p < MTW; ++p)
#pragma unroll
                  for (int t = 0; t < RT; ++t) acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][t], bq[ks][p], acc[t][p], 0, 0, 0);
            } else {
#pragma unroll
              for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int p = 0; p < MTW; ++p)
#pragma unroll
                  for (int t = 0; t < RT; ++t)
                    if (t < nt) acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][t], bq[ks][p], acc[t][p], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          wslot_c = wslot_c + 1 == DW ? 0 : wslot_c + 1;
          xslot_c = xslot_c + 1 == DX ? 0 : xslot_c + 1;
          ch_fifo_pop(fifo);
        };
        // One chunk per round: wait for this wave's share of it, meet (every share has landed; the slot consumed last round is
        // free), request the chunks DW - 1 / DX - 1 ahead, compute.  LEAN rounds (the steady state inside an item: both cursors
        // stay in the item, both rings full, D loads per chunk from this wave) have no decisions left in them -- the general
        // round's bookkeeping is ~25 scalar branches, which paced the first version of this loop at ~1 us per chunk.
        int c = c0;
        while (c < c1) {
          const int L = min(min(w_cend - w_c, x_cend - x_c) - 1, c1 - c);   // requests that stay inside the cursors' items, rounds that stay inside this one
          const int d = loads_w ? w_rounds() : XJ;
          const bool steady = L > 0 && w_valid && x_valid && nw == DW - 1 && nx == DX - 1 &&
                              ch_fifo_behind(fifo, issued) == (loads_w ? DW - 2 : DX - 2) * d;
          if (steady && (!loads_w || d >= WJ - 1)) {
#define CH_LEAN_ROUNDS(WAIT_N, LOADS)                                                                              \
            for (int i = 0; i < L; ++i) {                                                                          \
              ch_wait_imm<WAIT_N>();                                                                               \
              __builtin_amdgcn_s_barrier();                                                                        \
              LOADS;                                                                                               \
              wslot_i = wslot_i + 1 == DW ? 0 : wslot_i + 1;                                                       \
              xslot_i = xslot_i + 1 == DX ? 0 : xslot_i + 1;                                                       \
              ++w_c;                                                                                               \
              ++x_c;                                                                                               \
              compute();                                                                                           \
            }
            if (!loads_w) { CH_LEAN_ROUNDS((DX - 2) * XJ, load_x()) }
            else if (d == WJ) { CH_LEAN_ROUNDS((DW - 2) * WJ, load_w(WJ)) }
            else { CH_LEAN_ROUNDS((DW - 2) * (WJ - 1), load_w(WJ - 1)) }
#undef CH_LEAN_ROUNDS
            c += L;
            continue;
          }
          // general round.  Everything this wave requested up to its share of the chunk must be back: only the loads it issued
          // behind that may still be out (stores are not counted: a stricter wait, never a weaker one)
          ch_wait_vmcnt(ch_fifo_behind(fifo, issued));
          __builtin_amdgcn_s_barrier();
          if (stamps && c == c0 && item == b) CH_STAMP(opi, 1);
          if (w_valid && nw < DW) issue_w();
          if (x_valid && nx < DX) issue_x();
          compute();
          --nw;
          --nx;
          ++c;
        }
        // keep the last MFMA clear of the accumulator reads (hipcc, ROCm 7.2, was seen to read a just-written accumulator with
        // too few wait states at a branch target: lia_gemm.hip)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int direct = o.direct;
        if (direct == LIA_CH_DIRECT_NONE) {
          float* pp = o.slab + (long)slice * M * N;
#pragma unroll
          for (int t = 0; t < RT; ++t) {
            const int nn = n0 + (rb0 + t) * 16 + 4 * lq;
            if (t < nt && nn < N) {
#pragma unroll
              for (int p = 0; p < MTW; ++p) {
                const int m = 16 * (mb0 + p) + l15;
                if (m < M) {
                  const f32x4 v = acc[t][p];
                  lia_store16<true>(pp + (long)m * N + nn, uint4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])});
                }
              }
            }
          }
        } else if (direct == LIA_CH_DIRECT_PLAIN) {
          const LiaEpilogue ep = o.ep;
          const LiaOutMap& om = o.om;
#pragma unroll
          for (int t = 0; t < RT; ++t) {
            const int nn = n0 + (rb0 + t) * 16 + 4 * lq;
            if (t < nt && nn < N) {
#pragma unroll
              for (int p = 0; p < MTW; ++p) {
                const int m = 16 * (mb0 + p) + l15;
                if (m < M) {
                  const f32x4 r = epilogue_quad(acc[t][p], m, nn, ep);
                  lia_store8<true>(lia_out_ptr(om, m, nn), uint2{pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])});
                }
              }
            }
          }
        } else {
          // gate | up projection in ONE slice: park the finished bf16 tile [m][bn] in the x ring (dead: this step has one item
          // per workgroup, its last chunk is consumed), then pair 32 gate | 32 up column blocks: act = silu(gate) * up with the
          // device function of the stand-alone kernel and of lia_gemm_skinny2_kernel's epilogue
          const LiaEpilogue ep = o.ep;
          const LiaOutMap& om = o.om;
          __builtin_amdgcn_s_barrier();                            // every wave is done reading the last x chunk
          char* park = xring;
#pragma unroll
          for (int t = 0; t < RT; ++t) {
            const int nn = n0 + (rb0 + t) * 16 + 4 * lq;
            if (t < nt) {
#pragma unroll
              for (int p = 0; p < MTW; ++p) {
                const int ml = 16 * (mb0 + p) + l15;
                const f32x4 q = epilogue_quad(acc[t][p], min(ml, M - 1), min(nn, N - 4), ep);
                *(uint2*)(park + ((long)ml * bn + (nn - n0)) * 2) = uint2{pack_bf16x2(q[0], q[1]), pack_bf16x2(q[2], q[3])};
              }
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          const int OQ = bn / 8;                                    // output quads per tile row (bn / 2 columns)
          for (int q = tid; q < XROWS * OQ; q += 512) {
            const int ml = q / OQ, c4 = (q - ml * OQ) * 4;
            const int ng = (c4 / LIA_GU_BLOCK) * (2 * LIA_GU_BLOCK) + (c4 % LIA_GU_BLOCK);
            if (ml >= M || n0 + ng + LIA_GU_BLOCK >= N) continue;
            const uint2 gq = *(const uint2*)(park + ((long)ml * bn + ng) * 2);
            const uint2 uq = *(const uint2*)(park + ((long)ml * bn + ng + LIA_GU_BLOCK) * 2);
            lia_store8<true>(lia_out_ptr(om, ml, n0 / 2 + c4), uint2{lia_silu_mul_pair(gq.x, uq.x), lia_silu_mul_pair(gq.y, uq.y)});
          }
        }
      }
    } else if (kind == LIA_CH_REDUCE_NORM) {
      const int pk = o.post.kind, nv = o.N >> 3, Mr = o.M;
      // workgroups per row: as many as the grid has (2, 4 or 8), each with its share of the row's virtual waves
      int Q = 1;
      if (gran != nullptr && pk != LIA_POST_NONE && Mr <= 128) { if (8 * Mr <= G) Q = 8; else if (4 * Mr <= G) Q = 4; else if (2 * Mr <= G) Q = 2; }
      if (Q > 1) {
        const int m = b % Mr, q = b / Mr;                        // partners b, b + Mr, ...: the same blockIdx % 8 group whenever 8 | Mr
        if (q < Q) {
          unsigned* const err = sync + LIA_CHAIN_ERR_WORD;
          const unsigned tag = (epoch << 5) + (unsigned)opi * 2u + 1u;
          if (nv <= LIA_ROW_THREADS) {
            if (pk == LIA_POST_LAYERNORM) ch_reduce_norm_part<LIA_POST_LAYERNORM, 1>(o, m, q, Q, red, gran, tag, spin_limit, err, err_host);
            else ch_reduce_norm_part<LIA_POST_RMSNORM, 1>(o, m, q, Q, red, gran, tag, spin_limit, err, err_host);
          } else {
            if (pk == LIA_POST_LAYERNORM) ch_reduce_norm_part<LIA_POST_LAYERNORM, 2>(o, m, q, Q, red, gran, tag, spin_limit, err, err_host);
            else ch_reduce_norm_part<LIA_POST_RMSNORM, 2>(o, m, q, Q, red, gran, tag, spin_limit, err, err_host);
          }
        }
      } else {
        for (int m = b; m < Mr; m += G) {
          if (nv <= LIA_ROW_THREADS) {
            if (pk == LIA_POST_LAYERNORM) ch_reduce_norm_row<LIA_POST_LAYERNORM, 1>(o, m, red);
            else if (pk == LIA_POST_RMSNORM) ch_reduce_norm_row<LIA_POST_RMSNORM, 1>(o, m, red);
            else ch_reduce_norm_row<LIA_POST_NONE, 1>(o, m, red);
          } else {
            if (pk == LIA_POST_LAYERNORM) ch_reduce_norm_row<LIA_POST_LAYERNORM, 2>(o, m, red);
            else if (pk == LIA_POST_RMSNORM) ch_reduce_norm_row<LIA_POST_RMSNORM, 2>(o, m, red);
            else ch_reduce_norm_row<LIA_POST_NONE, 2>(o, m, red);
          }
          __syncthreads();                                         // `red` is reused by the next row
        }
      }
    } else if (kind == LIA_CH_REDUCE_MAP) {
      ch_reduce_map(o, pos0, (char*)red);
    }
    CH_STAMP(opi, 2);
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CH_STAMP(opi, 3); }
  }
#undef CH_STAMP
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct ChGeom { int mt, rt, mh, dw, dx; };
static bool ch_geom(int M, ChGeom* g) {
  if (M <= 0 || M > 128) return false;
  if (M > 64) *g = ChGeom{8, 2, 2, 7, 3};       // items of <= 128 weight rows: 7 x 16 KB + 3 x 16 KB = 160 KB, all of the CU's LDS
  else if (M > 32) *g = ChGeom{4, 2, 1, 4, 3};  // items of <= 256 weight rows: 4 x 32 KB + 3 x  8 KB = 152 KB
  else *g = ChGeom{2, 2, 1, 4, 4};              // 4 x 32 KB + 4 x  4 KB = 144 KB
  return true;
}
static int ch_bn_max(const ChGeom& g) { return (8 / g.mh) * g.rt * 16; }
static size_t ch_lds_bytes(const ChGeom& g) { return (size_t)g.dw * ch_bn_max(g) * 128 + (size_t)g.dx * 16 * g.mt * 128; }

extern "C" int lia_chain_supported(int M) {
  ChGeom g;
  return ch_geom(M, &g) ? 0 : -1;
}

extern "C" int lia_chain_cu_count(int device) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) return 0;
  return p.multiProcessorCount;
}

// One item per CU wherever the shape allows.  Cost of a candidate (bn rows x cps chunks per item, `rounds` items per CU) in
// per-CU bytes, weighted by where they come from: weights from HBM (~25 GB/s per CU), the x slice from L2 (~60 GB/s per CU: every
// item re-reads it), the fp32 partial slab written through and read back by the combine.
extern "C" int lia_chain_plan_gemm(int M, int N, int K, int glu, int n_cu, LiaChainPlan* out) {
  ChGeom g;
  if (!out || !ch_geom(M, &g) || N <= 0 || K <= 0 || (N % 16) || (K % 128) || n_cu <= 0) return -1;
  const int nchunks = K / 64, bn_max = ch_bn_max(g), xrows = 16 * g.mt;
  double best = 1e30;
  LiaChainPlan bp{0, 0, 0};
  for (int bn = 16; bn <= bn_max; bn += 16) {
    if (glu && (bn % (2 * LIA_GU_BLOCK) || (size_t)xrows * bn * 2 > (size_t)g.dx * xrows * 128)) continue;
    const int tiles = (N + bn - 1) / bn;
    for (int s0 = 1; s0 <= (glu ? 1 : 8); ++s0) {   // (8: what the per-op path's slab workspace holds too)
      const int cps = (nchunks + s0 - 1) / s0, split = (nchunks + cps - 1) / cps;
      if (split != s0) continue;                         // (the same slices under a smaller count: already seen)
      if (split > 1 && cps < 8) continue;                // keep a pipeline's worth of chunks per item
      const long items = (long)tiles * split;
      if (glu && items > n_cu) continue;                 // the glu epilogue parks its tile in the x ring: one item per workgroup
      const long rounds = (items + n_cu - 1) / n_cu;
      const double w_bytes = (double)bn * cps * 128, x_bytes = (double)xrows * cps * 128;
      const double slab = split > 1 ? (double)M * bn * 4 : 0.0;
      const double per_item = w_bytes / 25.0 + x_bytes / 60.0 + slab / 40.0 + 1500.0;      // ns; 1.5 us of fixed cost per item
      const double combine = split > 1 ? (double)split * M * N * 4 / n_cu / 60.0 : 0.0;
      const double cost = rounds * per_item + combine;
      if (cost < best) { best = cost; bp = LiaChainPlan{bn, split, cps}; }
    }
  }
  if (bp.bn == 0) return -1;
  *out = bp;
  return 0;
}

// diagnostic: a device buffer of `slots` x (n_cu x LIA_CHAIN_MAX_OPS x 4) words; launch i stamps into slot i % slots.  nullptr = off.
static unsigned long long* g_chain_stamps = nullptr;
static int g_chain_stamp_slots = 0;
static long g_chain_stamp_launch = 0;
extern "C" void lia_chain_set_stamps(unsigned long long* dev_buffer, int slots) { g_chain_stamps = dev_buffer; g_chain_stamp_slots = slots; g_chain_stamp_launch = 0; }

template <int MT, int RT, int MH, int DW, int DX>
static int ch_launch(const LiaChainProgram& prog, unsigned* sync, unsigned* err_host, int pos0, int n_cu, size_t lds, int spin_limit, unsigned long long* gran,
                     unsigned epoch, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)lia_chain_kernel<MT, RT, MH, DW, DX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    attr = true;
  }
  hipLaunchKernelGGL((lia_chain_kernel<MT, RT, MH, DW, DX>), dim3(n_cu), dim3(512), lds, st, prog, sync, err_host, pos0, spin_limit,
                     g_chain_stamps ? g_chain_stamps + (size_t)(g_chain_stamp_launch++ % g_chain_stamp_slots) * n_cu * LIA_CHAIN_MAX_OPS * 4 : nullptr, gran, epoch);
  return 0;
}

extern "C" int lia_chain_launch(const LiaChainProgram* prog, int M, unsigned* sync_block, unsigned* err_host, int pos0, int n_cu, unsigned long long* gran,
                                unsigned epoch, hipStream_t st) {
  ChGeom g;
  if (!prog || prog->n_ops <= 0 || prog->n_ops > LIA_CHAIN_MAX_OPS || !sync_block || n_cu <= 0 || !ch_geom(M, &g)) return -1;
  // a poll is ~1-2 us (an L2 round trip + s_sleep): 400k polls bound a lost barrier to well under a second
  const int spin_limit = 400000;
  const size_t lds = ch_lds_bytes(g);
  if (g.mt == 8) return ch_launch<8, 2, 2, 7, 3>(*prog, sync_block, err_host, pos0, n_cu, lds, spin_limit, gran, epoch, st);
  if (g.mt == 4) return ch_launch<4, 2, 1, 4, 3>(*prog, sync_block, err_host, pos0, n_cu, lds, spin_limit, gran, epoch, st);
  return ch_launch<2, 2, 1, 4, 4>(*prog, sync_block, err_host, pos0, n_cu, lds, spin_limit, gran, epoch, st);
}
