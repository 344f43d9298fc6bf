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
  if (n == 4) CH_WV(4);
  else if (n == 3) CH_WV(3);
  else if (n == 5) CH_WV(5);
  else if (n >= 10) CH_WV(10);
  else if (n >= 8) CH_WV(8);
  else if (n >= 6) CH_WV(6);
  else if (n == 2) CH_WV(2);
  else if (n == 1) CH_WV(1);
  else CH_WV(0);
}
#undef CH_WV

// Which requests may still be out when a chunk is consumed: per ring a FIFO, one byte per requested-and-unconsumed chunk (oldest
// in the low byte), holding the wave's load counter (mod 256) right after that chunk's own loads were issued.  One 64-bit scalar
// per ring (<= 8 stages; far fewer than 256 loads are ever in flight): no register arrays, no dynamic indexing.
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
    if (prev + 1 == (unsigned)k * gs) {
      __hip_atomic_fetch_add(sync + 16 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int spins = 0; CH_RLX_LOAD(sync + 16 * 32) < (unsigned)k * ng; ++spins) {
        if (spins > spin_limit || CH_RLX_LOAD(err) != 0u) { ch_fail(err, err_host, 0x100u + (unsigned)k); break; }
        __builtin_amdgcn_s_sleep(1);
      }
      __hip_atomic_store(sync + (8 + g) * 32, (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      for (int spins = 0; CH_RLX_LOAD(sync + (8 + g) * 32) < (unsigned)k; ++spins) {
        if (spins > spin_limit || CH_RLX_LOAD(err) != 0u) { ch_fail(err, err_host, 0x200u + (unsigned)k); break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // buffer_inv sc1: this CU's L1 forgets every line it holds
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // ... and has forgotten them when the workgroup goes on
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
  for (int s = 1; s < S; ++s) {
    const float* ps = prow + (long)s * M * N;
    f32x4 t[VT][NV][2];
#pragma unroll
    for (int h = 0; h < VT; ++h)
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = min(tid + RTH * h + LIA_ROW_THREADS * k, nv - 1);
        t[h][k][0] = *(const f32x4*)(ps + 8 * i);
        t[h][k][1] = *(const f32x4*)(ps + 8 * i + 4);
      }
#pragma unroll
    for (int h = 0; h < VT; ++h)
#pragma unroll
      for (int k = 0; k < NV; ++k) { acc[h][k][0] += t[h][k][0]; acc[h][k][1] += t[h][k][1]; }
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
// the bodies of lia_splitk_reduce_kernel / lia_splitk_reduce_rope_kernel with write-through stores.
__device__ __forceinline__ void ch_reduce_map(const LiaChainOp& o, int pos0) {
  const int M = o.M, N = o.N, S = o.slices;
  const LiaEpilogue ep = o.ep;
  const LiaOutMap& om = o.om;                    // (read in place: a copy with run-time segment indices would live in scratch)
  LiaPost post = o.post;
  const int opos = o.use_pos0 ? pos0 : om.pos0;
  if (o.use_pos0) post.pos0 = pos0;
  const long stride = (long)gridDim.x * blockDim.x;
  if (post.kind == LIA_POST_ROPE) {
    const int hd = post.hd, half = hd >> 1, gph = half >> 2, heads = N / hd;
    const long total = (long)M * heads * gph;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
      const int gi = (int)(idx % gph);
      const long mh = idx / gph;
      const int h = (int)(mh % heads), m = (int)(mh / heads);
      const int n0 = h * hd + 4 * gi, n1 = n0 + half;
      const f32x4 a = epilogue_quad(splitk_sum(o.slab, S, M, N, m, n0), m, n0, ep);
      const f32x4 b = epilogue_quad(splitk_sum(o.slab, S, M, N, m, n1), m, n1, ep);
      uint2 oa{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3])}, ob{pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
      if (h < post.rot_heads) {
        const int pos = post.pos0 + m % post.T;
        const uint2 c0 = *(const uint2*)(post.cos_t + (long)pos * hd + 4 * gi), c1 = *(const uint2*)(post.cos_t + (long)pos * hd + half + 4 * gi);
        const uint2 s0 = *(const uint2*)(post.sin_t + (long)pos * hd + 4 * gi), s1 = *(const uint2*)(post.sin_t + (long)pos * hd + half + 4 * gi);
        uint2 ra, rb;
        lia_rope_pair(oa.x, ob.x, c0.x, c1.x, s0.x, s1.x, ra.x, rb.x);
        lia_rope_pair(oa.y, ob.y, c0.y, c1.y, s0.y, s1.y, ra.y, rb.y);
        oa = ra; ob = rb;
      }
      lia_store8<true>(ch_out_ptr(om, opos, m, n0), oa);
      lia_store8<true>(ch_out_ptr(om, opos, m, n1), ob);
    }
    return;
  }
  const long nq = (long)M * (N / 4);
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
    const int m = (int)(q / (N / 4));
    const int n = (int)(q - (long)m * (N / 4)) * 4;
    const f32x4 r = epilogue_quad(splitk_sum(o.slab, S, M, N, m, n), m, n, ep);
    lia_store8<true>(ch_out_ptr(om, opos, m, n), uint2{pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])});
  }
}

// ---------------------------------------------------------------------------------------------
// the kernel.  MT: 16-row blocks of x (M <= 16 MT), RT: 16-row weight blocks per wave (items of up to 128 RT weight rows),
// DW / DX: stages of the W / x rings.  LDS: [512 B scratch][DW x (128 RT x 128 B)][DX x (16 MT x 128 B)].
// ---------------------------------------------------------------------------------------------
#define CH_SCRATCH 512

template <int MT, int RT, int DW, int DX>
__global__ __launch_bounds__(512) void lia_chain_kernel(const LiaChainProgram P, unsigned* sync, unsigned* err_host, int pos0, int spin_limit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WSTAGE = 128 * RT * 128;
  constexpr int XROWS = 16 * MT;
  constexpr int XSTAGE = XROWS * 128;
  constexpr int WJ = 2 * RT;                      // LDS-DMA rounds (64 rows each) that cover a W stage
  constexpr int XJ = (XROWS + 63) / 64;           // ... an x stage
  // the program lives in the kernel-argument segment: constant address space, so every field is a scalar load however often
  // the kernel's own write-through stores and atomics clobber global memory
  const LiaChainOp* const prog = P.op;
  const int n_ops = P.n_ops;
  float* const red = (float*)smem;
  char* const wring = smem + CH_SCRATCH;
  char* const xring = wring + DW * WSTAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int srow = lane >> 3, sc = lane & 7;
  const int b = blockIdx.x, G = gridDim.x;

  static_assert(DW <= 8 && DX <= 8, "ChFifo holds 8 entries");
  ChFifo fw{0ull, 0}, fx{0ull, 0};                // requested-and-unconsumed chunks of the W / x ring (fw.n, fx.n: how many)
  int issued = 0;                                 // LDS-DMA loads this wave has issued so far
  int wslot_i = 0, xslot_i = 0, wslot_c = 0, xslot_c = 0;      // next slot to fill / to consume

  // ---- W cursor: runs over this workgroup's (GEMM step, item, chunk) list, ahead of the consumer, across the seams ----
  int w_op = -1, w_item = 0, w_c = 0, w_cend = 0, w_bn = 0;
  bool w_valid = false;
  const bf16_t* wsrc[WJ];
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
    const bf16_t* W = o.W;
    const long ldw = o.ldw;
    const int N = o.N;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int row = 8 * wave + 64 * j + srow;
      wsrc[j] = W + (long)min(n0 + row, N - 1) * ldw + ((sc ^ ch_swz(row)) << 3);
    }
  };
  auto issue_w = [&]() {
    char* st = wring + wslot_i * WSTAGE;
    const long koff = (long)w_c * 64;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      if (8 * wave + 64 * j < w_bn) {
        // weights are read once by one workgroup: non-temporal (aux = 2) keeps them from evicting x and the slabs in L2
        __builtin_amdgcn_global_load_lds(CH_GL_AS1(wsrc[j] + koff), CH_LDS_AS3(st + j * 8192 + wave * 1024), 16, 0, 2);
        ++issued;
      }
    }
    ch_fifo_push(fw, issued);
    wslot_i = wslot_i + 1 == DW ? 0 : wslot_i + 1;
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
      if (w_valid && fw.n < DW) issue_w();
      ch_grid_barrier(sync, opi, spin_limit, err_host);
    }
    const LiaChainOp& o = prog[opi];
    const int kind = o.kind;
    if (kind == LIA_CH_GEMM) {
      const int n_items = o.n_items, split = o.split, cps = o.cps, bn = o.bn, nchunks = o.nchunks, N = o.N, M = o.M;
      // ---- x cursor: this step's items only (the operand exists once the seam in front of the step is behind us) ----
      int x_item = b, x_c = 0, x_cend = 0;
      bool x_valid = b < n_items;
      const bf16_t* xsrc[XJ];
      auto x_setup = [&]() {
        const int tile = x_item / split, slice = x_item - tile * split;
        x_c = slice * cps;
        x_cend = min(nchunks, x_c + cps);
        const bf16_t* x = o.x;
        const long ldx = o.ldx;
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
          const int row = 8 * wave + 64 * j + srow;
          xsrc[j] = x + (long)min(min(row, XROWS - 1), M - 1) * ldx + ((sc ^ ch_swz(row)) << 3);
        }
      };
      auto issue_x = [&]() {
        char* st = xring + xslot_i * XSTAGE;
        const long koff = (long)x_c * 64;
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
          if (8 * wave + 64 * j < XROWS) {
            __builtin_amdgcn_global_load_lds(CH_GL_AS1(xsrc[j] + koff), CH_LDS_AS3(st + j * 8192 + wave * 1024), 16, 0, 0);
            ++issued;
          }
        }
        ch_fifo_push(fx, issued);
        xslot_i = xslot_i + 1 == DX ? 0 : xslot_i + 1;
        if (++x_c >= x_cend) {
          x_item += G;
          x_valid = x_item < n_items;
          if (x_valid) x_setup();
        }
      };
      if (x_valid) x_setup();
      for (int k = 0; k < DX - 1 && x_valid; ++k) issue_x();

      const int rb0 = wave * RT;                                   // this wave's first 16-row block of the item
      int nt = bn / 16 - rb0;                                      // how many of its RT blocks exist in this step's items
      nt = nt < 0 ? 0 : (nt > RT ? RT : nt);
      for (int item = b; item < n_items; item += G) {
        const int tile = item / split, slice = item - tile * split;
        const int n0 = tile * bn, c0 = slice * cps, c1 = min(nchunks, c0 + cps);
        f32x4 acc[RT][MT];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int p = 0; p < MT; ++p) acc[t][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = c0; c < c1; ++c) {
          // chunk landed?  Everything this wave requested up to the later of (W chunk, x chunk) must be back: only the loads
          // issued behind BOTH may still be out (stores are not counted: a stricter wait, never a weaker one)
          ch_wait_vmcnt(min(ch_fifo_behind(fw, issued), ch_fifo_behind(fx, issued)));
          __builtin_amdgcn_s_barrier();                            // every wave's share has landed; the slot consumed last round is free
          if (w_valid && fw.n < DW) issue_w();
          if (x_valid && fx.n < DX) issue_x();
          const char* wt = wring + wslot_c * WSTAGE;
          const char* xt = xring + xslot_c * XSTAGE;
          if (nt > 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              bf16x8 a[RT];
#pragma unroll
              for (int t = 0; t < RT; ++t) {
                const int row = (rb0 + t) * 16 + l15;
                a[t] = __builtin_bit_cast(bf16x8, *(const uint4*)(wt + row * 128 + (((4 * ks + lq) ^ ch_swz(row)) << 4)));
              }
#pragma unroll
              for (int p = 0; p < MT; ++p) {
                const int row = 16 * p + l15;
                const bf16x8 bb = __builtin_bit_cast(bf16x8, *(const uint4*)(xt + row * 128 + (((4 * ks + lq) ^ ch_swz(row)) << 4)));
#pragma unroll
                for (int t = 0; t < RT; ++t)
                  if (t < nt) acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], bb, acc[t][p], 0, 0, 0);
              }
            }
          }
          wslot_c = wslot_c + 1 == DW ? 0 : wslot_c + 1;
          xslot_c = xslot_c + 1 == DX ? 0 : xslot_c + 1;
          ch_fifo_pop(fw);
          ch_fifo_pop(fx);
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
              for (int p = 0; p < MT; ++p) {
                const int m = 16 * p + l15;
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
              for (int p = 0; p < MT; ++p) {
                const int m = 16 * p + l15;
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
              for (int p = 0; p < MT; ++p) {
                const int ml = 16 * p + l15;
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
      const int pk = o.post.kind, nv = o.N >> 3;
      for (int m = b; m < o.M; m += G) {
        if (nv <= LIA_ROW_THREADS) {
          if (pk == LIA_POST_LAYERNORM) ch_reduce_norm_row<LIA_POST_LAYERNORM, 1>(o, m, red);
          else if (pk == LIA_POST_RMSNORM) ch_reduce_norm_row<LIA_POST_RMSNORM, 1>(o, m, red);
          else ch_reduce_norm_row<LIA_POST_NONE, 1>(o, m, red);
        } else {
          if (pk == LIA_POST_LAYERNORM) ch_reduce_norm_row<LIA_POST_LAYERNORM, 2>(o, m, red);
          else if (pk == LIA_POST_RMSNORM) ch_reduce_norm_row<LIA_POST_RMSNORM, 2>(o, m, red);
          else ch_reduce_norm_row<LIA_POST_NONE, 2>(o, m, red);
        }
        __syncthreads();                                           // `red` is reused by the next row
      }
    } else if (kind == LIA_CH_REDUCE_MAP) {
      ch_reduce_map(o, pos0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct ChGeom { int mt, rt, dw, dx; };
static bool ch_geom(int M, ChGeom* g) {
  if (M <= 0 || M > 128) return false;
  if (M > 64) *g = ChGeom{8, 1, 6, 3};          // 6 x 16 KB + 3 x 16 KB = 144 KB
  else if (M > 32) *g = ChGeom{4, 2, 4, 3};     // 4 x 32 KB + 3 x  8 KB = 152 KB
  else *g = ChGeom{2, 2, 4, 4};                 // 4 x 32 KB + 4 x  4 KB = 144 KB
  return true;
}
static size_t ch_lds_bytes(const ChGeom& g) { return CH_SCRATCH + (size_t)g.dw * 128 * g.rt * 128 + (size_t)g.dx * 16 * g.mt * 128; }

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
  const int nchunks = K / 64, bn_max = 128 * g.rt, xrows = 16 * g.mt;
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

template <int MT, int RT, int DW, int DX>
static int ch_launch(const LiaChainProgram& prog, unsigned* sync, unsigned* err_host, int pos0, int n_cu, size_t lds, int spin_limit, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)lia_chain_kernel<MT, RT, DW, DX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    attr = true;
  }
  hipLaunchKernelGGL((lia_chain_kernel<MT, RT, DW, DX>), dim3(n_cu), dim3(512), lds, st, prog, sync, err_host, pos0, spin_limit);
  return 0;
}

extern "C" int lia_chain_launch(const LiaChainProgram* prog, int M, unsigned* sync_block, unsigned* err_host, int pos0, int n_cu, hipStream_t st) {
  ChGeom g;
  if (!prog || prog->n_ops <= 0 || prog->n_ops > LIA_CHAIN_MAX_OPS || !sync_block || n_cu <= 0 || !ch_geom(M, &g)) return -1;
  // a poll is ~1-2 us (an L2 round trip + s_sleep): 400k polls bound a lost barrier to well under a second
  static const int spin_limit = [] { const char* e = getenv("LIA_CHAIN_SPIN_LIMIT"); return e ? atoi(e) : 400000; }();
  const size_t lds = ch_lds_bytes(g);
  if (g.mt == 8) return ch_launch<8, 1, 6, 3>(*prog, sync_block, err_host, pos0, n_cu, lds, spin_limit, st);
  if (g.mt == 4) return ch_launch<4, 2, 4, 3>(*prog, sync_block, err_host, pos0, n_cu, lds, spin_limit, st);
  return ch_launch<2, 2, 4, 4>(*prog, sync_block, err_host, pos0, n_cu, lds, spin_limit, st);
}
