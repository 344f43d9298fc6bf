// bf16 MFMA GEMMs for the LIA GPU sub-layers:  y[M,N] = epilogue(x[M,K] . W[N,K]^T)
//
// Replaces the reference's  torch.matmul(x, w.t()) + b  [+ relu] [residual + .]  sequences
// (decoder.py:79-105, 225-229, 282-285, 306-310; attentions.py:393-394, 418) and the per-use
// un-blocking copy of every streamed weight (attentions.py:381-382,412; decoder.py:25-58): weights are
// kept row-major [N,K] on the host, so nothing is re-laid-out on the GPU.
//
// Two regimes, one fragment convention (A operand = W rows, B operand = x rows, both K-contiguous, so
// every fragment is one 16-byte load; D[n_local][m_local], lane holds 4 consecutive n of one m):
//   * skinny  (decode, M <= 256): weight-bandwidth bound.  W and x chunks arrive by LDS-DMA into a
//     3-stage swizzled LDS ring (counted vmcnt, raw s_barrier), 8 waves x 16 weight rows per workgroup.
//     Split-K over workgroups fills the CUs when N/128 is small; fp32 partial slabs are combined by a
//     second tiny kernel that also applies the epilogue.
//   * tiled   (prefill, M in the thousands): MFMA bound.  128x128x64 tiles, LDS-DMA staging
//     (global_load_lds, 16 B/lane) with the XOR swizzle applied on the SOURCE address
//     (cdna_hip_programming.md rule 21), double-buffered, XCD-aware tile order.
#include <cstdio>
#include <cstring>
#include "lia_common.h"
#include "lia_epilogue.h"

#define GL_AS1(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_AS3(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int LIA_MAX_DEVICES = 64;
static inline int lia_current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= LIA_MAX_DEVICES) d = 0;
  return d;
}

// Combine split-K slabs [S][M][N] fp32 and apply the epilogue; one thread per 4 columns.
__global__ __launch_bounds__(256) void lia_splitk_reduce_kernel(const float* __restrict__ partial, int S, int M, int N,
                                                                 LiaEpilogue ep, LiaOutMap om) {
  long q = (long)blockIdx.x * 256 + threadIdx.x;
  long nq = (long)M * (N / 4);
  if (q >= nq) return;
  int m = (int)(q / (N / 4));
  int n = (int)(q - (long)m * (N / 4)) * 4;
  // the epilogue's operands and every slab value requested before the first add (r04: a slice per loop trip was S dependent
  // round trips, the bias / residual loads one more behind them); the sum is still slice 0 + 1 + ...
  const bool hb = ep.bias != nullptr, hr = ep.residual != nullptr;
  const uint2 bb = hb ? *(const uint2*)(ep.bias + n) : uint2{0u, 0u};
  const uint2 rr = hr ? *(const uint2*)(ep.residual + (long)m * ep.ldr + n) : uint2{0u, 0u};
  f32x4 a = *(const f32x4*)(partial + (long)m * N + n);
  for (int s0 = 1; s0 < S; s0 += 4) {
    f32x4 t[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) t[g] = *(const f32x4*)(partial + ((long)min(s0 + g, S - 1) * M + m) * N + n);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (s0 + g < S) a += t[g];
  }
  const f32x4 r = epilogue_quad_pre(a, bb, rr, hb, ep.relu, hr);
  *(uint2*)lia_out_ptr(om, m, n) = uint2{pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
}

// ---- split-K combines that also do the next op of the decode layer (LiaPost, lia_common.h) ----
// LAYERNORM / RMSNORM: one workgroup per output row; the row is combined, finished (bias / residual), stored, and normalised
// from registers into post.out -- the out-proj / fc2 (o / down) combine and the norm kernel behind it in one launch.
template <int KIND, int NV>
__global__ __launch_bounds__(LIA_ROW_THREADS) void lia_splitk_reduce_norm_kernel(const float* __restrict__ partial, int S, int M, int N,
                                                                      LiaEpilogue ep, LiaOutMap om, LiaPost post) {
  __shared__ float red[2 * LIA_ROW_WAVES];
  const int m = blockIdx.x, tid = threadIdx.x;
  const int nv = N >> 3;
  uint4 v[NV], gv[NV], bv[NV], eb[NV], er[NV];
  const bool hb = ep.bias != nullptr, hr = ep.residual != nullptr;
  // every slice's loads of this thread's pieces in flight together, GS slices at a time (r04: a slice per round trip is S
  // dependent L2 round trips -- 8 us for the 8 slabs of an OPT-30B fc2 row); the sum per element is still slice 0 + 1 + ...
  f32x4 acc[NV][2];
  const float* prow = partial + (long)m * N;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = min(tid + LIA_ROW_THREADS * k, nv - 1);          // (clamped: idle lanes re-read the last piece and drop it)
    gv[k] = *(const uint4*)(post.g + 8 * i);
    if (KIND == LIA_POST_LAYERNORM) bv[k] = *(const uint4*)(post.b + 8 * i);
    acc[k][0] = *(const f32x4*)(prow + 8 * i);
    acc[k][1] = *(const f32x4*)(prow + 8 * i + 4);
    eb[k] = hb ? *(const uint4*)(ep.bias + 8 * i) : uint4{0u, 0u, 0u, 0u};          // the epilogue's operands travel with the slabs
    er[k] = hr ? *(const uint4*)(ep.residual + (long)m * ep.ldr + 8 * i) : uint4{0u, 0u, 0u, 0u};
  }
  constexpr int GS = NV == 1 ? 7 : 4;
  for (int s0 = 1; s0 < S; s0 += GS) {
    f32x4 t[GS][NV][2];
#pragma unroll
    for (int g = 0; g < GS; ++g) {
      const float* ps = prow + (long)min(s0 + g, S - 1) * M * N;  // (clamped: a slice beyond the last is loaded again and dropped)
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = min(tid + LIA_ROW_THREADS * k, nv - 1);
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
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = tid + LIA_ROW_THREADS * k;
    v[k] = uint4{0u, 0u, 0u, 0u};
    if (i < nv) {
      const f32x4 lo = epilogue_quad_pre(acc[k][0], uint2{eb[k].x, eb[k].y}, uint2{er[k].x, er[k].y}, hb, ep.relu, hr);
      const f32x4 hi = epilogue_quad_pre(acc[k][1], uint2{eb[k].z, eb[k].w}, uint2{er[k].z, er[k].w}, hb, ep.relu, hr);
      v[k] = uint4{pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
      *(uint4*)(om.base[0] + (long)m * om.ld[0] + 8 * i) = v[k];
    }
  }
  bf16_t* yr = post.out + (long)m * post.ldo;
  if (KIND == LIA_POST_LAYERNORM) row_layernorm_block<NV>(v, gv, bv, nv, N, post.eps, yr, red);
  else row_rmsnorm_block<NV>(v, gv, nv, N, post.eps, yr, red);
}

// SILU_MUL: the gate | up projection's combine writes act = silu(gate) * up directly; the [M][2F] intermediate never exists.
__global__ __launch_bounds__(256) void lia_splitk_reduce_silu_kernel(const float* __restrict__ partial, int S, int M, int N,
                                                                      LiaEpilogue ep, LiaPost post) {
  const int F = N >> 1;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= (long)M * (F >> 2)) return;
  const int m = (int)(q / (F >> 2));
  const int c = (int)(q - (long)m * (F >> 2)) * 4;
  // column of gate value c and of its up partner: [gate | up] halves, or blocks of LIA_GU_BLOCK gate | LIA_GU_BLOCK up columns
  const int ng = post.gu_block ? (c / LIA_GU_BLOCK) * (2 * LIA_GU_BLOCK) + (c % LIA_GU_BLOCK) : c;
  const int nu = post.gu_block ? ng + LIA_GU_BLOCK : F + c;
  const f32x4 gq = epilogue_quad(splitk_sum(partial, S, M, N, m, ng), m, ng, ep);
  const f32x4 uq = epilogue_quad(splitk_sum(partial, S, M, N, m, nu), m, nu, ep);
  uint2 o;
  o.x = lia_silu_mul_pair(pack_bf16x2(gq[0], gq[1]), pack_bf16x2(uq[0], uq[1]));
  o.y = lia_silu_mul_pair(pack_bf16x2(gq[2], gq[3]), pack_bf16x2(uq[2], uq[3]));
  *(uint2*)(post.out + (long)m * post.ldo + c) = o;
}

// ROPE: the q | k | v projection's combine rotates the q and k heads (the first post.rot_heads heads of the row) on the way
// out; one thread per 4 pairs (i .. i+3, i+half .. i+half+3) of one head, the v heads pass through.
__global__ __launch_bounds__(256) void lia_splitk_reduce_rope_kernel(const float* __restrict__ partial, int S, int M, int N,
                                                                      LiaEpilogue ep, LiaOutMap om, LiaPost post) {
  const int hd = post.hd, half = hd >> 1, gph = half >> 2;          // groups of 4 pairs per head
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int heads = N / hd;
  if (idx >= (long)M * heads * gph) return;
  const int gi = (int)(idx % gph);
  const long mh = idx / gph;
  const int h = (int)(mh % heads), m = (int)(mh / heads);
  const int n0 = h * hd + 4 * gi, n1 = n0 + half;
  const f32x4 a = epilogue_quad(splitk_sum(partial, S, M, N, m, n0), m, n0, ep);
  const f32x4 b = epilogue_quad(splitk_sum(partial, S, M, N, m, n1), m, n1, ep);
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
  *(uint2*)lia_out_ptr(om, m, n0) = oa;
  *(uint2*)lia_out_ptr(om, m, n1) = ob;
}

// LDS slot (row, c) holds global 16-byte chunk (c ^ swz(row)) of that row; a 128-B row is half a
// 256-B bank row, so consecutive row pairs share a bank row and swz uses row>>1.
__device__ __forceinline__ int tl_swz(int row) { return (row >> 1) & 7; }

// ---------------------------------------------------------------------------------------------
// LDS-staged epilogue of the tiled kernels.  A wave's accumulators cover 64 columns n x (16*MB) rows m, the
// lane owning 4 consecutive n of one m: stored directly that is 32-byte fragments of 16 different rows per
// instruction.  Instead each wave parks its tile as bf16 [m][64 n] (128-byte rows, XOR-swizzled 16-byte
// chunks) in its own LDS region after matmul-rounding, bias and relu, then streams it out row-wise:
// 16 bytes per lane, whole 128-byte lines per row, the residual read the same way.  Rounding points are
// unchanged: bf16(acc) -> bf16(+bias) -> relu -> bf16(residual + .).
// ---------------------------------------------------------------------------------------------
enum { LIA_EF_BIAS = 1, LIA_EF_RELU = 2, LIA_EF_RESIDUAL = 4, LIA_EF_GLU = 8 };   // compile-time epilogue masks (EF below)
// (MBT, J0, MB: the m-blocks J0 .. J0+MB-1 of an accumulator array with MBT of them; m_base = first row of block J0)
// the residual rows of a wave tile of 16 MB rows x 64 columns, all requested at once (clamped addresses, no branch
// around a load: a load inside a per-row `if` is waited for on the spot -- 16 dependent round trips, ~13 us per tile)
template <int MB, int EF = -1>
__device__ __forceinline__ void epilogue_load_residual(uint4 (&rres)[2 * MB], int m_base, int n_base, int M, int N, const LiaEpilogue& ep,
                                                       int lane) {
  if (EF < 0 ? ep.residual == nullptr : !(EF & LIA_EF_RESIDUAL)) return;
  const int c = lane & 7;
#pragma unroll
  for (int r = 0; r < 2 * MB; ++r) {
    const int gm = min(m_base + r * 8 + (lane >> 3), M - 1), gn = min(n_base + c * 8, N - 8);
    rres[r] = *(const uint4*)(ep.residual + (long)gm * ep.ldr + gn);
  }
}

// Instruction count matters here: stamped on MI355X the epilogue of a 256 x 256 tile took 11 us with the chip otherwise
// idle -- VALU time, not memory (the second half, whose residual rows had long arrived, alone took 5 us).  So: no integer
// division per store (the output segment of a lane's 8 columns and the cache row of its first tile row are worked out
// once, rows then advance by 8), and the two roundings of a pair of values share one v_cvt_pk.
// EF: -1 = the epilogue's switches are read from `ep` at run time; otherwise a compile-time mask (LIA_EF_*) of what this launch's
// epilogue does -- the phased kernel is instantiated per mask: with run-time switches hipcc computes every variant of a value and
// selects (14 VALU instructions per output: 8.8 us per 256 x 256 tile without a residual, 12-13 us with one; tools/gemm_tile_stamps)
template <int MBT, int J0, int MB, int EF = -1>
__device__ __forceinline__ void epilogue_via_lds_part(const f32x4 (&acc)[4][MBT], const uint4 (&rres)[2 * MB], char* region, int m_base,
                                                      int n_base, int M, int N, const LiaEpilogue& ep, const LiaOutMap& om, int lane) {
  const int l15 = lane & 15, lq = lane >> 4;
  const bool hb = EF < 0 ? ep.bias != nullptr : (EF & LIA_EF_BIAS) != 0, hr = EF < 0 ? ep.residual != nullptr : (EF & LIA_EF_RESIDUAL) != 0;
  const bool relu = EF < 0 ? ep.relu != 0 : (EF & LIA_EF_RELU) != 0, glu = EF < 0 ? ep.glu != 0 : (EF & LIA_EF_GLU) != 0;
  const int c = lane & 7;
  // the four bias pieces of the lane are requested at once from clamped addresses: a load under `if (n < N)` is waited for on the
  // spot, and eight such round trips (two parts) were most of the 8-13 us this epilogue took per tile (tools/gemm_tile_stamps)
  uint2 bq[4] = {};
  if (hb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = *(const uint2*)(ep.bias + min(n_base + i * 16 + 4 * lq, N - 4));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    const int n = n_base + i * 16 + 4 * lq;
    if (hb && n < N) { b[0] = bf2f(bq[i].x & 0xffff); b[1] = bf2f(bq[i].x >> 16); b[2] = bf2f(bq[i].y & 0xffff); b[3] = bf2f(bq[i].y >> 16); }
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const int m = j * 16 + l15;
      uint32_t o2[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // bf16(acc) -> bf16(+bias) -> relu, two values per packed conversion (relu commutes with the rounding)
        uint32_t p = pack_bf16x2(acc[i][J0 + j][2 * h], acc[i][J0 + j][2 * h + 1]);
        float t0 = __uint_as_float(p << 16), t1 = __uint_as_float(p & 0xffff0000u);
        if (hb) { t0 += b[2 * h]; t1 += b[2 * h + 1]; }
        if (relu) { t0 = fmaxf(t0, 0.f); t1 = fmaxf(t1, 0.f); }
        o2[h] = (hb || relu) ? pack_bf16x2(t0, t1) : p;
      }
      const int chunk = (2 * i + (lq >> 1)) ^ (m & 7);
      *(uint2*)(region + m * 128 + chunk * 16 + (lq & 1) * 8) = uint2{o2[0], o2[1]};
    }
  }
  if (glu) {
    // gated-linear-unit output: the wave's 64 columns are 32 gate | 32 up columns of the same 32 outputs (the weight rows are
    // interleaved in blocks of LIA_GU_BLOCK).  Lanes c < 4 read their gate chunk c and the up chunk c + 4 of a row and store
    // silu(gate) * up -- the arithmetic of lia_silu_mul_kernel on the same bf16-rounded values -- as 8 of the N / 2 outputs.
    const int gn_o = (n_base >> 1) + c * 8;
    int gm = m_base + (lane >> 3);
#pragma unroll
    for (int r = 0; r < 2 * MB; ++r) {
      const int m = r * 8 + (lane >> 3);
      if (c < 4) {
        const uint4 gv = *(const uint4*)(region + m * 128 + ((c ^ (m & 7)) << 4));
        const uint4 uv = *(const uint4*)(region + m * 128 + (((c + 4) ^ (m & 7)) << 4));
        const uint4 o{lia_silu_mul_pair(gv.x, uv.x), lia_silu_mul_pair(gv.y, uv.y), lia_silu_mul_pair(gv.z, uv.z), lia_silu_mul_pair(gv.w, uv.w)};
        if (gm < M && gn_o < (N >> 1)) *(uint4*)(om.base[0] + (long)gm * om.ld[0] + gn_o) = o;
      }
      gm += 8;
    }
    return;
  }
  // where my 8 columns go: segment and column inside it (constant over the rows); the rows' offsets are worked out first (in cache
  // mode a division per part, then additions), then every LDS read is issued, then every store: interleaved per row the scalar
  // loads of the map's fields and the LDS round trip sat in front of each store
  const int gn = n_base + c * 8;
  const int seg = gn / om.seg_n;
  bf16_t* const obase = om.base[seg] + (gn - seg * om.seg_n);
  const long old = om.ld[seg];
  const bool cmode = om.cache_mode[seg] != 0;
  const int gm0 = m_base + (lane >> 3);
  long roff[2 * MB];
  if (cmode) {
    const int T_ = om.T, Bc_ = om.Bc;
    int cb = gm0 / T_, ct = gm0 - cb * T_;
    const long base_ = (long)om.pos0 * Bc_ + om.b0;
#pragma unroll
    for (int r = 0; r < 2 * MB; ++r) {
      roff[r] = (base_ + (long)ct * Bc_ + cb) * old;
      ct += 8;
      while (ct >= T_) { ct -= T_; ++cb; }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 2 * MB; ++r) roff[r] = (long)(gm0 + 8 * r) * old;
  }
  // same-wave LDS write -> read: program order + the compiler's lgkmcnt wait suffice (the region is private)
  uint4 v[2 * MB];
#pragma unroll
  for (int r = 0; r < 2 * MB; ++r) {
    const int m = r * 8 + (lane >> 3);
    v[r] = *(const uint4*)(region + m * 128 + ((c ^ (m & 7)) << 4));
  }
  if (hr) {
#pragma unroll
    for (int r = 0; r < 2 * MB; ++r) {
      const uint4 rr = rres[r];
      const uint32_t vw[4] = {v[r].x, v[r].y, v[r].z, v[r].w}, rw[4] = {rr.x, rr.y, rr.z, rr.w};
      uint32_t ow[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        ow[e] = pack_bf16x2(__uint_as_float(rw[e] << 16) + __uint_as_float(vw[e] << 16),
                            __uint_as_float(rw[e] & 0xffff0000u) + __uint_as_float(vw[e] & 0xffff0000u));
      v[r] = uint4{ow[0], ow[1], ow[2], ow[3]};
    }
  }
  if (m_base + 16 * MB <= M) {               // (wave-uniform) every row of the part exists: no per-row test
    if (gn < N) {
#pragma unroll
      for (int r = 0; r < 2 * MB; ++r) *(uint4*)(obase + roff[r]) = v[r];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 2 * MB; ++r)
      if (gm0 + 8 * r < M && gn < N) *(uint4*)(obase + roff[r]) = v[r];
  }
}

template <int MB>
__device__ __forceinline__ void epilogue_via_lds(const f32x4 (&acc)[4][MB], char* region, int m_base, int n_base, int M, int N,
                                                 const LiaEpilogue& ep, const LiaOutMap& om, int lane) {
  // the residual loads' latency hides behind the accumulator -> LDS pass
  uint4 rres[2 * MB];
  epilogue_load_residual<MB>(rres, m_base, n_base, M, N, ep, lane);
  epilogue_via_lds_part<MB, 0, MB>(acc, rres, region, m_base, n_base, M, N, ep, om, lane);
}

// ---------------------------------------------------------------------------------------------
// skinny regime (decode, M <= 256): weight-bandwidth bound.  Every operand byte arrives by LDS-DMA
// (global_load_lds, 16 B/lane, 128-B rows -> whole cache lines per request) into an S-stage LDS ring;
// S-1 chunks stay in flight per workgroup behind a COUNTED vmcnt and a raw s_barrier
// (cdna_hip_programming.md "Pipelining across barriers").  Workgroup = 8 waves = 128 weight rows x all M
// rows: the x chunk (L2-resident) is shared by 8 waves, so LDS-DMA moves 1.5 bytes per weight byte.
// Measured on MI355X (tools/gemm_bench.hip, cold weights, M = 64): 4.6-5.1 TB/s on the OPT-30B shapes with 128-row
// workgroups; RT = 2 (256 weight rows per workgroup, 1.25 LDS-DMA bytes per weight byte) 5.4-5.6 TB/s on qkv/fc1/fc2
// (fit: 6.7 TB/s asymptotic + 14 us fixed per launch), which the launcher picks when the coarser grid still fills the chip.
// History: a first version kept W in a 4-deep VGPR ring (plain loads) and staged x through registers;
// x and W then share one in-order vmcnt queue and the ring's depth collapses to one chunk (2.9 TB/s).
// A 64-row workgroup moves 2 LDS-DMA bytes per weight byte and saturates the CU's ~34 GB/s LDS-DMA path
// at 4.2 TB/s.
// ---------------------------------------------------------------------------------------------
constexpr int S2_BK = 64;     // K per chunk: 128-byte rows

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
  else if constexpr (N == 38) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
  else static_assert(N < 0, "add the immediate");
}

#ifdef LIA_GEMM_STAMPS
__device__ unsigned long long g_s2_stamps[8192 * 8];   // per workgroup: start, first chunk landed, after the K loop, stores issued, stores drained
#define S2_STAMP(i) do { if (threadIdx.x == 0) { const unsigned g_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; if (g_ < 8192) g_s2_stamps[g_ * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define S2_STAMP(i) do { } while (0)
#endif

template <int MT, int S, int NT, int WAVES, int RT = 1>
__global__ __launch_bounds__(64 * WAVES) void lia_gemm_skinny2_kernel(const bf16_t* __restrict__ x, long ldx,
                                                                       const bf16_t* __restrict__ W, long ldw, int M, int N,
                                                                       int K, int chunks_per_split,
                                                                       float* __restrict__ partial, LiaEpilogue ep, LiaOutMap om) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BN = 16 * WAVES * RT;               // W rows per workgroup (RT MFMA row tiles of 16 per wave)
  constexpr int RR = 8 * WAVES;                     // rows one LDS-DMA round of the workgroup covers (128 B each)
  constexpr int RB = RR * 128;                      // bytes per round
  constexpr int WL = BN / RR;                       // W rounds per chunk
  constexpr int XR = 16 * MT;                       // x rows in a stage
  constexpr int XL = (XR + RR - 1) / RR;            // x rounds per chunk
  constexpr int NL = WL + XL;                       // LDS-DMA instructions per thread per chunk
  constexpr int WTILE = BN * 128, STAGE = WTILE + XL * RB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int n_tile = blockIdx.x * BN;
  const int m_base = blockIdx.z * XR;               // grid.z > 1: the x rows are cut into blocks of 16 MT
  x += (long)m_base * ldx;
  const int Mloc = M - m_base;
  const int nchunks = K / S2_BK;
  const int c_begin = blockIdx.y * chunks_per_split;
  const int c_end = min(nchunks, c_begin + chunks_per_split);
  const int n = c_end - c_begin;
  S2_STAMP(0);

  // per-thread source rows (clamped: out-of-range rows re-read a valid one and are never stored)
  const int srow = tid >> 3, sc = tid & 7;
  const bf16_t* wsrc[WL];
#pragma unroll
  for (int r = 0; r < WL; ++r) {
    int row = srow + RR * r;
    wsrc[r] = W + (long)min(n_tile + row, N - 1) * ldw + ((sc ^ tl_swz(row)) << 3);
  }
  const bf16_t* xsrc[XL];
#pragma unroll
  for (int r = 0; r < XL; ++r) {
    int row = srow + RR * r;
    xsrc[r] = x + (long)min(min(row, XR - 1), Mloc - 1) * ldx + ((sc ^ tl_swz(row)) << 3);
  }
  auto issue = [&](int c, int stage) {
    char* st = smem + stage * STAGE;
    const long koff = (long)c * S2_BK;
#pragma unroll
    for (int r = 0; r < WL; ++r) {
      // weights are read once by one workgroup: non-temporal (aux = 2) keeps them from evicting x in L2
      if constexpr (NT) __builtin_amdgcn_global_load_lds(GL_AS1(wsrc[r] + koff), LDS_AS3(st + r * RB + wave * 1024), 16, 0, 2);
      else __builtin_amdgcn_global_load_lds(GL_AS1(wsrc[r] + koff), LDS_AS3(st + r * RB + wave * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < XL; ++r)
      __builtin_amdgcn_global_load_lds(GL_AS1(xsrc[r] + koff), LDS_AS3(st + WTILE + r * RB + wave * 1024), 16, 0, 0);
  };

  f32x4 acc[RT][MT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int p = 0; p < MT; ++p) acc[t][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (n > 0) {
    const int c_last = c_end - 1;
#pragma unroll
    for (int j = 0; j < S - 1; ++j) issue(min(c_begin + j, c_last), j);
    const int wrow = wave * 16 * RT + l15;
    for (int i = 0; i < n; ++i) {
      // retire chunk i (issued S-1 groups ago); the groups behind it stay in flight
      const int behind = min(S - 2, n - 1 - i);       // groups issued after chunk i that may still be pending
      if (behind >= S - 2) wait_vmcnt<(S - 2) * NL>();
      else if (S > 3 && behind == S - 3) wait_vmcnt<(S > 3 ? (S - 3) : 0) * NL>();
      else if (S > 4 && behind == S - 4) wait_vmcnt<(S > 4 ? (S - 4) : 0) * NL>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
#ifdef LIA_GEMM_STAMPS
      if (i == 0) S2_STAMP(1);
#endif
      if (i + S - 1 < n) issue(c_begin + i + S - 1, (i + S - 1) % S);
      const char* wt = smem + (i % S) * STAGE;
      const char* xt = wt + WTILE;
      // every fragment read of the chunk first, then the MFMAs (r04): left to itself hipcc alternates ds_read_b128 / s_waitcnt
      // lgkmcnt(0) / MFMA -- a dependent LDS round trip in front of every MFMA, ~1 us per chunk whatever the memory system does
      // (same products into the same accumulators in the same order: bit-identical)
      bf16x8 a[2][RT], bq[2][MT];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const int row = wrow + 16 * t;
          a[ks][t] = __builtin_bit_cast(bf16x8, *(const uint4*)(wt + row * 128 + (((4 * ks + lq) ^ tl_swz(row)) << 4)));
        }
#pragma unroll
        for (int p = 0; p < MT; ++p) {
          int row = 16 * p + l15;
          bq[ks][p] = __builtin_bit_cast(bf16x8, *(const uint4*)(xt + row * 128 + (((4 * ks + lq) ^ tl_swz(row)) << 4)));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p = 0; p < MT; ++p)
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks][t], bq[ks][p], acc[t][p], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // keep the last MFMA well clear of the accumulator reads below (see the note in v1)
  __builtin_amdgcn_s_barrier();
  S2_STAMP(2);

#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int n_wave = n_tile + (wave * RT + t) * 16;
    if (n_wave >= N) continue;
    const int nn = n_wave + 4 * lq;
    if (partial != nullptr) {
      float* pp = partial + (long)blockIdx.y * M * N;
#pragma unroll
      for (int p = 0; p < MT; ++p) {
        int m = m_base + 16 * p + l15;
        if (m < M) *(f32x4*)(pp + (long)m * N + nn) = acc[t][p];
      }
    } else if (ep.glu) {
      // gate | up projection, one slice: park the finished bf16 tile in the (dead) ring, pair the columns below
#pragma unroll
      for (int p = 0; p < MT; ++p) {
        const int ml = 16 * p + l15;
        const f32x4 q = epilogue_quad(acc[t][p], min(m_base + ml, M - 1), nn, ep);
        *(uint2*)(smem + ((long)ml * BN + (nn - n_tile)) * 2) = uint2{pack_bf16x2(q[0], q[1]), pack_bf16x2(q[2], q[3])};
      }
    } else {
#pragma unroll
      for (int p = 0; p < MT; ++p) {
        int m = m_base + 16 * p + l15;
        if (m < M) store_quad(acc[t][p], m, nn, ep, om);
      }
    }
  }
  if (partial == nullptr && ep.glu) {
    // LlamaMLP act_fn(gate) * up inside the GEMM launch (r03): the N columns are blocks of LIA_GU_BLOCK gate | LIA_GU_BLOCK up
    // columns (lia_llama_desc.gu_block), a 128-column tile holds both factors of 64 outputs.  Same device function as the
    // stand-alone kernel and the split-K combine (lia_silu_mul_pair): the three routes give the same bits for the same sums.
    __syncthreads();
    constexpr int OQ = BN / 8;                        // output quads per tile row (BN / 2 columns)
    for (int q = tid; q < XR * OQ; q += 64 * WAVES) {
      const int ml = q / OQ, c4 = (q % OQ) * 4;
      const int ng = (c4 / LIA_GU_BLOCK) * (2 * LIA_GU_BLOCK) + (c4 % LIA_GU_BLOCK);
      const int m = m_base + ml;
      if (m >= M || n_tile + ng + LIA_GU_BLOCK >= N) continue;        // (a ragged last tile: N is a multiple of 64 here)
      const uint2 g = *(const uint2*)(smem + ((long)ml * BN + ng) * 2);
      const uint2 u = *(const uint2*)(smem + ((long)ml * BN + ng + LIA_GU_BLOCK) * 2);
      *(uint2*)lia_out_ptr(om, m, n_tile / 2 + c4) = uint2{lia_silu_mul_pair(g.x, u.x), lia_silu_mul_pair(g.y, u.y)};
    }
    return;
  }
#ifdef LIA_GEMM_STAMPS
  S2_STAMP(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  S2_STAMP(4);
#endif
  // (r02-r04 could also combine the split-K slabs inside this launch -- the last-arriving slice of a tile, found by a ticket --;
  // bit-identical and measured SLOWER at every decode shape: the last arriver reads 3-8 x 64 KB alone while its CU's neighbours
  // still stream, and every slice pays an agent-scope release.  Removed in r05; the slabs are combined by the small second
  // kernel, which also runs the op behind the GEMM.)
}

// ---------------------------------------------------------------------------------------------
// tiled regime
// ---------------------------------------------------------------------------------------------
constexpr int TL_BM = 128, TL_BN = 128, TL_BK = 64;
constexpr int TL_TILE_BYTES = 128 * TL_BK * 2;  // one operand tile: 128 rows x 128 B


__device__ __forceinline__ void tl_stage(const bf16_t* __restrict__ g, long ld, int row0, int rows_valid, int k0,
                                         char* lds_tile, int tid) {
  const int wave = tid >> 6;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int row = r * 32 + (tid >> 3);
    int c = tid & 7;
    int grow = min(row0 + row, rows_valid - 1);
    const bf16_t* src = g + (long)grow * ld + k0 + ((c ^ tl_swz(row)) << 3);
    __builtin_amdgcn_global_load_lds(GL_AS1(src), LDS_AS3(lds_tile + r * 4096 + wave * 1024), 16, 0, 0);
  }
}

__global__ __launch_bounds__(256) void lia_gemm_tiled_kernel(const bf16_t* __restrict__ x, long ldx,
                                                              const bf16_t* __restrict__ W, long ldw, int M, int N, int K,
                                                              int tiles_m, int tiles_n, LiaEpilogue ep, LiaOutMap om) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][W tile | x tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int wn = wave & 1, wm = wave >> 1;

  // XCD-aware tile order: blocks that share an XCD (same blockIdx % 8) walk a contiguous run of
  // tiles, and runs sweep GM m-tiles per n-tile so the W panel and the x panels stay in that L2.
  const int nwg = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r8 = nwg & 7;
  const int lin = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
#ifdef LIA_GM
  constexpr int GM = LIA_GM;
#else
  constexpr int GM = 8;
#endif
  const int group = lin / (GM * tiles_n);
  const int first_m = group * GM;
  const int gsz = min(tiles_m - first_m, GM);
  const int in_g = lin - group * GM * tiles_n;
  const int tm = first_m + in_g % gsz, tn = in_g / gsz;
  const int m0 = tm * TL_BM, n0 = tn * TL_BN;

  f32x4 acc[4][4];  // [n-block][m-block]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / TL_BK;
  tl_stage(W, ldw, n0, N, 0, smem, tid);
  tl_stage(x, ldx, m0, M, 0, smem + TL_TILE_BYTES, tid);
  __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and publishes the tile
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    char* wt = smem + cur * 2 * TL_TILE_BYTES;
    char* xt = wt + TL_TILE_BYTES;
    if (kt + 1 < nk) {
      char* nw = smem + (cur ^ 1) * 2 * TL_TILE_BYTES;
      tl_stage(W, ldw, n0, N, (kt + 1) * TL_BK, nw, tid);
      tl_stage(x, ldx, m0, M, (kt + 1) * TL_BK, nw + TL_TILE_BYTES, tid);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int row = wn * 64 + i * 16 + l15;
        a[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(wt + row * 128 + (((4 * ks + lq) ^ tl_swz(row)) << 4)));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int row = wm * 64 + j * 16 + l15;
        b[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(xt + row * 128 + (((4 * ks + lq) ^ tl_swz(row)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    cur ^= 1;
  }

  epilogue_via_lds<4>(acc, smem + wave * 8192, m0 + wm * 64, n0 + wn * 64, M, N, ep, om, lane);
}

// ---------------------------------------------------------------------------------------------
// tiled regime, large: 256 x 256 x 64 tiles, 8 waves (2 along n x 4 along m... each wave 64 n x 128 m),
// one workgroup per CU.  Twice the flops per staged byte of the 128^2 tile: the 128^2 structure tops out
// near 0.9 PFLOP/s because its LDS-DMA traffic (64 flop/B) saturates the CU's load path first.
// LDS: 2 buffers x (W tile 32 KB + x tile 32 KB) = 128 KB.
// ---------------------------------------------------------------------------------------------
constexpr int T2_BM = 256, T2_BN = 256, T2_BK = 64;
constexpr int T2_TILE_BYTES = 256 * T2_BK * 2;  // one operand tile: 256 rows x 128 B = 32 KB

__device__ __forceinline__ void t2_stage(const bf16_t* __restrict__ g, long ld, int row0, int rows_valid, int k0,
                                         char* lds_tile, int tid) {
  const int wave = tid >> 6;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int row = r * 64 + (tid >> 3);
    int c = tid & 7;
    int grow = min(row0 + row, rows_valid - 1);
    const bf16_t* src = g + (long)grow * ld + k0 + ((c ^ tl_swz(row)) << 3);
    __builtin_amdgcn_global_load_lds(GL_AS1(src), LDS_AS3(lds_tile + r * 8192 + wave * 1024), 16, 0, 0);
  }
}

// XCD-aware tile order of the 256^2 kernels: blocks that share an XCD (same blockIdx % 8) walk a contiguous run of tiles, and runs
// sweep GM m-tiles per n-tile so the W panel and the x panels stay in that XCD's L2
#ifdef LIA_GM
constexpr int T2_GM = LIA_GM;
#else
constexpr int T2_GM = 4;   // m-tiles per XCD group: 2 / 4 / 8 / 16 / 32 measured, 4 is 1-4 % ahead of 8 on three of the four OPT-30B shapes
#endif
__device__ __forceinline__ void t2_tile_of_block(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int nwg = tiles_m * tiles_n;
  const int xcd = bid & 7, q = nwg >> 3, r8 = nwg & 7;
  const int lin = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  const int group = lin / (T2_GM * tiles_n);
  const int first_m = group * T2_GM;
  const int gsz = min(tiles_m - first_m, T2_GM);
  const int in_g = lin - group * T2_GM * tiles_n;
  tm = first_m + in_g % gsz;
  tn = in_g / gsz;
}

__global__ __launch_bounds__(512) void lia_gemm_tiled256_kernel(const bf16_t* __restrict__ x, long ldx,
                                                                 const bf16_t* __restrict__ W, long ldw, int M, int N, int K,
                                                                 int tiles_m, int tiles_n, LiaEpilogue ep, LiaOutMap om) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][W tile | x tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int wn = wave & 3, wm = wave >> 2;   // wave tile: n rows [64 wn, +64), m rows [128 wm, +128)

  int tm, tn;
  t2_tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * T2_BM, n0 = tn * T2_BN;

  f32x4 acc[4][8];  // [n-block][m-block]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / T2_BK;
  t2_stage(W, ldw, n0, N, 0, smem, tid);
  t2_stage(x, ldx, m0, M, 0, smem + T2_TILE_BYTES, tid);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    char* wt = smem + cur * 2 * T2_TILE_BYTES;
    char* xt = wt + T2_TILE_BYTES;
    if (kt + 1 < nk) {
      char* nw = smem + (cur ^ 1) * 2 * T2_TILE_BYTES;
      t2_stage(W, ldw, n0, N, (kt + 1) * T2_BK, nw, tid);
      t2_stage(x, ldx, m0, M, (kt + 1) * T2_BK, nw + T2_TILE_BYTES, tid);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[4], b[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int row = wn * 64 + i * 16 + l15;
        a[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(wt + row * 128 + (((4 * ks + lq) ^ tl_swz(row)) << 4)));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int row = wm * 128 + j * 16 + l15;
        b[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(xt + row * 128 + (((4 * ks + lq) ^ tl_swz(row)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    cur ^= 1;
  }

  // every wave is past its last LDS read (the loop's closing barrier): reuse the staging buffers
  epilogue_via_lds<8>(acc, smem + wave * 16384, m0 + wm * 128, n0 + wn * 64, M, N, ep, om, lane);
}

// ---------------------------------------------------------------------------------------------
// tiled regime, large, PHASED schedule (r02; r05: the one form that is launched -- the four-phase / global_load_lds / debug
// template forms it was selected from are in the git history, LABNOTES.md has their measurements).  Same 256 x 256 x 64 tile, LDS
// image ([W 256 rows | x 256 rows] x 128 B, XOR-swizzled chunks, two buffers = 128 KB), fragment convention and accumulation
// order (k ascending inside every accumulator: outputs are bit-identical to lia_gemm_tiled256_kernel) -- what changes is
// WHEN bytes move:
//   * the K-tile is staged as four half-tiles of 128 rows (W0 = W rows 0-127, W1, X0 = x rows 0-127, X1); a wave's 128 x rows
//     are 64 rows of X0 + 64 rows of X1 (m-blocks 0-3 / 4-7) and a K-tile is TWO phases of 32 MFMA:
//       PA(t): read all W + the X0 part (16)   stage X1(t+1) -> other buffer        vmcnt(8): X1(t) has landed      MFMA acc0
//       PB(t): read the X1 part (8)            stage X0, W0, W1 (t+2) -> this one   vmcnt(8): X0, W0, W1 (t+1)      MFMA acc1
//     so every half-tile has one full K-tile time between its LDS-DMA and the counted vmcnt that retires it, up to four (64 KB)
//     are in flight, and there is never a drain inside the loop (cdna_hip_programming.md "The 256^2 8-phase template"; the
//     one-barrier kernel above drains vmcnt(0) per K-tile and so exposes one full load latency, ~1.7 us, per 64-deep K-step);
//   * waves 0-3 (x rows wm = 0) and waves 4-7 run half a phase apart (one extra barrier for group 1): while one wave of a
//     SIMD issues its MFMAs the other reads its fragments and issues LDS-DMA;
//   * the LDS-DMA goes through buffer_load ... lds with a per-tile resource, a per-lane 32-bit row offset and the K offset in
//     an SGPR -- no 64-bit address arithmetic per piece (cdna_hip_programming.md T8).
// Ordering: RAW -- a half-tile is read one phase after the wait that retires it, and both groups pass their wait and a barrier
// in between; WAR -- a region is re-staged ONE phase after its last read, which is legal because the reading phase waits
// lgkmcnt(0) BEFORE its first barrier (group 1 lags half a phase).
// ---------------------------------------------------------------------------------------------
constexpr int T4_BUF_BYTES = 65536;   // one K-tile: W rows 0-255 at row * 128, x rows at 32768 + row * 128
#ifdef LIA_GEMM_STAMPS
__device__ unsigned long long g_t4_stamps[65536 * 8];   // per workgroup: start, first K-tile landed, after the K loop, epilogue issued, drained, hw id
#define T4_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 65536) g_t4_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define T4_STAMP(i) do { } while (0)
#endif

// SK (r06): the launch is split over K -- grid.y slices of `cps` K-tiles each (even, >= 4; the last slice takes what is left), every
// workgroup writes its fp32 accumulators to slab blockIdx.y of `partial` ([slices][M][N], the skinny regime's layout) and the
// epilogue runs in the combine kernel.  For 256 < M < ~2000 with N / 256 column tiles too few to fill 256 CUs (OPT-30B out-proj /
// fc2 at the reference's batch 900: 4 x 28 = 112 tiles).
template <int EF, bool SK = false>
__global__ __launch_bounds__(512) void lia_gemm_tiled256p_kernel(const bf16_t* __restrict__ x, long ldx,
                                                                  const bf16_t* __restrict__ W, long ldw, int M, int N, int K,
                                                                  int tiles_m, int tiles_n, LiaEpilogue ep, LiaOutMap om,
                                                                  float* __restrict__ partial, int cps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int wn = wave & 3, wm = wave >> 2;   // wave tile: W rows [64 wn, +64) x (x rows [64 wm, +64) and [128 + 64 wm, +64))

  T4_STAMP(0);
  int tm, tn;
  t2_tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * T2_BM, n0 = tn * T2_BN;

  f32x4 acc0[4][4], acc1[4][4];   // [W row-block][x row-block]: acc0 = the X0 part of the wave's x rows, acc1 = the X1 part
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc0[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // -- staging: a half-tile = two LDS-DMA rounds of the whole workgroup, 64 rows each; my lane moves chunk (tid & 7) of
  // row 64 q + (tid >> 3) of the 256-row operand tile, q = 2 half + round, into LDS slot tid & 7 of that row (linear
  // destination, swizzle on the source: rule 21).  Resources over the tile's first row; voffset = (clamped row - first row)
  // * ld * 2 + chunk * 16 (< 2^31: 256 rows)
  const int srow = tid >> 3;
  const int sck = ((tid & 7) ^ tl_swz(srow)) << 3;
  char* const sdst = smem + wave * 1024;
  const int kt0 = SK ? (int)blockIdx.y * cps : 0;           // first K-tile of this slice
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (long)n0 * ldw + (long)kt0 * T2_BK), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)m0 * ldx + (long)kt0 * T2_BK), 0, 0x7fffffff, 0x00020000);
  const int wv0 = (int)((min(n0 + srow, N - 1) - n0) * ldw + sck) * 2, wv1 = (int)((min(n0 + 64 + srow, N - 1) - n0) * ldw + sck) * 2;
  const int wv2 = (int)((min(n0 + 128 + srow, N - 1) - n0) * ldw + sck) * 2, wv3 = (int)((min(n0 + 192 + srow, N - 1) - n0) * ldw + sck) * 2;
  const int xv0 = (int)((min(m0 + srow, M - 1) - m0) * ldx + sck) * 2, xv1 = (int)((min(m0 + 64 + srow, M - 1) - m0) * ldx + sck) * 2;
  const int xv2 = (int)((min(m0 + 128 + srow, M - 1) - m0) * ldx + sck) * 2, xv3 = (int)((min(m0 + 192 + srow, M - 1) - m0) * ldx + sck) * 2;
#define T4_PIECE(rs, vo, t, dst) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_AS3(dst), 16, vo, (int)(t) * (T2_BK * 2), 0, 0)
#define T4_STAGE_W0(t, buf) do { T4_PIECE(rsw, wv0, t, sdst + (buf) * T4_BUF_BYTES);         T4_PIECE(rsw, wv1, t, sdst + (buf) * T4_BUF_BYTES + 8192); } while (0)
#define T4_STAGE_W1(t, buf) do { T4_PIECE(rsw, wv2, t, sdst + (buf) * T4_BUF_BYTES + 16384); T4_PIECE(rsw, wv3, t, sdst + (buf) * T4_BUF_BYTES + 24576); } while (0)
#define T4_STAGE_X0(t, buf) do { T4_PIECE(rsx, xv0, t, sdst + (buf) * T4_BUF_BYTES + 32768); T4_PIECE(rsx, xv1, t, sdst + (buf) * T4_BUF_BYTES + 40960); } while (0)
#define T4_STAGE_X1(t, buf) do { T4_PIECE(rsx, xv2, t, sdst + (buf) * T4_BUF_BYTES + 49152); T4_PIECE(rsx, xv3, t, sdst + (buf) * T4_BUF_BYTES + 57344); } while (0)

  // -- fragment reads: lane (l15, lq) reads chunk 4 ks + lq of row base + l15, stored in slot chunk ^ swz(row);
  // swz(row) = (row >> 1) & 7 depends on l15 only (row bases are multiples of 16), and the two k-steps differ in bit 6
  const int c0 = (lq ^ tl_swz(l15)) << 4;
  const char* const wf0 = smem + (wn * 64 + l15) * 128 + c0;               // k-step 0; k-step 1 = ^ 64
  const char* const wf1 = smem + (wn * 64 + l15) * 128 + (c0 ^ 64);
  const char* const xf0 = smem + 32768 + (wm * 64 + l15) * 128 + c0;
  const char* const xf1 = smem + 32768 + (wm * 64 + l15) * 128 + (c0 ^ 64);
  bf16x8 a[4][2], b[4][2];
#define T4_LD(p) __builtin_bit_cast(bf16x8, *(const uint4*)(p))
#define T4_READ_W(buf, i0) do {                                                                                          \
    a[i0][0] = T4_LD(wf0 + (buf) * T4_BUF_BYTES + (i0) * 2048);       a[i0][1] = T4_LD(wf1 + (buf) * T4_BUF_BYTES + (i0) * 2048);             \
    a[i0 + 1][0] = T4_LD(wf0 + (buf) * T4_BUF_BYTES + (i0 + 1) * 2048); a[i0 + 1][1] = T4_LD(wf1 + (buf) * T4_BUF_BYTES + (i0 + 1) * 2048);   \
  } while (0)
#define T4_READ_X(buf, part) do {                                                                                        \
    b[0][0] = T4_LD(xf0 + (buf) * T4_BUF_BYTES + (part) * 16384);        b[0][1] = T4_LD(xf1 + (buf) * T4_BUF_BYTES + (part) * 16384);        \
    b[1][0] = T4_LD(xf0 + (buf) * T4_BUF_BYTES + (part) * 16384 + 2048); b[1][1] = T4_LD(xf1 + (buf) * T4_BUF_BYTES + (part) * 16384 + 2048); \
    b[2][0] = T4_LD(xf0 + (buf) * T4_BUF_BYTES + (part) * 16384 + 4096); b[2][1] = T4_LD(xf1 + (buf) * T4_BUF_BYTES + (part) * 16384 + 4096); \
    b[3][0] = T4_LD(xf0 + (buf) * T4_BUF_BYTES + (part) * 16384 + 6144); b[3][1] = T4_LD(xf1 + (buf) * T4_BUF_BYTES + (part) * 16384 + 6144); \
  } while (0)
#define T4_MMA(ACC, i0) do {                                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                                        \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_)                                                                   \
      _Pragma("unroll") for (int i_ = (i0); i_ < (i0) + 2; ++i_)                                                          \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                                  \
          ACC[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i_][ks_], b[j_][ks_], ACC[i_][j_], 0, 0, 0);            \
    __builtin_amdgcn_s_setprio(0);                                                                                        \
  } while (0)
#define T4_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define T4_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
  // one K-tile.  S12: tile t+1 exists (stage its X1); S34: tile t+2 exists (stage its X0, W0, W1); W2 / W4: the vmcnt
  // immediates of the two waits (-1 = no wait); LAST: group 1 skips the closing barrier (it entered one barrier late).
  // W1 is staged with X0 and W0 in PB, one phase after the last read of those regions in PA (with W1 issued in PA it had ONE
  // phase of flight and the kernel ran 20 % slower when the waits were tightened further).
#define T4_TILE2(t, buf, S12, S34, W2, W4, LAST) do {                                                                    \
    T4_READ_W(buf, 0); T4_READ_W(buf, 2); T4_READ_X(buf, 0);                                                              \
    if (S12) T4_STAGE_X1((t) + 1, (buf) ^ 1);                                                                             \
    wait_vmcnt<W2>();                                                                                                     \
    T4_LGKM0(); T4_BARRIER(); T4_MMA(acc0, 0); T4_MMA(acc0, 2); T4_BARRIER();                                             \
    T4_READ_X(buf, 1);                                                                                                    \
    if (S34) { T4_STAGE_X0((t) + 2, buf); T4_STAGE_W0((t) + 2, buf); T4_STAGE_W1((t) + 2, buf); }                         \
    if (W4 >= 0) wait_vmcnt<(W4 >= 0 ? W4 : 0)>();                                                                        \
    T4_LGKM0(); T4_BARRIER(); T4_MMA(acc1, 2); T4_MMA(acc1, 0);                                                           \
    if (!(LAST) || wm == 0) T4_BARRIER();                                                                                 \
  } while (0)

  const int nk = SK ? min(cps, K / T2_BK - kt0) : K / T2_BK;      // even and >= 4 (the launcher falls back to the one-barrier kernel otherwise)
  T4_STAGE_X0(0, 0); T4_STAGE_W0(0, 0); T4_STAGE_W1(0, 0); T4_STAGE_X1(0, 0);
  T4_STAGE_X0(1, 1); T4_STAGE_W0(1, 1); T4_STAGE_W1(1, 1);
  wait_vmcnt<8>();             // X0, W0, W1 of tile 0 have landed (X1(0) is waited for in PA)
  T4_BARRIER();
  T4_STAMP(1);
  if (wm == 1) T4_BARRIER();     // the stagger
  int t = 0;
  for (; t + 2 < nk; t += 2) {
    T4_TILE2(t, 0, true, true, 8, 8, false);
    T4_TILE2(t + 1, 1, true, true, 8, 8, false);
  }
  T4_TILE2(t, 0, true, false, 8, 2, false);
  T4_TILE2(t + 1, 1, false, false, 0, -1, true);
#undef T4_TILE2
#undef T4_LGKM0
#undef T4_BARRIER
#undef T4_MMA
#undef T4_READ_X
#undef T4_READ_W
#undef T4_LD
#undef T4_STAGE_X1
#undef T4_STAGE_X0
#undef T4_STAGE_W1
#undef T4_STAGE_W0
#undef T4_PIECE
  // group 0's closing barrier is the one group 1 crossed before its last MFMAs: every LDS read of the workgroup has been
  // retired (PB's reads were waited for), no LDS-DMA is pending (the last wait was vmcnt(0)), and each wave's epilogue region
  // is its own.  The residual rows of BOTH halves of the wave tile are requested before the first store: vmcnt retires in
  // order, so a residual load issued behind the first half's stores would wait for them as well (the fragment registers are
  // free now)
  T4_STAMP(2);
  if constexpr (SK) {
    // the slice's partial sums, fp32, as they stand in the accumulators: lane (l15, lq) holds 4 consecutive n of row l15 of each block
    float* const ps = partial + (long)blockIdx.y * M * N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wn * 64 + i * 16 + 4 * lq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + l15;
        if (n < N && m < M) *(f32x4*)(ps + (long)m * N + n) = acc0[i][j];
        if (n < N && m + 128 < M) *(f32x4*)(ps + (long)(m + 128) * N + n) = acc1[i][j];
      }
    }
    return;
  }
  uint4 rres0[8], rres1[8];
  epilogue_load_residual<4, EF>(rres0, m0 + wm * 64, n0 + wn * 64, M, N, ep, lane);
  epilogue_load_residual<4, EF>(rres1, m0 + 128 + wm * 64, n0 + wn * 64, M, N, ep, lane);
  epilogue_via_lds_part<4, 0, 4, EF>(acc0, rres0, smem + wave * 16384, m0 + wm * 64, n0 + wn * 64, M, N, ep, om, lane);
  epilogue_via_lds_part<4, 0, 4, EF>(acc1, rres1, smem + wave * 16384 + 8192, m0 + 128 + wm * 64, n0 + wn * 64, M, N, ep, om, lane);
#ifdef LIA_GEMM_STAMPS
  T4_STAMP(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  T4_STAMP(4);
  if (threadIdx.x == 0 && blockIdx.x < 65536)
    g_t4_stamps[blockIdx.x * 8 + 5] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);   // XCC_ID, HW_ID
#endif
}

// ---------------------------------------------------------------------------------------------
// host launcher
// ---------------------------------------------------------------------------------------------
extern "C" size_t lia_gemm_workspace_bytes(int M, int N) {
  // worst case split-K = 8 fp32 slabs of a skinny problem; 256 < M < 2048 may split the phased tiled kernel up to 4 ways
  if (M > 256) return M < 2048 ? (size_t)4 * M * N * sizeof(float) : 0;
  return (size_t)8 * M * N * sizeof(float);
}

template <int MT, int S, int NT, int WAVES, int RT = 1>
static void launch_skinny2(const bf16_t* x, long ldx, const bf16_t* W, long ldw, int M, int N, int K, int split, int cps,
                           float* partial, const LiaEpilogue& ep, const LiaOutMap& om, hipStream_t st) {
  constexpr int BN = 16 * WAVES * RT, RR = 8 * WAVES;
  constexpr int XL = (16 * MT + RR - 1) / RR;
  dim3 grid((N + BN - 1) / BN, split, (M + 16 * MT - 1) / (16 * MT));
  size_t lds = (size_t)S * (BN * 128 + XL * RR * 128);
  // (an idempotent driver call, not a setting; per DEVICE -- the attribute belongs to the function on one device, and a second GPU in
  // the same process would otherwise never get the LDS opt-in: r05 verdict, hygiene)
  static bool attr_set[LIA_MAX_DEVICES] = {};
  const int dev = lia_current_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)lia_gemm_skinny2_kernel<MT, S, NT, WAVES, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((lia_gemm_skinny2_kernel<MT, S, NT, WAVES, RT>), grid, dim3(64 * WAVES), lds, st, x, ldx, W, ldw, M, N, K, cps,
                     split > 1 ? partial : nullptr, ep, om);
}

// the combine kernel that also runs `post`, if this shape has one; false: nothing launched
static bool launch_fused_combine(const float* ws, int split, int M, int N, const LiaEpilogue& ep, const LiaOutMap& om, const LiaPost& post,
                                 LiaGemmOpts* opts, hipStream_t st) {
  if (opts && !opts->fuse_combine) return false;
  if (post.kind == LIA_POST_LAYERNORM || post.kind == LIA_POST_RMSNORM) {
    if (om.seg_n != N || om.cache_mode[0] || (N & 7) || (N >> 3) > LIA_ROW_THREADS * 2 || (om.ld[0] & 7) || (post.ldo & 7) || !post.g || !post.out) return false;
    if (post.kind == LIA_POST_LAYERNORM && !post.b) return false;
#define LIA_NORM_COMBINE(K)                                                                                                            \
    do {                                                                                                                               \
      if ((N >> 3) <= LIA_ROW_THREADS) hipLaunchKernelGGL((lia_splitk_reduce_norm_kernel<K, 1>), dim3(M), dim3(LIA_ROW_THREADS), 0, st, ws, split, M, N, ep, om, post); \
      else hipLaunchKernelGGL((lia_splitk_reduce_norm_kernel<K, 2>), dim3(M), dim3(LIA_ROW_THREADS), 0, st, ws, split, M, N, ep, om, post);                              \
    } while (0)
    if (post.kind == LIA_POST_LAYERNORM) LIA_NORM_COMBINE(LIA_POST_LAYERNORM); else LIA_NORM_COMBINE(LIA_POST_RMSNORM);
#undef LIA_NORM_COMBINE
    if (opts) ++opts->fused_combines[post.kind];
    return true;
  }
  if (post.kind == LIA_POST_SILU_MUL) {
    if ((N & 7) || (post.ldo & 3) || !post.out || (post.gu_block && (post.gu_block != LIA_GU_BLOCK || (N % (2 * LIA_GU_BLOCK))))) return false;
    const long nq = (long)M * (N >> 3);
    hipLaunchKernelGGL(lia_splitk_reduce_silu_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, ws, split, M, N, ep, post);
    if (opts) ++opts->fused_combines[post.kind];
    return true;
  }
  if (post.kind == LIA_POST_ROPE) {
    if (post.hd <= 0 || (post.hd & 7) || N % post.hd || om.seg_n % post.hd || post.T <= 0 || !post.cos_t || !post.sin_t) return false;
    const long nt = (long)M * (N / post.hd) * (post.hd >> 3);
    hipLaunchKernelGGL(lia_splitk_reduce_rope_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, ws, split, M, N, ep, om, post);
    if (opts) ++opts->fused_combines[post.kind];
    return true;
  }
  return false;
}

// Returns 0 on success, -1 on unsupported shape.  workspace is only touched when split-K is chosen.
// opts (nullable = defaults): the per-context switches and counters (LiaGemmOpts, lia_common.h; lia_ctx_set_option) -- the
// library keeps no process-wide setting.
extern "C" int lia_gemm_launch(const bf16_t* x, long ldx, const bf16_t* W, long ldw, int M, int N, int K,
                               const LiaEpilogue* ep, const LiaOutMap* om, float* workspace, size_t workspace_bytes,
                               LiaGemmOpts* opts, int force_split, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, int* regime,
                               const LiaPost* post, int* post_done) {
  // post (nullable): the op that follows this GEMM in the decode layer; when the GEMM is split over K its combine kernel does
  // that op too and *post_done = 1 -- otherwise *post_done = 0 and the caller launches the stand-alone kernel.
  if (post_done) *post_done = 0;
  // ev0/ev1 (nullable): recorded on `st` immediately around the MAIN kernel launch only (bench.py's live
  // roofline timing; the split-K combine kernel is outside the bracket).  *regime: 1 skinny, 2 tiled.
  if (regime) *regime = 0;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if ((N % 16) != 0 || (om->seg_n % 4) != 0) return -1;
  const bool fuse = !opts || opts->fuse_combine;
  if (M <= 256 && (K % (2 * S2_BK)) == 0) {
    constexpr int WAVES = 8;
    const int nchunks = K / S2_BK;
    // Two workgroup shapes.  RT = 1: 128 weight rows, LDS-DMA moves (128 + 16 MT)/128 bytes per weight byte, two
    // workgroups per CU up to M = 64.  RT = 2 (32 < M <= 128): 256 weight rows per workgroup halve the x share of
    // the LDS-DMA traffic (OPT-30B fc1 at M = 64: 4.9 -> 5.5 TB/s) but leave a quarter as many workgroups, so it is
    // chosen only when tiles x slices still fill the 256 CUs and every slice keeps >= 32 chunks of pipeline.
    int rt = 1, split = 1;
    {
      const int tiles2 = (N + 255) / 256;
      int split2 = (256 + tiles2 / 2) / tiles2;
      split2 = split2 < 1 ? 1 : (split2 > 8 ? 8 : split2);
      while (split2 > 1 && nchunks / split2 < 32) --split2;
      const double waves2 = tiles2 * split2 / 256.0;
      const double fill2 = waves2 / (double)(int)(waves2 + 0.999999);
      const bool fits = (size_t)split2 * M * N * sizeof(float) <= workspace_bytes || split2 == 1;
      if (M > 32 && M <= 128 && fill2 >= 0.85 && nchunks / split2 >= 32 && fits && force_split <= 0) { rt = 2; split = split2; }
    }
    // 64 < M <= 128 with 128-row workgroups: cut the x rows into two blocks of 64 (grid.z = 2, MT = 4).  Each
    // workgroup then stages half the x bytes, two fit a CU, and the second read of a weight tile comes from L2 / the
    // Infinity Cache.  Worth it when the one-block grid cannot fill the chip (Llama-3-8B q/k/v, o at B = 128).
    bool mcut = false;
    if (rt == 1) {
      const int BN = 16 * WAVES;
      int tiles = (N + BN - 1) / BN;
      // ... unless K is long: with >= 24 chunks per slice at eight slices the one-block workgroups (every weight byte read once)
      // finish sooner than the two-block ones (Llama-3-8B down-proj, K = 14336, B = 128: 41.4 -> 37.7 us with the combine)
      const bool long_k = M > 64 && M <= 128 && force_split <= 0 && tiles * 8 <= 256 && nchunks / 8 >= 24 &&
                          (size_t)8 * M * N * sizeof(float) <= workspace_bytes;
      if (long_k) force_split = 8;
      if (M > 64 && M <= 128 && force_split <= 0 && tiles * 4 <= 256) {
        mcut = true;
        tiles *= 2;
      }
      // M <= 64: 3 stages x 24 KB -> two 8-wave workgroups per CU = 512 slots on the chip
      const int slots = (M <= 64 || mcut) ? 512 : 256;
      if (force_split > 0) {
        split = force_split;
      } else {
        // fill the slots once; beyond 4 slices the fp32 slab traffic (2 x M x BN x 4 B per workgroup) costs more
        // than the idle CUs it recovers (measured on OPT-30B shapes, tools/gemm_bench.hip)
        split = (slots + tiles / 2) / tiles;
        if (tiles * 10 > slots * 6 && tiles < slots) split = 3;   // 0.6..1 waves: three slices balance better than one
        split = split < 1 ? 1 : (split > 4 ? 4 : split);
        while (split > 1 && nchunks / split < 8) --split;
      }
    }
    // r03: a gate | up projection (LIA_POST_SILU_MUL over interleaved rows) whose 128-row tiles alone fill >= 3/4 of the chip runs as
    // ONE slice of 128-row workgroups: no fp32 slabs (29 MB written and read back for Llama-3-8B at B = 128), no combine launch, and
    // the kernel's epilogue pairs the columns itself -- 72 -> 64 us per launch against two 256-row slices + the fused combine
    // (tools/gemm_bench, cold weights).  The same split is used when the fusion is switched off (plain store, then the stand-alone
    // SiLU kernel), so both routes add the same products in the same order.
    LiaEpilogue ep_s = *ep;
    LiaOutMap om_s = *om;
    const bool glu_shape = post && post->kind == LIA_POST_SILU_MUL && post->gu_block == LIA_GU_BLOCK && (N % (2 * LIA_GU_BLOCK)) == 0 &&
                           M > 64 && M <= 128 && force_split <= 0 && (N + 127) / 128 >= 192 && !ep->residual;
    if (glu_shape) {
      rt = 1; mcut = false; split = 1;
      if (post_done && fuse && post->out && (post->ldo & 3) == 0) {
        ep_s.glu = 1;
        memset(&om_s, 0, sizeof(om_s));
        om_s.base[0] = post->out; om_s.ld[0] = post->ldo; om_s.seg_n = N / 2; om_s.T = 1;
        *post_done = 1;
        if (opts) ++opts->fused_combines[LIA_POST_SILU_MUL];
      }
      ep = &ep_s;
      om = &om_s;
    }
    if (split > nchunks) split = nchunks;
    if (split > 1 && (size_t)split * M * N * sizeof(float) > workspace_bytes) split = 1;
    int cps = (nchunks + split - 1) / split;
    split = (nchunks + cps - 1) / cps;
    if (regime) *regime = 1;
    if (ev0) (void)hipEventRecord(ev0, st);
    if (M <= 16) launch_skinny2<1, 3, 1, WAVES>(x, ldx, W, ldw, M, N, K, split, cps, workspace, *ep, *om, st);
    else if (M <= 32) launch_skinny2<2, 3, 1, WAVES>(x, ldx, W, ldw, M, N, K, split, cps, workspace, *ep, *om, st);
    else if (M <= 64) {
      if (rt == 2) launch_skinny2<4, 3, 1, WAVES, 2>(x, ldx, W, ldw, M, N, K, split, cps, workspace, *ep, *om, st);
      else launch_skinny2<4, 3, 1, WAVES>(x, ldx, W, ldw, M, N, K, split, cps, workspace, *ep, *om, st);
    } else if (M <= 128) {
      if (rt == 2) launch_skinny2<8, 3, 1, WAVES, 2>(x, ldx, W, ldw, M, N, K, split, cps, workspace, *ep, *om, st);
      else if (mcut) launch_skinny2<4, 3, 0, WAVES>(x, ldx, W, ldw, M, N, K, split, cps, workspace, *ep, *om, st);   // default cache policy: the other x block re-reads W
      else launch_skinny2<8, 3, 1, WAVES>(x, ldx, W, ldw, M, N, K, split, cps, workspace, *ep, *om, st);
    }
    else launch_skinny2<16, 3, 1, WAVES>(x, ldx, W, ldw, M, N, K, split, cps, workspace, *ep, *om, st);
    if (ev1) (void)hipEventRecord(ev1, st);
    if (split > 1) {
      if (post && post_done && launch_fused_combine(workspace, split, M, N, *ep, *om, *post, opts, st)) { *post_done = 1; return 0; }
      long nq = (long)M * (N / 4);
      hipLaunchKernelGGL(lia_splitk_reduce_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, workspace,
                         split, M, N, *ep, *om);
    }
    return 0;
  }
  if ((K % TL_BK) != 0) return -1;
  if (regime) *regime = 2;
  // the tiled kernels' epilogue writes silu(gate) * up itself when the weight rows are interleaved (LiaEpilogue::glu): the
  // [M, 2F] intermediate of a prefill (7.5 GB per Llama-3-8B layer at B 128 x T 1024) is never written nor read back
  LiaEpilogue ep_t = *ep;
  LiaOutMap om_t = *om;
  if (post && post_done && fuse && post->kind == LIA_POST_SILU_MUL && post->gu_block == LIA_GU_BLOCK &&
      (N % (2 * LIA_GU_BLOCK)) == 0 && post->out && (post->ldo & 7) == 0 && !ep->residual) {
    ep_t.glu = 1;
    memset(&om_t, 0, sizeof(om_t));
    om_t.base[0] = post->out; om_t.ld[0] = post->ldo; om_t.seg_n = N / 2; om_t.T = 1;
    *post_done = 1;
    if (opts) ++opts->fused_combines[LIA_POST_SILU_MUL];
  }
  ep = &ep_t;
  om = &om_t;
#ifdef LIA_MIDM_OLD
  if (M >= 1024 && N >= 512) {          // (tools/gemm_bench -DLIA_MIDM_OLD: r05's dispatch, the 128 x 128 kernel below M = 1024)
#else
  if (M > 256 && N >= 512) {
#endif
    // 256 x 256 tiles: the phased kernel, or -- K / 64 odd or < 4 -- the one-barrier-per-K-tile kernel (same bits).  r06: also for
    // 256 < M < 1024 (r01's 128 x 128 kernel served that range until now: the reference's batch-900 lines put every decode GEMM there)
    const int tiles_m = (M + T2_BM - 1) / T2_BM, tiles_n = (N + T2_BN - 1) / T2_BN;
    const int nkt = K / T2_BK;
    const bool phased = nkt >= 4 && (nkt & 1) == 0;
    // split over K where the tiles alone leave CUs idle: a cost model in microseconds -- rounds of 256 workgroups x (K-tiles of a
    // slice x 1.5 us + 12 us of prologue / epilogue) + the slabs written and read back at ~4 TB/s (tools/gemm_bench, M = 900)
    int split = 1, cps = nkt;
    if (phased && tiles_m * tiles_n < 256 && M < 2048 && !ep->glu && force_split >= 0) {
      double best = 1e30;
      for (int sp = (force_split > 0 ? force_split : 1); sp <= (force_split > 0 ? force_split : 4); ++sp) {
        int c = (nkt + sp - 1) / sp;
        c += c & 1;                                               // even slices
        const int ns = (nkt + c - 1) / c;                         // slices that really exist
        if (c < 4 || nkt - (ns - 1) * c < 4) continue;            // (the last one keeps >= 4 K-tiles; nkt and c even -> it is even)
        if (ns > 1 && (size_t)ns * M * N * sizeof(float) > workspace_bytes) continue;
        const int rounds = (tiles_m * tiles_n * ns + 255) / 256;
        const double t = rounds * (c * 1.5 + 12.0) + (ns > 1 ? (double)ns * M * N * 8.0 / 4e6 : 0.0);
        if (t < best) { best = t; split = ns; cps = c; }
      }
    }
    static bool attr_set[LIA_MAX_DEVICES] = {};
    const int dev = lia_current_device();
    if (!attr_set[dev]) {
#define LIA_T4_EACH(X) X(0) X(LIA_EF_BIAS) X(LIA_EF_BIAS | LIA_EF_RELU) X(LIA_EF_BIAS | LIA_EF_RESIDUAL) X(LIA_EF_RESIDUAL) X(LIA_EF_GLU) X(-1)
#define LIA_T4_ATTR(EF) (void)hipFuncSetAttribute((const void*)lia_gemm_tiled256p_kernel<EF>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * T4_BUF_BYTES);
      LIA_T4_EACH(LIA_T4_ATTR)
#undef LIA_T4_ATTR
      (void)hipFuncSetAttribute((const void*)lia_gemm_tiled256p_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * T4_BUF_BYTES);
      (void)hipFuncSetAttribute((const void*)lia_gemm_tiled256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * T2_TILE_BYTES);
      attr_set[dev] = true;
    }
    // M <= 384: three 128-row tiles waste fewer rows than two 256-row ones, and the 128 x 128 kernel (two workgroups per CU: 512
    // slots, ~1.12 us per K-tile + 8) wins where its grid fills them -- OPT-30B q|k|v at M = 300: 133 against 147 us; it loses
    // everywhere else (out-proj 115 against 64 us, fc1 247 / 188, fc2 415 / 228: tools/gemm_bench, r06)
    bool small_tiles = false;
    if (phased && M <= 384 && force_split <= 0) {
      const int t128 = ((M + TL_BM - 1) / TL_BM) * ((N + TL_BN - 1) / TL_BN);
      const double fill256 = (double)(tiles_m * tiles_n * split) / (256.0 * ((tiles_m * tiles_n * split + 255) / 256));
      const double est256 = ((tiles_m * tiles_n * split + 255) / 256) * (cps * (fill256 < 0.7 ? 1.25 : 1.5) + 12.0) +
                            (split > 1 ? (double)split * M * N * 8.0 / 4e6 : 0.0);
      const double est128 = ((t128 + 511) / 512) * (nkt * 1.12 + 8.0);
      small_tiles = est128 < est256;
    }
    if (!small_tiles) {
    if (ev0) (void)hipEventRecord(ev0, st);
    if (phased && split > 1) {
      hipLaunchKernelGGL((lia_gemm_tiled256p_kernel<0, true>), dim3(tiles_m * tiles_n, split), dim3(512), 2 * T4_BUF_BYTES, st, x, ldx, W, ldw, M, N, K,
                         tiles_m, tiles_n, *ep, *om, workspace, cps);
      if (ev1) (void)hipEventRecord(ev1, st);
      // the norms stay kernels of their own here: beyond the row-kernel's row limit the stand-alone LayerNorm / RMSNorm sums a row in
      // another order than the combine's row block does, and a prefill must not depend on the fuse switch (tests/test_gpu_fused_combine.py);
      // the elementwise posts (SiLU * up, RoPE) are the same arithmetic either way
      if (post && post_done && !*post_done && post->kind != LIA_POST_LAYERNORM && post->kind != LIA_POST_RMSNORM &&
          launch_fused_combine(workspace, split, M, N, *ep, *om, *post, opts, st)) { *post_done = 1; return 0; }
      const long nq = (long)M * (N / 4);
      hipLaunchKernelGGL(lia_splitk_reduce_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, workspace, split, M, N, *ep, *om);
      return 0;
    }
    if (phased) {
      // the epilogue's switches as a compile-time mask: the layers' own combinations have an instantiation each, anything else runs
      // the run-time form (same arithmetic)
      const int mask = (ep->bias ? LIA_EF_BIAS : 0) | (ep->relu ? LIA_EF_RELU : 0) | (ep->residual ? LIA_EF_RESIDUAL : 0) | (ep->glu ? LIA_EF_GLU : 0);
      bool done = false;
#define LIA_T4_LAUNCH(EF)                                                                                                             \
      if (!done && ((EF) == -1 || mask == (EF))) {                                                                                        \
        hipLaunchKernelGGL(lia_gemm_tiled256p_kernel<EF>, dim3(tiles_m * tiles_n), dim3(512), 2 * T4_BUF_BYTES, st, x, ldx, W, ldw, M, N, K, tiles_m, tiles_n, *ep, *om, \
                           (float*)nullptr, 0);                                                                                          \
        done = true;                                                                                                                    \
      }
      LIA_T4_EACH(LIA_T4_LAUNCH)
#undef LIA_T4_LAUNCH
#undef LIA_T4_EACH
    } else
      hipLaunchKernelGGL(lia_gemm_tiled256_kernel, dim3(tiles_m * tiles_n), dim3(512), 4 * T2_TILE_BYTES, st, x, ldx, W, ldw, M, N, K, tiles_m, tiles_n, *ep, *om);
    if (ev1) (void)hipEventRecord(ev1, st);
    return 0;
    }
  }
  int tiles_m = (M + TL_BM - 1) / TL_BM, tiles_n = (N + TL_BN - 1) / TL_BN;
  if (ev0) (void)hipEventRecord(ev0, st);
  hipLaunchKernelGGL(lia_gemm_tiled_kernel, dim3(tiles_m * tiles_n), dim3(256), 4 * TL_TILE_BYTES, st, x, ldx, W, ldw,
                     M, N, K, tiles_m, tiles_n, *ep, *om);
  if (ev1) (void)hipEventRecord(ev1, st);
  return 0;
}
