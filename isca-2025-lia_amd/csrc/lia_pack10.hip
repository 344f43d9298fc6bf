// "pack10": a lossless wire format for streamed bf16 weights, 10.8 bits per value.
//
// The path is bound by the host link: 54.27 GB of OPT-30B weights cross PCIe on every forward at gpu%=10.
// Trained (and N(0, sigma) initialised) weights use only a handful of the 256 bf16 exponents, so each value is re-encoded as one
// byte  sign(1) | mantissa(7)  (incompressible) + an entropy-coded exponent, and the decode is exact: the streamer moves the
// encoded bytes with the same pinned hipMemcpyAsync into a per-slot staging area and a kernel on a side stream rebuilds the bf16
// layer in the HBM slot (0.39 ms per OPT-30B layer, hidden behind the next layer's copy).  The reference moves raw (TPP-blocked)
// bf16, load_layer lia/modeling_opt.py:270-293.
//
// History: pack12 (r01: one 4-bit code per value against a 14-binade window, 12 bits) and pack11 (r01: a 3-bit code in bit-planes
// + a 4-bit overflow stream, 11.13 bits) were the first two generations; pack10 replaced them as the default in r02 and they were
// removed in r05 (git history; LABNOTES.md has their measurements: 88.7 / 95.7 / 98.6 decode tokens/s for 12 / 11 / 10).
#include "lia_common.h"
#include <string.h>

static inline size_t lp12_align(size_t v) { return (v + 255) / 256 * 256; }

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int& total) {
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  total = __shfl(inc, 63, 64);
  return inc - v;
}


// Move `bytes` at dst+from down to dst+to (to <= from) on the device; overlapping ranges go through a temporary
// (a device-to-device hipMemcpy of overlapping ranges is undefined).
static bool lp_move_down(char* dst, size_t to, size_t from, size_t bytes) {
  if (to == from || bytes == 0) return true;
  if (to + bytes <= from) return hipMemcpy(dst + to, dst + from, bytes, hipMemcpyDeviceToDevice) == hipSuccess;
  void* tmp = nullptr;
  if (hipMalloc(&tmp, bytes) != hipSuccess) return false;
  const bool ok = hipMemcpy(tmp, dst + from, bytes, hipMemcpyDeviceToDevice) == hipSuccess &&
                  hipMemcpy(dst + to, tmp, bytes, hipMemcpyDeviceToDevice) == hipSuccess;
  (void)hipFree(tmp);
  return ok;
}

// =====================================================================================================
// The format: a three-level exponent code, close to the exponent entropy (10.55 bits per value for N(0, sigma) weights).  Level 1: 2 bits per value as two bit-planes - codes 0..2 = the three most frequent
// exponents, 3 = "see level 2".  Level 2: 2 bits per level-1 escape, compacted in value order - codes
// 0..2 = the next three exponents, 3 = "see level 3".  Level 3: a 4-bit code per level-2 escape (14-binade
// window, 14 = exponent 0 i.e. +-0 and denormals, 15 = escape record).  The symbol sets are arbitrary exponents, so +-0
// and denormals need no special case.
// The symbol sets are PER REGION of 65536 values (a table of {sym1, sym2, e3} entries behind the header, built on the
// device from a histogram of each region): the 16 tensors of a real checkpoint's layer differ in scale (q/k/v vs fc1/fc2,
// LayerNorm weights near 1, biases near 0) and one table per layer would spread over all their binades -- r01 measured the
// format only on one-sigma Gaussians.  16 bytes per 128 KB of values: 0.002 bits per value.
// Two u32 offset tables (one entry per 1024 values each) + two wave prefix sums of popcounts locate every lane's
// level-2 / level-3 codes.  8 + 2 + 0.274*2 + 0.041*4 + 0.06 = 10.77 bits per value for N(0, sigma) = 67.3 % of the raw bytes.
// Buffer: [header 256][region tables: 16 B each][plane A: n][b0: n/8][b1: n/8][tab2: n/1024 u32][tab3: n/1024 u32][level 2][level 3][escapes]
// =====================================================================================================
constexpr int LP10_REGION_SHIFT = 6;     // 2^6 blocks of 1024 values per region
struct LiaPack10Header {
  uint32_t magic;        // 'LP10'
  uint32_t version;      // 2: per-region symbol tables
  uint32_t n_regions;
  uint32_t region_shift; // blocks per region = 1 << region_shift
  uint64_t n;            // values, multiple of 1024
  uint64_t n_l2;         // level-2 codes (2 bits each)
  uint64_t n_l3;         // level-3 codes (4 bits each)
  uint64_t off_rtab, off_a, off_b0, off_b1, off_tab2, off_tab3, off_l2, off_l3, off_esc;
  uint64_t l2_cap, l3_cap;
  uint32_t n_esc, esc_cap;
  uint32_t overflow;
  uint32_t pad[29];
};
static_assert(sizeof(LiaPack10Header) == 256, "header is 256 bytes");
constexpr uint32_t LP10_MAGIC = 0x3031504cu;
struct Lp10Region { uint32_t sym1, sym2, e3, pad; };   // sym1 / sym2: bytes 0..2 = the exponents of codes 0..2

struct Lp10Codes {
  uint32_t a[4];       // sign|mantissa bytes
  uint32_t b0, b1;     // level-1 planes (16 bits)
  uint32_t l2;         // level-2 codes of the level-1 escapes, 2 bits each, in value order
  uint64_t l3;         // level-3 nibbles of the level-2 escapes, in value order
  int n2, n3;
  uint32_t esc_mask;   // values that need an escape record
};

__device__ __forceinline__ Lp10Codes lp10_codes(const uint32_t (&w)[8], uint32_t sym1, uint32_t sym2, uint32_t e3) {
  Lp10Codes r;
  r.a[0] = r.a[1] = r.a[2] = r.a[3] = 0; r.b0 = r.b1 = 0; r.l2 = 0; r.l3 = 0; r.n2 = r.n3 = 0; r.esc_mask = 0;
  const uint32_t s10 = sym1 & 0xff, s11 = (sym1 >> 8) & 0xff, s12 = (sym1 >> 16) & 0xff;
  const uint32_t s20 = sym2 & 0xff, s21 = (sym2 >> 8) & 0xff, s22 = (sym2 >> 16) & 0xff;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const uint32_t x = (w[k >> 1] >> ((k & 1) * 16)) & 0xffff;
    const uint32_t ex = (x >> 7) & 0xff;
    const uint32_t c1 = ex == s10 ? 0u : ex == s11 ? 1u : ex == s12 ? 2u : 3u;
    r.b0 |= (c1 & 1) << k; r.b1 |= (c1 >> 1) << k;
    if (c1 == 3) {
      const uint32_t c2 = ex == s20 ? 0u : ex == s21 ? 1u : ex == s22 ? 2u : 3u;
      r.l2 |= c2 << (2 * r.n2);
      ++r.n2;
      if (c2 == 3) {
        uint32_t nib = ex - e3;
        if (ex == 0) nib = 14;
        else if (nib > 13) { nib = 15; r.esc_mask |= 1u << k; }
        r.l3 |= (uint64_t)nib << (4 * r.n3);
        ++r.n3;
      }
    }
    r.a[k >> 2] |= (((x >> 8) & 0x80) | (x & 0x7f)) << ((k & 3) * 8);
  }
  return r;
}

__device__ __forceinline__ void lp10_load16(const bf16_t* p, uint32_t (&w)[8]) {
  const uint4 v0 = *(const uint4*)p, v1 = *(const uint4*)(p + 8);
  w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w; w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
}

// pass 0: one workgroup per region -- exponent histogram, then the six most frequent exponents (by count, ties: lower
// exponent first, never the same exponent twice) and the 14 consecutive exponents (>= 1) that cover most of the rest
__global__ __launch_bounds__(256) void lia_pack10_region_kernel(const bf16_t* __restrict__ src, char* __restrict__ dst) {
  LiaPack10Header* hd = (LiaPack10Header*)dst;
  Lp10Region* rtab = (Lp10Region*)(dst + hd->off_rtab);
  __shared__ unsigned h[256];
  const size_t per = (size_t)1024 << hd->region_shift;
  for (unsigned reg = blockIdx.x; reg < hd->n_regions; reg += gridDim.x) {
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t lo = (size_t)reg * per, hi = min(hd->n, lo + per);
    for (size_t i = lo + threadIdx.x * 8; i < hi; i += 256 * 8) {
      const uint4 v = *(const uint4*)(src + i);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { atomicAdd(&h[(w[k] >> 7) & 0xff], 1u); atomicAdd(&h[(w[k] >> 23) & 0xff], 1u); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int top[6];
      unsigned taken[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int t = 0; t < 6; ++t) {
        int best = -1;
        for (int e = 0; e < 256; ++e)
          if (!((taken[e >> 5] >> (e & 31)) & 1) && (best < 0 || h[e] > h[best])) best = e;
        top[t] = best; taken[best >> 5] |= 1u << (best & 31);
      }
      int e3 = 1;
      unsigned long long bestc = 0, cur = 0;
      for (int k = 1; k <= 14; ++k) cur += ((taken[k >> 5] >> (k & 31)) & 1) ? 0 : h[k];
      bestc = cur;
      for (int st = 2; st <= 255 - 14 + 1; ++st) {          // slide the window [st, st + 13]
        const int out = st - 1, in = st + 13;
        cur -= ((taken[out >> 5] >> (out & 31)) & 1) ? 0 : h[out];
        cur += ((taken[in >> 5] >> (in & 31)) & 1) ? 0 : h[in];
        if (cur > bestc) { bestc = cur; e3 = st; }
      }
      Lp10Region r;
      r.sym1 = (uint32_t)top[0] | ((uint32_t)top[1] << 8) | ((uint32_t)top[2] << 16);
      r.sym2 = (uint32_t)top[3] | ((uint32_t)top[4] << 8) | ((uint32_t)top[5] << 16);
      r.e3 = (uint32_t)e3; r.pad = 0;
      rtab[reg] = r;
    }
    __syncthreads();
  }
}

// pass 1: level-2 and level-3 code counts per 1024-value block
__global__ __launch_bounds__(256) void lia_pack10_count_kernel(const bf16_t* __restrict__ src, char* __restrict__ dst) {
  LiaPack10Header* hd = (LiaPack10Header*)dst;
  const size_t nblk = hd->n / 1024;
  uint32_t* tab2 = (uint32_t*)(dst + hd->off_tab2);
  uint32_t* tab3 = (uint32_t*)(dst + hd->off_tab3);
  const Lp10Region* rtab = (const Lp10Region*)(dst + hd->off_rtab);
  const int rshift = (int)hd->region_shift;
  const int lane = threadIdx.x & 63;
  size_t blk = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t stride = (size_t)gridDim.x * 4;
  for (; blk < nblk; blk += stride) {
    const Lp10Region rt = rtab[blk >> rshift];
    uint32_t w[8];
    lp10_load16(src + blk * 1024 + lane * 16, w);
    const Lp10Codes c = lp10_codes(w, rt.sym1, rt.sym2, rt.e3);
    int t2, t3;
    (void)wave_excl_scan(c.n2, lane, t2);
    (void)wave_excl_scan(c.n3, lane, t3);
    if (lane == 0) { tab2[blk] = (uint32_t)t2; tab3[blk] = (uint32_t)t3; }
  }
}

// exclusive scan of one offset table in place (one workgroup); the total goes to *total_out
__global__ __launch_bounds__(1024) void lia_pack10_scan_kernel(char* __restrict__ dst, int which) {
  LiaPack10Header* hd = (LiaPack10Header*)dst;
  const size_t nblk = hd->n / 1024;
  uint32_t* tab = (uint32_t*)(dst + (which == 2 ? hd->off_tab2 : hd->off_tab3));
  __shared__ uint32_t wsum[16];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (size_t base = 0; base < nblk; base += 1024) {
    const size_t i = base + threadIdx.x;
    int v = i < nblk ? (int)tab[i] : 0;
    int wtot;
    int ex = wave_excl_scan(v, lane, wtot);
    if (lane == 63) wsum[wave] = (uint32_t)wtot;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < wave; ++k) woff += wsum[k];
    const unsigned long long c = carry;
    if (i < nblk) tab[i] = (uint32_t)(c + woff + ex);
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + woff + ex + v;
    __syncthreads();
  }
  if (threadIdx.x == 0) { if (which == 2) hd->n_l2 = carry; else hd->n_l3 = carry; }
}

// OR `bits` (nbits <= 64 significant) into a zeroed 32-bit-word stream at bit offset `bitoff`
__device__ __forceinline__ void lp10_or_bits(uint32_t* stream, uint64_t bitoff, uint64_t bits, int nbits) {
  const uint64_t word = bitoff >> 5;
  const int sh = (int)(bitoff & 31);
  atomicOr(&stream[word], (uint32_t)(bits << sh));
  const uint64_t rest = sh ? (bits >> (32 - sh)) : (bits >> 32);
  if (nbits + sh > 32) atomicOr(&stream[word + 1], (uint32_t)rest);
  if (nbits + sh > 64) atomicOr(&stream[word + 2], (uint32_t)(rest >> 32));
}

// pass 2: planes, level-2 / level-3 streams (atomicOr into zeroed streams), escape records
__global__ __launch_bounds__(256) void lia_pack10_encode_kernel(const bf16_t* __restrict__ src, char* __restrict__ dst) {
  LiaPack10Header* hd = (LiaPack10Header*)dst;
  const size_t nblk = hd->n / 1024;
  uint8_t* pa = (uint8_t*)(dst + hd->off_a);
  uint16_t *p0 = (uint16_t*)(dst + hd->off_b0), *p1 = (uint16_t*)(dst + hd->off_b1);
  const uint32_t* tab2 = (const uint32_t*)(dst + hd->off_tab2);
  const uint32_t* tab3 = (const uint32_t*)(dst + hd->off_tab3);
  uint32_t* l2w = (uint32_t*)(dst + hd->off_l2);
  uint32_t* l3w = (uint32_t*)(dst + hd->off_l3);
  uint2* esc = (uint2*)(dst + hd->off_esc);
  const Lp10Region* rtab = (const Lp10Region*)(dst + hd->off_rtab);
  const int rshift = (int)hd->region_shift;
  const int lane = threadIdx.x & 63;
  size_t blk = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t stride = (size_t)gridDim.x * 4;
  for (; blk < nblk; blk += stride) {
    const Lp10Region rt = rtab[blk >> rshift];
    const size_t g = blk * 64 + lane;           // 16-value group index
    uint32_t w[8];
    lp10_load16(src + g * 16, w);
    const Lp10Codes c = lp10_codes(w, rt.sym1, rt.sym2, rt.e3);
    *(uint4*)(pa + g * 16) = uint4{c.a[0], c.a[1], c.a[2], c.a[3]};
    p0[g] = (uint16_t)c.b0; p1[g] = (uint16_t)c.b1;
    int t2, t3;
    const int ex2 = wave_excl_scan(c.n2, lane, t2);
    const int ex3 = wave_excl_scan(c.n3, lane, t3);
    if (c.n2) {
      const uint64_t off = (uint64_t)tab2[blk] + ex2;
      if (off + c.n2 > hd->l2_cap) hd->overflow = 1;
      else lp10_or_bits(l2w, off * 2, c.l2, 2 * c.n2);
    }
    if (c.n3) {
      const uint64_t off = (uint64_t)tab3[blk] + ex3;
      if (off + c.n3 > hd->l3_cap) hd->overflow = 1;
      else lp10_or_bits(l3w, off * 4, c.l3, 4 * c.n3);
    }
    uint32_t m = c.esc_mask;
    while (m) {
      const int k = __ffs(m) - 1;
      m &= m - 1;
      const uint32_t x = (w[k >> 1] >> ((k & 1) * 16)) & 0xffff;
      unsigned slot = atomicAdd(&hd->n_esc, 1u);
      if (slot < hd->esc_cap) esc[slot] = uint2{(uint32_t)(g * 16 + k), x};
      else hd->overflow = 1;
    }
  }
}

// up to 64 bits starting at any bit of a 32-bit-word stream (the stream is padded by >= 16 bytes)
__device__ __forceinline__ uint64_t lp10_get_bits(const uint32_t* stream, uint64_t bitoff, int nbits) {
  const uint64_t word = bitoff >> 5;
  const int sh = (int)(bitoff & 31);
  const uint64_t lo = (uint64_t)stream[word] | ((uint64_t)stream[word + 1] << 32);
  uint64_t v = lo >> sh;
  if (sh && nbits + sh > 64) v |= (uint64_t)stream[word + 2] << (64 - sh);
  return v;
}

// ---- decode (r05) ----------------------------------------------------------------------------------------------
// r01-r04 decoded a lane's 16 values one at a time (extract c1, branch to level 2, branch to level 3, assemble x): ~30 VALU
// operations per value with every branch taken by some lane of the wave -- 616 M values x 30 / (256 CUs x 64 lanes x 2.4 GHz)
// = 0.47 ms of vector ALU time per OPT-30B layer against 0.26 ms of memory time (0.83 GB in + 1.23 GB out at 8 TB/s): the
// kernel was VALU-bound at 0.56 ms.  Now four values travel per 32-bit register, one per byte, and nothing branches:
//   * level 1: the two plane nibbles of a group index a 256-entry LDS table -> the four 2-bit codes as bytes + the escape mask;
//     the exponents are ONE v_perm_b32 (the region's symbol word as the byte source, the code bytes as the selector);
//   * level 2: the next four 2-bit codes of the lane's stream are spread to bytes and DEPOSITED onto the escape positions by a
//     second v_perm_b32 whose selector comes from a 16-entry LDS table indexed by the escape nibble (selector byte = rank of the
//     position among the set bits, 0x0c = constant zero elsewhere) -- a 4-wide PDEP; exponents again by v_perm_b32;
//   * level 3: same deposit for the 4-bit codes, exponent e3 + nibble (0 for codes 14 / 15) out of two 8-byte v_perm tables;
//   * sign|mantissa and exponent bytes become four bf16 by byte arithmetic + two byte interleaves.
// ~14 operations per value.  Same format, same bits out (tests/test_gpu_ops.py::test_pack10_*, tools/pack10_decode_bench).
constexpr unsigned LP10_DECODE_GRID = 32768;
__device__ __forceinline__ uint32_t lp10_bytemask(uint32_t bits01) {      // bytes 0x01 -> 0xff, 0x00 -> 0x00
  return __builtin_amdgcn_perm(0u, 0u, bits01 + 0x0c0c0c0cu);           // v_perm selector 0x0c = 0x00, 0x0d = 0xff
}

// inclusive wave prefix sum by DPP (row_shr 1/2/4/8, row_bcast 15 / 31): six VALU operations, no LDS traffic
__device__ __forceinline__ int wave_incl_scan_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}

// The encoded planes are loaded and the rebuilt values stored NON-TEMPORALLY (each is touched once by this kernel), in a grid of
// 32768 workgroups: measured on MI355X (tools/pack10_decode_bench, one OPT-30B layer, results/r05_pack10_decode_ab.log) 586 us
// (r04 kernel) -> 414 (this arithmetic, r04's 8192 workgroups, plain accesses) -> 397 (32768 workgroups) -> 385 us (non-temporal)
// = 5.36 TB/s = 0.67 of 8 TB/s; issuing the next block's plane loads before decoding this one (a software prefetch) gained
// nothing on top (391-402 us), one workgroup per four blocks without a loop lost (414-460 us).
__global__ __launch_bounds__(256) void lia_pack10_decode_kernel(const char* __restrict__ src, bf16_t* __restrict__ dst) {
  __shared__ uint2 lut1[256];        // [n1 << 4 | n0] -> {code bytes c1 of the four values, 0xff where c1 == 3}
  __shared__ uint32_t lut4[16];      // [position nibble] -> v_perm selector that deposits compact bytes 0.. onto the set positions
  {
    const uint32_t t = threadIdx.x, n0 = t & 15, n1 = t >> 4;
    uint32_t c = 0, m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t ci = ((n0 >> i) & 1) | (((n1 >> i) & 1) << 1);
      c |= ci << (8 * i);
      if (ci == 3) m |= 0xffu << (8 * i);
    }
    lut1[t] = uint2{c, m};
    if (t < 16) {
      uint32_t sel = 0, rank = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if ((t >> i) & 1) sel |= rank++ << (8 * i);
        else sel |= 0x0cu << (8 * i);
      }
      lut4[t] = sel;
    }
  }
  __syncthreads();
  const LiaPack10Header* hd = (const LiaPack10Header*)src;
  const size_t nblk = hd->n / 1024;
  const Lp10Region* rtab = (const Lp10Region*)(src + hd->off_rtab);
  const int rshift = (int)hd->region_shift;
  const uint8_t* pa = (const uint8_t*)(src + hd->off_a);
  const uint16_t *p0 = (const uint16_t*)(src + hd->off_b0), *p1 = (const uint16_t*)(src + hd->off_b1);
  const uint32_t* tab2 = (const uint32_t*)(src + hd->off_tab2);
  const uint32_t* tab3 = (const uint32_t*)(src + hd->off_tab3);
  const uint32_t* l2w = (const uint32_t*)(src + hd->off_l2);
  const uint32_t* l3w = (const uint32_t*)(src + hd->off_l3);
  const int lane = threadIdx.x & 63;
  size_t blk = (size_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t stride = (size_t)gridDim.x * 4;
  typedef uint32_t lp_u32x4 __attribute__((ext_vector_type(4)));
  for (; blk < nblk; blk += stride) {
    const Lp10Region rt = rtab[blk >> rshift];
    const uint32_t sym1 = __builtin_amdgcn_readfirstlane(rt.sym1), sym2 = __builtin_amdgcn_readfirstlane(rt.sym2);
    const uint32_t e3b = __builtin_amdgcn_readfirstlane(rt.e3) * 0x01010101u;
    // level-3 exponent tables: byte i of q0|q1|q2|q3 = e3 + i, entries 14 / 15 = 0
    const uint32_t q0 = e3b + 0x03020100u, q1 = e3b + 0x07060504u, q2 = e3b + 0x0b0a0908u, q3 = (e3b + 0x00000d0cu) & 0x0000ffffu;
    const size_t g = blk * 64 + lane;
    const lp_u32x4 av = __builtin_nontemporal_load((const lp_u32x4*)(pa + g * 16));
    const uint32_t a[4] = {av[0], av[1], av[2], av[3]};
    const uint32_t b0 = __builtin_nontemporal_load(p0 + g), b1 = __builtin_nontemporal_load(p1 + g);
    const uint32_t esc1 = b0 & b1;
    const int n2 = __popc(esc1);
    const int ex2 = wave_incl_scan_dpp(n2) - n2;
    uint32_t l2 = 0;
    if (n2) l2 = (uint32_t)lp10_get_bits(l2w, ((uint64_t)tab2[blk] + ex2) * 2, 2 * n2);
    if (n2 < 16) l2 &= (1u << (2 * n2)) - 1u;
    const int n3 = __popc(l2 & (l2 >> 1) & 0x55555555u);
    const int ex3 = wave_incl_scan_dpp(n3) - n3;
    uint64_t l3 = 0;
    if (n3) l3 = lp10_get_bits(l3w, ((uint64_t)tab3[blk] + ex3) * 4, 4 * n3);
    uint32_t o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // ---- level 1
      const uint32_t n0 = (b0 >> (4 * j)) & 15u, n1 = (b1 >> (4 * j)) & 15u;
      const uint2 c1m = lut1[(n1 << 4) | n0];
      uint32_t E = __builtin_amdgcn_perm(0u, sym1, c1m.x);
      // ---- level 2: the lane's next popc(nib) codes go to the escape positions of this group
      const uint32_t nib = n0 & n1;
      const uint32_t c8 = l2 & 0xffu;
      uint32_t s2 = (c8 | (c8 << 12)) & 0x000f000fu;
      s2 = (s2 | (s2 << 6)) & 0x03030303u;
      const uint32_t C2 = __builtin_amdgcn_perm(0u, s2, lut4[nib]);
      l2 >>= 2 * __popc(nib);
      E = (__builtin_amdgcn_perm(0u, sym2, C2) & c1m.y) | (E & ~c1m.y);
      // ---- level 3: positions whose level-2 code is 3
      const uint32_t t3 = C2 & (C2 >> 1) & 0x01010101u;
      const uint32_t idx3 = __builtin_amdgcn_udot4(t3, 0x08040201u, 0u, false);
      const uint32_t x16 = (uint32_t)l3 & 0xffffu;
      uint32_t s3 = (x16 | (x16 << 8)) & 0x00ff00ffu;
      s3 = (s3 | (s3 << 4)) & 0x0f0f0f0fu;
      const uint32_t N = __builtin_amdgcn_perm(0u, s3, lut4[idx3]);
      l3 >>= 4 * __popc(idx3);
      const uint32_t sel7 = N & 0x07070707u;
      const uint32_t hi8 = lp10_bytemask((N >> 3) & 0x01010101u);
      const uint32_t E3 = (__builtin_amdgcn_perm(q3, q2, sel7) & hi8) | (__builtin_amdgcn_perm(q1, q0, sel7) & ~hi8);
      const uint32_t m3 = lp10_bytemask(t3);
      E = (E3 & m3) | (E & ~m3);
      // ---- bf16 = sign << 15 | exponent << 7 | mantissa, four at a time: low bytes, high bytes, interleave
      const uint32_t A = a[j];
      const uint32_t lo = (A & 0x7f7f7f7fu) | ((E & 0x01010101u) << 7);
      const uint32_t hi = (A & 0x80808080u) | ((E >> 1) & 0x7f7f7f7fu);
      o[2 * j] = __builtin_amdgcn_perm(hi, lo, 0x05010400u);
      o[2 * j + 1] = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
    }
    __builtin_nontemporal_store(lp_u32x4{o[0], o[1], o[2], o[3]}, (lp_u32x4*)(dst + g * 16));
    __builtin_nontemporal_store(lp_u32x4{o[4], o[5], o[6], o[7]}, (lp_u32x4*)(dst + g * 16 + 8));
  }
}

__global__ __launch_bounds__(256) void lia_pack10_patch_kernel(const char* __restrict__ src, bf16_t* __restrict__ dst) {
  const LiaPack10Header* hd = (const LiaPack10Header*)src;
  const uint2* esc = (const uint2*)(src + hd->off_esc);
  const unsigned n = hd->n_esc;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    uint2 r = esc[i];
    dst[r.x] = (bf16_t)r.y;
  }
}

// room an encode may need: level 2 up to n codes, level 3 up to n/2 nibbles, escapes n/16 records
static inline size_t lp10_regions(size_t n_values) { return (n_values / 1024 + ((size_t)1 << LP10_REGION_SHIFT) - 1) >> LP10_REGION_SHIFT; }
extern "C" size_t lia_pack10_bound(size_t n_values) {
  return 256 + lp12_align(lp10_regions(n_values) * sizeof(Lp10Region)) + lp12_align(n_values) + 2 * lp12_align(n_values / 8) + 2 * lp12_align((n_values / 1024) * 4 + 16) +
         2 * lp12_align(n_values / 4 + 16) + lp12_align((n_values / 16) * 8);
}

// Encode n_values bf16 (device; a multiple of 1024) into dst (device, >= lia_pack10_bound).  Synchronous (model placement time).
// *out_bytes = bytes to ship (header + tables + planes + the level-2 / level-3 streams and escape records actually used).
// Returns 0, 1 if the layer does not fit the format (too many escapes) and must travel raw, negative on error.
extern "C" int lia_pack10_encode(const bf16_t* src, size_t n_values, char* dst, size_t dst_capacity, size_t* out_bytes) {
  if (!src || !dst || !out_bytes || (n_values % 1024) || dst_capacity < lia_pack10_bound(n_values)) return -1;
  LiaPack10Header hd;
  memset(&hd, 0, sizeof(hd));
  hd.magic = LP10_MAGIC; hd.version = 2;
  hd.n = n_values; hd.region_shift = LP10_REGION_SHIFT; hd.n_regions = (uint32_t)lp10_regions(n_values);
  hd.esc_cap = (uint32_t)(n_values / 16); hd.l2_cap = n_values; hd.l3_cap = n_values / 2;
  const size_t tab_bytes = lp12_align((n_values / 1024) * 4 + 16);
  const size_t l2_bytes = lp12_align(n_values / 4 + 16), l3_bytes = lp12_align(n_values / 4 + 16);
  hd.off_rtab = 256; hd.off_a = hd.off_rtab + lp12_align((size_t)hd.n_regions * sizeof(Lp10Region));
  hd.off_b0 = hd.off_a + lp12_align(n_values); hd.off_b1 = hd.off_b0 + lp12_align(n_values / 8);
  hd.off_tab2 = hd.off_b1 + lp12_align(n_values / 8); hd.off_tab3 = hd.off_tab2 + tab_bytes;
  // provisional stream positions at full capacity; compacted below once the real sizes are known
  hd.off_l2 = hd.off_tab3 + tab_bytes; hd.off_l3 = hd.off_l2 + l2_bytes; hd.off_esc = hd.off_l3 + l3_bytes;
  if (hipMemcpy(dst, &hd, sizeof(hd), hipMemcpyHostToDevice) != hipSuccess) return -3;
  if (hipMemset(dst + hd.off_l2, 0, l2_bytes + l3_bytes) != hipSuccess) return -3;
  (void)hipGetLastError();
  hipLaunchKernelGGL(lia_pack10_region_kernel, dim3(hd.n_regions < 4096 ? hd.n_regions : 4096), dim3(256), 0, 0, src, dst);
  hipLaunchKernelGGL(lia_pack10_count_kernel, dim3(2048), dim3(256), 0, 0, src, dst);
  hipLaunchKernelGGL(lia_pack10_scan_kernel, dim3(1), dim3(1024), 0, 0, dst, 2);
  hipLaunchKernelGGL(lia_pack10_scan_kernel, dim3(1), dim3(1024), 0, 0, dst, 3);
  hipLaunchKernelGGL(lia_pack10_encode_kernel, dim3(2048), dim3(256), 0, 0, src, dst);
  // a launch that failed leaves the header as it was written above (overflow = 0, empty planes): that must not read as success
  if (hipGetLastError() != hipSuccess) return -3;
  if (hipMemcpy(&hd, dst, sizeof(hd), hipMemcpyDeviceToHost) != hipSuccess) return -3;
  if (hd.overflow || hd.n_esc > hd.esc_cap || hd.n_l2 > hd.l2_cap || hd.n_l3 > hd.l3_cap) return 1;
  // compact: level 3 right behind the level-2 bytes in use, escape records right behind level 3 (downward moves, in order)
  const size_t used_l2 = lp12_align((size_t)((hd.n_l2 + 3) / 4) + 16);
  const size_t used_l3 = lp12_align((size_t)((hd.n_l3 + 1) / 2) + 16);
  const size_t new_l3 = hd.off_l2 + used_l2, new_esc = new_l3 + used_l3;
  if (!lp_move_down(dst, new_l3, hd.off_l3, used_l3)) return -3;
  if (!lp_move_down(dst, new_esc, hd.off_esc, (size_t)hd.n_esc * 8)) return -3;
  hd.off_l3 = new_l3; hd.off_esc = new_esc;
  if (hipMemcpy(dst, &hd, sizeof(hd), hipMemcpyHostToDevice) != hipSuccess) return -3;
  *out_bytes = (size_t)hd.off_esc + lp12_align((size_t)hd.n_esc * 8);
  return 0;
}

// Host-side check of an encoded buffer BEFORE it is trusted: the decode kernel sizes its loops and takes every offset
// from the header, so a stale or corrupt layer file whose header disagrees with the slot would write past it.  `buf`: the first
// 256 bytes (host memory: a mapped checkpoint file, a pinned copy); staged_bytes: the bytes the caller holds / will copy;
// n_values: the bf16 values the destination slot has room for.  0 = consistent, negative = the reason (lia_last_error has the text).
extern "C" int lia_pack10_validate(const void* buf, size_t staged_bytes, size_t n_values) {
  if (!buf || staged_bytes < sizeof(LiaPack10Header)) return -1;
  LiaPack10Header hd;
  memcpy(&hd, buf, sizeof(hd));
  if (hd.magic != LP10_MAGIC || hd.version != 2) return -2;
  if (hd.n != n_values || (n_values % 1024) || hd.region_shift != LP10_REGION_SHIFT || hd.n_regions != (uint32_t)lp10_regions(n_values)) return -3;
  const uint64_t offs[] = {hd.off_rtab, hd.off_a, hd.off_b0, hd.off_b1, hd.off_tab2, hd.off_tab3, hd.off_l2, hd.off_l3, hd.off_esc};
  uint64_t prev = sizeof(LiaPack10Header);
  for (uint64_t o : offs) { if (o < prev || o > staged_bytes) return -4; prev = o; }
  if (hd.off_a + n_values > hd.off_b0 || hd.off_b1 + n_values / 8 > hd.off_tab2) return -4;
  if (hd.overflow || hd.n_esc > hd.esc_cap || hd.n_l2 > hd.l2_cap || hd.n_l3 > hd.l3_cap || hd.n_l2 > n_values || hd.n_l3 > n_values) return -5;
  if (hd.off_l2 + (hd.n_l2 + 3) / 4 > hd.off_l3 || hd.off_l3 + (hd.n_l3 + 1) / 2 > hd.off_esc || hd.off_esc + (uint64_t)hd.n_esc * 8 > staged_bytes) return -4;
  return 0;
}

// Decode a pack10 buffer whose header the caller has validated (lia_pack10_validate at placement time: the kernel reads its loop
// bounds and every offset from the header, `n_values` only sizes the grid); asynchronous on `st`.
// ev0 / ev1 (nullable): recorded on `st` immediately around the MAIN decode kernel (lia_stream_decode_stats; the patch kernel
// behind it -- a few hundred escape records -- is outside the bracket, as a profiler's per-kernel duration would have it).
extern "C" void lia_packed_decode_launch(const char* src, bf16_t* dst, size_t n_values, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  const size_t nblk = n_values / 1024;
  unsigned blocks = (unsigned)((nblk + 3) / 4);
  if (blocks > LP10_DECODE_GRID) blocks = LP10_DECODE_GRID;
  if (blocks == 0) return;
  if (ev0) (void)hipEventRecord(ev0, st);
  hipLaunchKernelGGL(lia_pack10_decode_kernel, dim3(blocks), dim3(256), 0, st, src, dst);
  if (ev1) (void)hipEventRecord(ev1, st);
  hipLaunchKernelGGL(lia_pack10_patch_kernel, dim3(64), dim3(256), 0, st, src, dst);
}
