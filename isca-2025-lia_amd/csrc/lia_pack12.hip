// "pack12": a lossless 12-bit wire format for streamed bf16 weights.
//
// The path is bound by the host link: 54.27 GB of OPT-30B weights cross PCIe on every forward at gpu%=10
// (SURVEY.md section 8d).  Trained (and N(0, sigma) initialised) weights use only a handful of the 256 bf16
// exponents, so each value is re-encoded as  sign(1) | mantissa(7)  +  a 4-bit exponent code:
//     code 0..13 : exponent = e0 + code        (e0 = start of the best 14-binade window of this layer)
//     code 14    : the value is +-0
//     code 15    : escape -- the raw bf16 lives in a side list of {index, value} records
// i.e. 12 bits instead of 16 (75 % of the bytes) for every value inside the window, and the decode is exact: the
// streamer moves the packed bytes with the same pinned hipMemcpyAsync and a kernel on the copy stream rebuilds the
// bf16 layer in the HBM slot (reads 0.75 B, writes 2 B per value: ~0.4 ms per OPT-30B layer, hidden behind the
// next layer's copy).  The reference moves raw (TPP-blocked) bf16, load_layer lia/modeling_opt.py:270-293.
//
// Buffer: [header 256 B][plane A: n bytes sign|mantissa][plane B: n/2 bytes, two codes per byte][escapes: 8 B each]
#include "lia_common.h"
#include <string.h>

struct LiaPack12Header {
  uint32_t magic;        // 'LP12'
  uint32_t e0;           // first exponent of the window
  uint64_t n;            // bf16 values (multiple of 16)
  uint32_t n_esc;        // escape records that follow plane B
  uint32_t esc_cap;      // records the buffer has room for (encode only)
  uint64_t off_a, off_b, off_esc;
  uint32_t overflow;     // encode: more escapes than room -> the caller must ship the layer raw
  uint32_t pad[51];
};
static_assert(sizeof(LiaPack12Header) == 256, "header is 256 bytes");

constexpr uint32_t LP12_MAGIC = 0x3231504cu;

__global__ __launch_bounds__(256) void lia_pack12_hist_kernel(const bf16_t* __restrict__ src, size_t n, unsigned* __restrict__ hist) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) atomicAdd(&h[(src[i] >> 7) & 0xff], 1u);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void lia_pack12_encode_kernel(const bf16_t* __restrict__ src, char* __restrict__ dst) {
  LiaPack12Header* hd = (LiaPack12Header*)dst;
  const size_t n16 = hd->n / 16;
  const uint32_t e0 = hd->e0;
  uint8_t* pa = (uint8_t*)(dst + hd->off_a);
  uint8_t* pb = (uint8_t*)(dst + hd->off_b);
  uint2* esc = (uint2*)(dst + hd->off_esc);
  size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; g < n16; g += stride) {
    const uint4 v0 = *(const uint4*)(src + g * 16), v1 = *(const uint4*)(src + g * 16 + 8);
    const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    uint32_t a[4] = {0, 0, 0, 0}, b[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t x = (w[k >> 1] >> ((k & 1) * 16)) & 0xffff;
      const uint32_t ex = (x >> 7) & 0xff;
      uint32_t code = ex - e0;   // wraps for ex < e0 -> large -> escape
      if ((x & 0x7fff) == 0) code = 14;
      else if (code > 13) {
        code = 15;
        unsigned slot = atomicAdd(&hd->n_esc, 1u);
        if (slot < hd->esc_cap) esc[slot] = uint2{(uint32_t)(g * 16 + k), x};
        else hd->overflow = 1;
      }
      a[k >> 2] |= (((x >> 8) & 0x80) | (x & 0x7f)) << ((k & 3) * 8);
      b[k >> 3] |= code << ((k & 7) * 4);
    }
    *(uint4*)(pa + g * 16) = uint4{a[0], a[1], a[2], a[3]};
    *(uint2*)(pb + g * 8) = uint2{b[0], b[1]};
  }
}

__global__ __launch_bounds__(256) void lia_pack12_decode_kernel(const char* __restrict__ src, bf16_t* __restrict__ dst) {
  const LiaPack12Header* hd = (const LiaPack12Header*)src;
  const size_t n16 = hd->n / 16;
  const uint32_t e0 = hd->e0;
  const uint8_t* pa = (const uint8_t*)(src + hd->off_a);
  const uint8_t* pb = (const uint8_t*)(src + hd->off_b);
  size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; g < n16; g += stride) {
    const uint4 av = *(const uint4*)(pa + g * 16);
    const uint2 bv = *(const uint2*)(pb + g * 8);
    const uint32_t a[4] = {av.x, av.y, av.z, av.w}, b[2] = {bv.x, bv.y};
    uint32_t o[8];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t sm = (a[k >> 2] >> ((k & 3) * 8)) & 0xff;
      const uint32_t code = (b[k >> 3] >> ((k & 7) * 4)) & 0xf;
      uint32_t x = ((sm & 0x80) << 8);
      if (code < 14) x |= ((e0 + code) << 7) | (sm & 0x7f);
      if (k & 1) o[k >> 1] |= x << 16; else o[k >> 1] = x;
    }
    *(uint4*)(dst + g * 16) = uint4{o[0], o[1], o[2], o[3]};
    *(uint4*)(dst + g * 16 + 8) = uint4{o[4], o[5], o[6], o[7]};
  }
}

__global__ __launch_bounds__(256) void lia_pack12_patch_kernel(const char* __restrict__ src, bf16_t* __restrict__ dst) {
  const LiaPack12Header* hd = (const LiaPack12Header*)src;
  const uint2* esc = (const uint2*)(src + hd->off_esc);
  const unsigned n = hd->n_esc;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    uint2 r = esc[i];
    dst[r.x] = (bf16_t)r.y;
  }
}

static inline size_t lp12_align(size_t v) { return (v + 255) / 256 * 256; }

// room an encode of n values may need (escape capacity n/16 records)
extern "C" size_t lia_pack12_bound(size_t n_values) {
  return 256 + lp12_align(n_values) + lp12_align(n_values / 2) + lp12_align((n_values / 16) * 8);
}

// Encode n_values bf16 (device) into dst (device, >= lia_pack12_bound).  Synchronous (model placement time).
// *out_bytes = bytes to ship (header + planes + the escape records actually used).  Returns 0, or 1 if the layer
// does not fit the format (too many escapes) and must travel raw.
extern "C" int lia_pack12_encode(const bf16_t* src, size_t n_values, char* dst, size_t dst_capacity, size_t* out_bytes) {
  if (!src || !dst || !out_bytes || (n_values % 16) || dst_capacity < lia_pack12_bound(n_values)) return -1;
  unsigned* hist = nullptr;
  if (hipMalloc((void**)&hist, 256 * sizeof(unsigned)) != hipSuccess) return -2;
  (void)hipMemset(hist, 0, 256 * sizeof(unsigned));
  hipLaunchKernelGGL(lia_pack12_hist_kernel, dim3(1024), dim3(256), 0, 0, src, n_values, hist);
  unsigned h[256];
  if (hipMemcpy(h, hist, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipFree(hist); return -3; }
  (void)hipFree(hist);
  unsigned long long best = 0, cur = 0;
  int e0 = 1;
  for (int e = 1; e <= 255 - 14; ++e) {   // exponent 0 (zero / denormals) is never inside the window
    cur = 0;
    for (int k = 0; k < 14; ++k) cur += h[e + k];
    if (cur > best) { best = cur; e0 = e; }
  }
  LiaPack12Header hd;
  memset(&hd, 0, sizeof(hd));
  hd.magic = LP12_MAGIC; hd.e0 = (uint32_t)e0; hd.n = n_values; hd.n_esc = 0; hd.esc_cap = (uint32_t)(n_values / 16);
  hd.off_a = 256; hd.off_b = hd.off_a + lp12_align(n_values); hd.off_esc = hd.off_b + lp12_align(n_values / 2);
  if (hipMemcpy(dst, &hd, sizeof(hd), hipMemcpyHostToDevice) != hipSuccess) return -3;
  hipLaunchKernelGGL(lia_pack12_encode_kernel, dim3(2048), dim3(256), 0, 0, src, dst);
  if (hipMemcpy(&hd, dst, sizeof(hd), hipMemcpyDeviceToHost) != hipSuccess) return -3;
  if (hd.overflow || hd.n_esc > hd.esc_cap) return 1;
  *out_bytes = (size_t)hd.off_esc + lp12_align((size_t)hd.n_esc * 8);
  return 0;
}

// Rebuild the bf16 values: asynchronous on `st`.  src/dst are device pointers; the header is read on the device.
extern "C" void lia_pack12_decode_launch(const char* src, bf16_t* dst, size_t n_values, hipStream_t st) {
  size_t n16 = n_values / 16;
  unsigned blocks = (unsigned)((n16 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) return;
  hipLaunchKernelGGL(lia_pack12_decode_kernel, dim3(blocks), dim3(256), 0, st, src, dst);
  hipLaunchKernelGGL(lia_pack12_patch_kernel, dim3(64), dim3(256), 0, st, src, dst);
}
