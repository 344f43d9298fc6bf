// A/B of the pack10 wire-format decode: the r01-r04 kernel (one value at a time; kept HERE, the product no longer carries it) against the r05 kernel
// (four values per register, v_perm deposits), on one OPT-30B layer's worth of values: N(0, 0.02) weights + a sprinkle of zeros,
// denormals, large outliers and Inf / NaN.  Checks both against the source bit for bit, then times them (HIP events, cold input:
// 0.83 GB in + 1.23 GB out per launch is larger than L2 + Infinity Cache).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I isca-2025-lia_amd/csrc -I include tools/pack10_decode_bench.hip -o tools/pack10_decode_bench
#include "../isca-2025-lia_amd/csrc/lia_pack10.hip"

// the r01-r04 decode, kept for tools/pack10_decode_bench.hip only (bit-identity + timing of the two)
__global__ __launch_bounds__(256) void lia_pack10_decode_v1_kernel(const char* __restrict__ src, bf16_t* __restrict__ dst) {
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
  size_t blk = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t stride = (size_t)gridDim.x * 4;
  for (; blk < nblk; blk += stride) {
    const Lp10Region rt = rtab[blk >> rshift];
    const uint32_t sym1 = rt.sym1, sym2 = rt.sym2, e3 = rt.e3;
    const size_t g = blk * 64 + lane;
    const uint4 av = *(const uint4*)(pa + g * 16);
    const uint32_t a[4] = {av.x, av.y, av.z, av.w};
    const uint32_t b0 = p0[g], b1 = p1[g];
    const uint32_t esc1 = b0 & b1;
    const int n2 = __popc(esc1);
    int t2;
    const int ex2 = wave_excl_scan(n2, lane, t2);
    uint32_t l2 = 0;
    if (n2) l2 = (uint32_t)lp10_get_bits(l2w, ((uint64_t)tab2[blk] + ex2) * 2, 2 * n2);
    if (n2 < 16) l2 &= (1u << (2 * n2)) - 1u;
    const int n3 = __popc(l2 & (l2 >> 1) & 0x55555555u);
    int t3;
    const int ex3 = wave_excl_scan(n3, lane, t3);
    uint64_t l3 = 0;
    if (n3) l3 = lp10_get_bits(l3w, ((uint64_t)tab3[blk] + ex3) * 4, 4 * n3);
    uint32_t o[8];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t sm = (a[k >> 2] >> ((k & 3) * 8)) & 0xff;
      const uint32_t c1 = ((b0 >> k) & 1) | (((b1 >> k) & 1) << 1);
      uint32_t ex;
      if (c1 < 3) ex = (sym1 >> (8 * c1)) & 0xff;
      else {
        const uint32_t c2 = l2 & 3;
        l2 >>= 2;
        if (c2 < 3) ex = (sym2 >> (8 * c2)) & 0xff;
        else {
          const uint32_t nib = (uint32_t)(l3 & 0xf);
          l3 >>= 4;
          ex = nib < 14 ? e3 + nib : 0;      // 14: exponent 0; 15: placeholder, patched from the escape records
        }
      }
      const uint32_t x = ((sm & 0x80) << 8) | (ex << 7) | (sm & 0x7f);
      if (k & 1) o[k >> 1] |= x << 16; else o[k >> 1] = x;
    }
    *(uint4*)(dst + g * 16) = uint4{o[0], o[1], o[2], o[3]};
    *(uint4*)(dst + g * 16 + 8) = uint4{o[4], o[5], o[6], o[7]};
  }
}


#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void fill_normal(uint16_t* p, size_t n, uint32_t seed, float sigma) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t h = (uint32_t)(i * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    uint32_t h2 = h * 747796405u + 2891336453u; h2 ^= h2 >> 15; h2 *= 2246822519u; h2 ^= h2 >> 13;
    const float u1 = ((h >> 8) + 1) * (1.0f / 16777217.0f), u2 = (h2 >> 8) * (1.0f / 16777216.0f);
    float f = sigma * sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
    uint32_t bits = __float_as_uint(f);
    uint16_t v = (uint16_t)((bits + 0x7fffu + ((bits >> 16) & 1u)) >> 16);
    const uint32_t r = h2 & 0xffff;
    if (r == 1) v = 0; else if (r == 2) v = 0x8000; else if (r == 3) v = 0x0001; else if (r == 4) v = 0x7f80; else if (r == 5) v = 0x7fc1;
    else if (r == 6) v = 0x4489; else if (r == 7) v = 0xff7f;
    p[i] = v;
  }
}

__global__ void count_diff(const uint16_t* a, const uint16_t* b, size_t n, unsigned long long* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned long long c = 0;
  for (; i < n; i += stride) c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)616657920;      // one OPT-30B layer (1,233,315,072 B packed buffer / 2, a multiple of 1024)
  const float sigma = argc > 2 ? (float)atof(argv[2]) : 0.02f;
  uint16_t *srcv, *out; char* enc;
  CK(hipMalloc(&srcv, n * 2)); CK(hipMalloc(&out, n * 2));
  const size_t cap = lia_pack10_bound(n);
  CK(hipMalloc(&enc, cap));
  fill_normal<<<4096, 256>>>(srcv, n, 12345u, sigma);
  CK(hipDeviceSynchronize());
  size_t bytes = 0;
  const int rc = lia_pack10_encode(srcv, n, enc, cap, &bytes);
  printf("encode rc %d: %zu values -> %zu bytes = %.3f bits per value\n", rc, n, bytes, 8.0 * bytes / n);
  if (rc != 0) return 1;
  unsigned long long* dcount; CK(hipMalloc(&dcount, 8));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned blocks = (unsigned)((n / 1024 + 3) / 4); if (blocks > 8192) blocks = 8192;
  const unsigned grids[] = {8192u, 16384u, 32768u, 65536u};
  for (int which = 0; which < 2; ++which) {
    for (unsigned gsel = 0; gsel < (which >= 1 ? 4u : 1u); ++gsel) {
      const unsigned gb = grids[gsel] < (unsigned)((n / 1024 + 3) / 4) ? grids[gsel] : (unsigned)((n / 1024 + 3) / 4);
      CK(hipMemsetAsync(out, 0xee, n * 2, st));
      auto launch = [&]() {
        if (which == 0) hipLaunchKernelGGL(lia_pack10_decode_v1_kernel, dim3(gb), dim3(256), 0, st, enc, out);
        else hipLaunchKernelGGL(lia_pack10_decode_kernel, dim3(gb), dim3(256), 0, st, enc, out);
      };
      auto launch_all = [&]() {
        launch();
        hipLaunchKernelGGL(lia_pack10_patch_kernel, dim3(64), dim3(256), 0, st, enc, out);
      };
      launch_all();
      CK(hipMemsetAsync(dcount, 0, 8, st));
      count_diff<<<2048, 256, 0, st>>>(srcv, out, n, dcount);
      unsigned long long nd = 0;
      CK(hipMemcpyAsync(&nd, dcount, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
      float best = 1e9f, sum = 0.f;
      const int reps = 10;
      for (int it = 0; it < reps; ++it) {
        CK(hipEventRecord(e0, st));
        launch();
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
      }
      const double traffic = (double)bytes + 2.0 * n;
      printf("%s grid %5u: mismatches vs source %llu; decode kernel %.1f us best, %.1f us mean -> %.2f TB/s (in %.2f GB + out %.2f GB) = %.3f of 8 TB/s\n",
             which == 0 ? "v1 (r04)" : "v2 (r05)", gb, nd, best * 1e3, sum / reps * 1e3, traffic / (sum / reps * 1e-3) / 1e12, bytes / 1e9, 2.0 * n / 1e9,
             traffic / (sum / reps * 1e-3) / 8e12);
      if (nd) return 2;
    }
  }
  return 0;
}
