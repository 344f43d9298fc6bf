// Host side of LIA's cooperative policies (product code, C++/OpenMP/AVX-512):
//   * lia_host_attention  -- the policy-2 attention that runs on the CPU cores over the host KV cache
//   * lia_tpp_block / lia_tpp_unblock -- the reference's blocked weight wire format <-> row-major
//   * numa_alloc_* -- the CXL / NUMA tier allocator (same four exports as lia/cxl/numa_alloc.c)
// Built for x86-64-v4 (AVX-512): the intersection of the build container and the GPU box's EPYC 9575F.
#include <errno.h>
#include <immintrin.h>
#include <math.h>
#include <numa.h>
#include <numaif.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <vector>

#include "../../include/lia_hip.h"

extern "C" void lia_set_error(const char* fmt, ...);

// ------------------------------------------------------------------------------------------------
// host attention
// ------------------------------------------------------------------------------------------------
static inline __m512 bf16x16_to_f32(const lia_bf16* p) {
  __m256i h = _mm256_loadu_si256((const __m256i*)p);
  return _mm512_castsi512_ps(_mm512_slli_epi32(_mm512_cvtepu16_epi32(h), 16));
}

static inline lia_bf16 f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (lia_bf16)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (lia_bf16)(u >> 16);
}

// Restates scale_dot_product_for_indirect_access_kv_cache (MaskedMultiHeadAttentionKrnl.cpp:513-842):
// the new K/V rows are written into the cache (:588-613, :740-764), scores = q.k / sqrt(d) in fp32
// (:676-691, zero mask for equal-length prompts), fp32 softmax, fp32 weighted V sum, one bf16 rounding at
// the output (:826-828).  beam_idx is the identity under greedy search (:1393-1402) and is not
// materialised.  Work is split over (batch row, group of heads) so that for each cached position a
// thread streams one contiguous run of the [S][B][h][d] cache.
extern "C" int lia_host_attention(const lia_bf16* q, const lia_bf16* k, const lia_bf16* v, lia_bf16* kcache,
                                  lia_bf16* vcache, lia_bf16* out, int B, int T, int pos0, int heads, int head_dim,
                                  int cache_batch, int b0, int n_threads) {
  if (!q || !k || !v || !kcache || !vcache || !out) {
    lia_set_error("lia_host_attention: NULL tensor");
    return LIA_ERR_MISSING;
  }
  if (B <= 0 || T <= 0 || pos0 < 0 || heads <= 0 || head_dim <= 0 || (head_dim % 16) != 0 || b0 < 0 ||
      b0 + B > cache_batch) {
    lia_set_error("lia_host_attention: bad shape B=%d T=%d pos0=%d heads=%d d=%d cache_batch=%d b0=%d", B, T, pos0,
                  heads, head_dim, cache_batch, b0);
    return LIA_ERR_INVALID;
  }
  const int d = head_dim;
  const long hd = (long)heads * d;
  const long row = (long)cache_batch * hd;
  const int S = pos0 + T;
  const float inv_scale = 1.0f / sqrtf((float)d);
  const int nv = d / 16;
  int G = 8;
  while (heads % G) G >>= 1;
  const int ngroups = heads / G;
  if (n_threads <= 0) n_threads = omp_get_max_threads();

#pragma omp parallel num_threads(n_threads)
  {
    std::vector<float> sc((size_t)G * S);
#pragma omp for collapse(2) schedule(dynamic, 1)
    for (int b = 0; b < B; ++b)
      for (int g = 0; g < ngroups; ++g) {
        const long coff = (long)(b0 + b) * hd + (long)g * G * d;
        // append the fresh rows of this (batch row, head group)
        for (int t = 0; t < T; ++t) {
          const long src = ((long)b * T + t) * hd + (long)g * G * d;
          memcpy(kcache + (long)(pos0 + t) * row + coff, k + src, (size_t)G * d * sizeof(lia_bf16));
          memcpy(vcache + (long)(pos0 + t) * row + coff, v + src, (size_t)G * d * sizeof(lia_bf16));
        }
        for (int t = 0; t < T; ++t) {
          const int lim = pos0 + t;  // keys 0..lim (causal inside the new block)
          const lia_bf16* qp = q + ((long)b * T + t) * hd + (long)g * G * d;
          for (int j = 0; j <= lim; ++j) {
            const lia_bf16* kp = kcache + (long)j * row + coff;
            if (j + 2 <= lim) _mm_prefetch((const char*)(kcache + (long)(j + 2) * row + coff), _MM_HINT_T0);
            for (int hh = 0; hh < G; ++hh) {
              __m512 a = _mm512_setzero_ps();
              for (int i = 0; i < nv; ++i)
                a = _mm512_fmadd_ps(bf16x16_to_f32(qp + hh * d + 16 * i), bf16x16_to_f32(kp + hh * d + 16 * i), a);
              sc[(size_t)hh * S + j] = _mm512_reduce_add_ps(a) * inv_scale;
            }
          }
          for (int hh = 0; hh < G; ++hh) {
            float* s = &sc[(size_t)hh * S];
            float mx = -INFINITY;
            for (int j = 0; j <= lim; ++j) mx = s[j] > mx ? s[j] : mx;
            float sum = 0.f;
            for (int j = 0; j <= lim; ++j) {
              s[j] = expf(s[j] - mx);
              sum += s[j];
            }
            const float inv = 1.0f / sum;
            for (int j = 0; j <= lim; ++j) s[j] *= inv;
          }
          __m512 o[8][8];  // [head in group][16-wide slice of d] (d <= 128)
          for (int hh = 0; hh < G; ++hh)
            for (int i = 0; i < nv; ++i) o[hh][i] = _mm512_setzero_ps();
          for (int j = 0; j <= lim; ++j) {
            const lia_bf16* vp = vcache + (long)j * row + coff;
            if (j + 2 <= lim) _mm_prefetch((const char*)(vcache + (long)(j + 2) * row + coff), _MM_HINT_T0);
            for (int hh = 0; hh < G; ++hh) {
              __m512 p = _mm512_set1_ps(sc[(size_t)hh * S + j]);
              for (int i = 0; i < nv; ++i) o[hh][i] = _mm512_fmadd_ps(p, bf16x16_to_f32(vp + hh * d + 16 * i), o[hh][i]);
            }
          }
          lia_bf16* op = out + ((long)b * T + t) * hd + (long)g * G * d;
          for (int hh = 0; hh < G; ++hh)
            for (int i = 0; i < nv; ++i) {
              float tmp[16];
              _mm512_storeu_ps(tmp, o[hh][i]);
              for (int e = 0; e < 16; ++e) op[hh * d + 16 * i + e] = f32_to_bf16(tmp[e]);
            }
        }
      }
  }
  return LIA_OK;
}

// ------------------------------------------------------------------------------------------------
// TPP blocked layout (intel_extension_for_pytorch/nn/utils/_weight_prepack.py:19-63): element (n, k) of
// the row-major weight sits at blocked[n/16][k/64][(k%64)/2][n%16][k%2].
// ------------------------------------------------------------------------------------------------
extern "C" int lia_tpp_unblock(const lia_bf16* blocked, lia_bf16* plain, int N, int K) {
  if (!blocked || !plain) return LIA_ERR_MISSING;
  if (N % 16 || K % 64) {
    lia_set_error("lia_tpp_unblock: N %% 16 or K %% 64 != 0 (tpp_fallback shapes stay plain)");
    return LIA_ERR_INVALID;
  }
  const int kb = K / 64;
#pragma omp parallel for schedule(static)
  for (int nb = 0; nb < N / 16; ++nb)
    for (int kk = 0; kk < kb; ++kk) {
      const lia_bf16* blk = blocked + ((long)nb * kb + kk) * (32 * 16 * 2);
      for (int p = 0; p < 32; ++p)
        for (int r = 0; r < 16; ++r) {
          lia_bf16* dst = plain + (long)(nb * 16 + r) * K + kk * 64 + 2 * p;
          dst[0] = blk[(p * 16 + r) * 2];
          dst[1] = blk[(p * 16 + r) * 2 + 1];
        }
    }
  return LIA_OK;
}

extern "C" int lia_tpp_block(const lia_bf16* plain, lia_bf16* blocked, int N, int K) {
  if (!blocked || !plain) return LIA_ERR_MISSING;
  if (N % 16 || K % 64) {
    lia_set_error("lia_tpp_block: N %% 16 or K %% 64 != 0");
    return LIA_ERR_INVALID;
  }
  const int kb = K / 64;
#pragma omp parallel for schedule(static)
  for (int nb = 0; nb < N / 16; ++nb)
    for (int kk = 0; kk < kb; ++kk) {
      lia_bf16* blk = blocked + ((long)nb * kb + kk) * (32 * 16 * 2);
      for (int p = 0; p < 32; ++p)
        for (int r = 0; r < 16; ++r) {
          const lia_bf16* src = plain + (long)(nb * 16 + r) * K + kk * 64 + 2 * p;
          blk[(p * 16 + r) * 2] = src[0];
          blk[(p * 16 + r) * 2 + 1] = src[1];
        }
    }
  return LIA_OK;
}

// ------------------------------------------------------------------------------------------------
// NUMA / CXL tier.  Same exports and failure behaviour as lia/cxl/numa_alloc.c (NULL + a line on
// stderr); numa_alloc_interleave's node set comes from lia_numa_set_interleave_nodes / LIA_CXL_NODES
// instead of the hard-coded {2,3} (numa_alloc.c:80-81), falling back to {2,3} like the reference.
// ------------------------------------------------------------------------------------------------
static int g_nodes[64] = {2, 3};
static int g_n_nodes = 2;
static int g_nodes_from_env = 0;

extern "C" int lia_numa_available(void) { return numa_available() != -1; }

extern "C" int lia_numa_set_interleave_nodes(const int* nodes, int n) {
  if (!nodes || n <= 0 || n > 64) return LIA_ERR_INVALID;
  if (numa_available() == -1) {
    lia_set_error("NUMA is not available");
    return LIA_ERR_MEMORY;
  }
  for (int i = 0; i < n; ++i)
    if (nodes[i] < 0 || nodes[i] > numa_max_node()) {
      lia_set_error("lia_numa_set_interleave_nodes: node %d outside 0..%d", nodes[i], numa_max_node());
      return LIA_ERR_INVALID;
    }
  memcpy(g_nodes, nodes, n * sizeof(int));
  g_n_nodes = n;
  g_nodes_from_env = 1;
  return LIA_OK;
}

static void nodes_from_env(void) {
  if (g_nodes_from_env) return;
  g_nodes_from_env = 1;
  const char* e = getenv("LIA_CXL_NODES");
  if (!e || !*e) return;
  int n = 0;
  const char* p = e;
  while (*p && n < 64) {
    char* end;
    long v = strtol(p, &end, 10);
    if (end == p) break;
    g_nodes[n++] = (int)v;
    p = (*end == ',') ? end + 1 : end;
  }
  if (n > 0) g_n_nodes = n;
}

static void* alloc_on_mask(size_t size, const int* nodes, int n) {
  if (numa_available() == -1) {
    fprintf(stderr, "NUMA is not available\n");
    return NULL;
  }
  struct bitmask* mask = numa_bitmask_alloc(numa_max_node() + 1);
  for (int i = 0; i < n; ++i) {
    if (nodes[i] < 0 || nodes[i] > numa_max_node()) {
      fprintf(stderr, "Memory allocation failed: node %d does not exist (max node %d)\n", nodes[i], numa_max_node());
      numa_bitmask_free(mask);
      return NULL;
    }
    numa_bitmask_setbit(mask, nodes[i]);
  }
  struct bitmask* old_mask = numa_get_interleave_mask();
  numa_set_interleave_mask(mask);
  numa_set_strict(1);
  void* memory = numa_alloc(size);
  if (!memory) fprintf(stderr, "Memory allocation failed on the requested NUMA nodes\n");
  numa_set_interleave_mask(old_mask);
  numa_set_strict(0);
  numa_bitmask_free(old_mask);
  numa_bitmask_free(mask);
  return memory;
}

extern "C" void* numa_alloc_node(size_t size, int node) { return alloc_on_mask(size, &node, 1); }

extern "C" void* numa_alloc_interleave(size_t size) {
  nodes_from_env();
  return alloc_on_mask(size, g_nodes, g_n_nodes);
}

extern "C" void numa_free_node(void* memory, size_t size) { numa_free(memory, size); }

extern "C" void check_memory_node(void* memory, int num) {
  if (num <= 0) return;
  std::vector<int> status(num);
  std::vector<void*> pages(num);
  size_t page_size = (size_t)getpagesize();
  for (int i = 0; i < num; ++i) pages[i] = (char*)memory + (size_t)i * page_size;
  if (numa_move_pages(0, num, pages.data(), NULL, status.data(), 0) != 0) {
    perror("Error checking NUMA node of memory pages");
    return;
  }
  for (int i = 0; i < num; ++i) printf("Page %d is on node %d\n", i, status[i]);
}
