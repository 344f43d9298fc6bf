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
#include <atomic>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <vector>

#include "../../include/lia_hip.h"

extern "C" void lia_set_error(const char* fmt, ...);

// ------------------------------------------------------------------------------------------------
// One OpenMP region per host layer.  The launchers park the OpenMP team between bursts (OMP_WAIT_POLICY=PASSIVE /
// GOMP_SPINCOUNT=0 in bench.py and lia_amd/run.py: spinning workers would eat the CFS quota the copy-issuing main thread
// needs -- the headline loses 1-2 % with a spinning team).  With a parked team every `parallel` region AND every OpenMP
// barrier is a futex wake: nine regions per host layer cost ~0.45 ms of a 13 ms OPT-30B layer and 3 of the 4.3 ms an opt-125m
// token takes at batch 1 (configs[0]: 213 -> 810 tokens/s with a spinning team).  So the ops below are written as TEAM
// functions -- orphaned `omp for ... nowait` loops that bind to whatever region encloses them -- and a host layer runs them
// in ONE region, separated by this spinning barrier (it yields after a while: an oversubscribed box must not live-lock).
// ------------------------------------------------------------------------------------------------
struct LiaTeamBarrier {
  alignas(64) std::atomic<int> count{0};
  alignas(64) std::atomic<int> sense{0};
};
static inline void team_barrier(LiaTeamBarrier& b, int& local_sense) {
  const int n = omp_in_parallel() ? omp_get_num_threads() : 1;
  if (n == 1) return;
  local_sense ^= 1;
  if (b.count.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
    b.count.store(0, std::memory_order_relaxed);
    b.sense.store(local_sense, std::memory_order_release);
  } else {
    int spins = 0;
    while (b.sense.load(std::memory_order_acquire) != local_sense) {
      _mm_pause();
      if (++spins > 4096) { sched_yield(); spins = 0; }
    }
  }
}
static inline int team_size() { return omp_in_parallel() ? omp_get_num_threads() : omp_get_max_threads(); }

// a scratch block per thread that only grows (fp32 tiles of the linears, score rows of the attention)
// A worker whose block cannot grow gets nullptr: it raises the CALL's failure flag and SKIPS its share of the loop (every thread
// still meets every worksharing construct and team barrier, so the region ends normally); the entry point that opened the region
// returns LIA_ERR_MEMORY afterwards.  State is per call and per thread, never per process (r05): a HostCall lives on the stack of
// the exported entry point, the team's threads point at it for the length of the region (team_enter / team_leave), and the limit
// it carries is the CALLING thread's (lia_host_thread_scratch_limit: bytes per thread above which a request is refused, 0 = none
// -- a cap for memory-tight containers, and how the tests reach this path).  Two callers on two threads cannot see each other.
struct HostCall {
  std::atomic<int> failed{0};
  size_t limit = 0;
};
static thread_local size_t tl_scratch_limit = 0;
static thread_local HostCall* tl_call = nullptr;
struct HostCallScope {          // on the calling thread; an entry point reached from another one (layer -> attention) joins its call
  HostCall own;
  const bool outer;
  HostCallScope() : outer(tl_call == nullptr) {
    if (outer) { own.limit = tl_scratch_limit; tl_call = &own; }
  }
  ~HostCallScope() { if (outer) tl_call = nullptr; }
  HostCall* call() const { return tl_call; }
  int result(const char* who) {
    if (!tl_call->failed.exchange(0, std::memory_order_acq_rel)) return LIA_OK;
    lia_set_error("%s: out of host memory (a worker thread could not get its scratch block)", who);
    return LIA_ERR_MEMORY;
  }
};
static inline HostCall* team_enter(HostCall* c) { HostCall* prev = tl_call; tl_call = c; return prev; }
static inline void team_leave(HostCall* prev) { tl_call = prev; }
static inline float* thread_scratch(size_t floats) {
  struct Block { float* p = nullptr; size_t n = 0; ~Block() { free(p); } };
  static thread_local Block blk;
  HostCall* const call = tl_call;
  const size_t bytes = ((floats * sizeof(float)) + 63) & ~(size_t)63;
  if (call && call->limit && bytes > call->limit) { call->failed.store(1, std::memory_order_relaxed); return nullptr; }
  if (blk.n < floats) {
    free(blk.p);
    blk.p = (float*)aligned_alloc(64, bytes);
    blk.n = blk.p ? floats : 0;
    if (!blk.p) { if (call) call->failed.store(1, std::memory_order_relaxed); return nullptr; }
  }
  return blk.p;
}
extern "C" void lia_host_thread_scratch_limit(size_t bytes_per_thread) { tl_scratch_limit = bytes_per_thread; }

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
// the attention of one layer as a TEAM function (see the note above): every thread of the enclosing region calls it
static void host_attention_team(const lia_bf16* q, const lia_bf16* k, const lia_bf16* v, lia_bf16* kcache, lia_bf16* vcache,
                                lia_bf16* out, int B, int T, int pos0, int heads, int head_dim, int cache_batch, int b0) {
  const int d = head_dim;
  const long hd = (long)heads * d;
  const long row = (long)cache_batch * hd;
  const int S = pos0 + T;
  const float inv_scale = 1.0f / sqrtf((float)d);
  const int nv = d / 16;
  int G = 8;
  while (heads % G) G >>= 1;
  const int ngroups = heads / G;
  const int run_bytes = G * d * (int)sizeof(lia_bf16);
  // rows of look-ahead (0 / 2 / 4 / 6 / 8 / 12 measured: 6.0 / 5.6 / 3.1 / 2.5 / 2.2 / 4.4 ms at 16 threads), requested non-temporally
  // (into L1 / L2 / non-temporal: 2.22-2.26 / 1.90-2.25 / 1.99-2.11 ms -- LABNOTES r03; the A/B switches are gone)
  constexpr int PF = 8;
#define LIA_ATTN_PREFETCH(p) _mm_prefetch((p), _MM_HINT_NTA)
  {
    float* sc = thread_scratch((size_t)G * S);
#pragma omp for collapse(2) schedule(dynamic, 1) nowait
    for (int b = 0; b < B; ++b)
      for (int g = 0; g < ngroups; ++g) {
        if (!sc) continue;                     // no scratch: thread_scratch raised the flag, the caller reports LIA_ERR_MEMORY
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
            // the run of this (row, head group) is G*d*2 bytes at a stride far beyond a page: the hardware prefetcher does
            // not follow it, so every line of the run PF rows ahead is requested here (one line alone: 13 GB/s per thread)
            if (j + PF <= lim) {
              const char* pf = (const char*)(kcache + (long)(j + PF) * row + coff);
              for (int c = 0; c < run_bytes; c += 64) LIA_ATTN_PREFETCH(pf + c);
            }
            for (int hh = 0; hh < G; ++hh) {
              __m512 a = _mm512_setzero_ps();
              for (int i = 0; i < nv; ++i)
                a = _mm512_fmadd_ps(bf16x16_to_f32(qp + hh * d + 16 * i), bf16x16_to_f32(kp + hh * d + 16 * i), a);
              sc[(size_t)hh * S + j] = _mm512_reduce_add_ps(a) * inv_scale;
            }
          }
          for (int hh = 0; hh < G; ++hh) {
            float* s = sc + (size_t)hh * S;
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
            if (j + PF <= lim) {
              const char* pf = (const char*)(vcache + (long)(j + PF) * row + coff);
              for (int c = 0; c < run_bytes; c += 64) LIA_ATTN_PREFETCH(pf + c);
            }
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
}

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
  if (head_dim > 128) {   // the per-head accumulators are a fixed 8 x 8 zmm block (OPT / Llama heads are 64 or 128 wide)
    lia_set_error("lia_host_attention: head_dim %d > 128 is not supported", head_dim);
    return LIA_ERR_INVALID;
  }
  if (n_threads <= 0) n_threads = omp_get_max_threads();
  HostCallScope scope;
  HostCall* const call = scope.call();
#pragma omp parallel num_threads(n_threads)
  {
    HostCall* const prev = team_enter(call);
    host_attention_team(q, k, v, kcache, vcache, out, B, T, pos0, heads, head_dim, cache_batch, b0);
    team_leave(prev);
  }
  return scope.result("lia_host_attention");
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

// ------------------------------------------------------------------------------------------------
// policy 1: the whole decoder layer on the host cores ("compute everything on CPU", lia/modeling_opt.py:1168;
// the reference runs it through IPEX: tpp_linear_bias/_relu/_add, csrc/cpu/tpp/kernels/TPPGEMMKrnl.h:89-176,
// 671-765, 858-951 + the masked MHA kernel).  AVX-512-BF16 (vdpbf16ps) dot products on row-major bf16
// weights; CPU rounding semantics: the bias is added to the fp32 accumulator before the single rounding
// (tpp_linear_bias), "+ residual" is a second bf16 op (_IPEXlinearAddRef).
// ------------------------------------------------------------------------------------------------
static inline float bf16_to_f32(lia_bf16 v) {
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

static void host_layernorm_team(const lia_bf16* x, const lia_bf16* g, const lia_bf16* b, lia_bf16* y, long rows, int H, float eps) {
#pragma omp for schedule(static) nowait
  for (long r = 0; r < rows; ++r) {
    const lia_bf16* xr = x + r * (long)H;
    lia_bf16* yr = y + r * (long)H;
    __m512 s = _mm512_setzero_ps();
    for (int i = 0; i < H; i += 16) s = _mm512_add_ps(s, bf16x16_to_f32(xr + i));
    const float mean = _mm512_reduce_add_ps(s) / (float)H;
    const __m512 vm = _mm512_set1_ps(mean);
    __m512 q = _mm512_setzero_ps();
    for (int i = 0; i < H; i += 16) {
      __m512 d = _mm512_sub_ps(bf16x16_to_f32(xr + i), vm);
      q = _mm512_fmadd_ps(d, d, q);
    }
    const float rstd = 1.0f / sqrtf(_mm512_reduce_add_ps(q) / (float)H + eps);
    for (int i = 0; i < H; ++i) yr[i] = f32_to_bf16((bf16_to_f32(xr[i]) - mean) * rstd * bf16_to_f32(g[i]) + bf16_to_f32(b[i]));
  }
}
static void host_layernorm(const lia_bf16* x, const lia_bf16* g, const lia_bf16* b, lia_bf16* y, long rows, int H, float eps) {
#pragma omp parallel
  host_layernorm_team(x, g, b, y, rows, H, eps);
}

#if defined(__AVX512BF16__)
#define LIA_HAVE_DPBF16 1
static inline __m512 dp32(__m512 acc, const lia_bf16* a, const lia_bf16* b) {
  return _mm512_dpbf16_ps(acc, (__m512bh)_mm512_loadu_si512((const void*)a), (__m512bh)_mm512_loadu_si512((const void*)b));
}
#else
#define LIA_HAVE_DPBF16 0
static inline __m512 dp32(__m512 acc, const lia_bf16* a, const lia_bf16* b) {
  acc = _mm512_fmadd_ps(bf16x16_to_f32(a), bf16x16_to_f32(b), acc);
  return _mm512_fmadd_ps(bf16x16_to_f32(a + 16), bf16x16_to_f32(b + 16), acc);
}
#endif

// sums of four 16-lane accumulators as one xmm (a0, a1, a2, a3): unpack/add transposes instead of four horizontal reductions
static inline __m128 reduce4(__m512 a0, __m512 a1, __m512 a2, __m512 a3) {
  const __m512 s01 = _mm512_add_ps(_mm512_unpacklo_ps(a0, a1), _mm512_unpackhi_ps(a0, a1));   // per 128-bit lane: a0[0]+a0[2], a1[0]+a1[2], a0[1]+a0[3], a1[1]+a1[3]
  const __m512 s23 = _mm512_add_ps(_mm512_unpacklo_ps(a2, a3), _mm512_unpackhi_ps(a2, a3));
  const __m512 s = _mm512_add_ps(_mm512_castpd_ps(_mm512_unpacklo_pd(_mm512_castps_pd(s01), _mm512_castps_pd(s23))),
                                 _mm512_castpd_ps(_mm512_unpackhi_pd(_mm512_castps_pd(s01), _mm512_castps_pd(s23))));   // lane: sum4(a0), sum4(a1), sum4(a2), sum4(a3)
  const __m256 h = _mm256_add_ps(_mm512_castps512_ps256(s), _mm512_extractf32x8_ps(s, 1));
  return _mm_add_ps(_mm256_castps256_ps128(h), _mm256_extractf128_ps(h, 1));
}

// f32_to_bf16 on 16 lanes, bit for bit (round to nearest even, NaNs quieted), as fp32 values again
static inline __m512 round_bf16_x16(__m512 t) {
  const __m512i u = _mm512_castps_si512(t);
  const __mmask16 nan = _mm512_cmpgt_epu32_mask(_mm512_and_si512(u, _mm512_set1_epi32(0x7fffffff)), _mm512_set1_epi32(0x7f800000));
  __m512i r = _mm512_add_epi32(u, _mm512_add_epi32(_mm512_set1_epi32(0x7fff), _mm512_and_si512(_mm512_srli_epi32(u, 16), _mm512_set1_epi32(1))));
  r = _mm512_and_si512(r, _mm512_set1_epi32((int)0xffff0000u));
  const __m512i q = _mm512_and_si512(_mm512_or_si512(u, _mm512_set1_epi32(0x00400000)), _mm512_set1_epi32((int)0xffff0000u));
  return _mm512_castsi512_ps(_mm512_mask_blend_epi32(nan, r, q));
}
static inline void store_bf16_x16(lia_bf16* p, __m512 rounded, __mmask16 m) {
  _mm256_mask_storeu_epi16((void*)p, m, _mm512_cvtepi32_epi16(_mm512_srli_epi32(_mm512_castps_si512(rounded), 16)));
}
static inline __m512 load_bf16_x16(const lia_bf16* p, __mmask16 m) {
  return _mm512_castsi512_ps(_mm512_slli_epi32(_mm512_cvtepu16_epi32(_mm256_maskz_loadu_epi16(m, (const void*)p)), 16));
}

template <int RN>   // RN weight rows x 4 activation rows per register block: 4 (16 accumulators) or 6 (24 + 4 x rows + 1 w row = 29 zmm)
static void host_linear_skinny_team_t(const lia_bf16* x, const lia_bf16* w, const lia_bf16* bias, const lia_bf16* residual, lia_bf16* y,
                                      int M, int N, int K, int relu) {
  constexpr int RB = 4;
  // rows per tile: 8 ... 16 register blocks, chosen so the tiles deal out evenly over the threads (N = 7168 in 96-row tiles is
  // 75 tiles = 4.7 rounds of 16 threads, the last one a third empty; in 90-row tiles it is 80 = 5 rounds exactly)
  int NT = 16 * RN;
  {
    const int T = team_size(), blocks = (N + RN - 1) / RN;
    long best = -1;
    for (int per = 16; per >= 8; --per) {
      const int tiles = (blocks + per - 1) / per;
      const long span = (long)((tiles + T - 1) / T) * per;                 // makespan in register blocks
      if (best < 0 || span < best) { best = span; NT = per * RN; }
    }
  }
  // K-chunk length, and the NEXT weight rows' chunk prefetched into L2 while this one is multiplied (measured against no
  // prefetch / into L1 and chunks of 1024 / 4096 with tools/host_linear_bench.py, LABNOTES r03).  The rows of a tile are K * 2
  // bytes apart, so every block of rows starts on cold lines the hardware prefetcher has not seen; its 16-24 KB are spread over
  // the block's m-loop.
  constexpr int KC = 2048;
  constexpr int PF = 1;
  const int ntiles = (N + NT - 1) / NT;
  const int mblocks = (M + RB - 1) / RB;
  const size_t crow = (size_t)NT + 8;                                      // C rows padded: an edge block's xmm stores stay inside
  {
    float* C = thread_scratch((size_t)((M + 3) & ~3) * crow);
#pragma omp for schedule(dynamic, 1) nowait
    for (int tile = 0; tile < ntiles; ++tile) {
      if (!C) continue;                        // no scratch: see thread_scratch
      const int nt0 = tile * NT, ntn = N - nt0 < NT ? N - nt0 : NT;
      memset(C, 0, (size_t)((M + 3) & ~3) * crow * sizeof(float));
      for (int k0 = 0; k0 < K; k0 += KC) {
        const int kl = K - k0 < KC ? K - k0 : KC;
        const int lines = kl / 32, lp = (lines + mblocks - 1) / mblocks;     // 64-byte lines per row chunk, per m-block
        for (int nb = 0; nb < ntn; nb += RN) {
          const int nr = ntn - nb < RN ? ntn - nb : RN;
          const lia_bf16* wr = w + (long)(nt0 + nb) * K + k0;
          // what this thread reads next: the following rows of the chunk, or the tile's first rows in the next chunk
          const lia_bf16* wn = nullptr;
          int wn_rows = 0;
          if (PF) {
            if (nb + RN < ntn) { wn = wr + (long)RN * K; wn_rows = ntn - nb - RN < RN ? ntn - nb - RN : RN; }
            else if (k0 + KC < K) { wn = w + (long)nt0 * K + k0 + KC; wn_rows = ntn < RN ? ntn : RN; }
          }
          for (int m0 = 0; m0 < M; m0 += RB) {
            const int mr = M - m0 < RB ? M - m0 : RB;
            const lia_bf16* xr = x + (long)m0 * K + k0;
            if (wn) {
              const int l0 = (m0 / RB) * lp, l1 = l0 + lp < lines ? l0 + lp : lines;
              for (int j = 0; j < wn_rows; ++j)
                for (int l = l0; l < l1; ++l) {
                  _mm_prefetch((const char*)(wn + (long)j * K) + 64 * l, _MM_HINT_T1);
                }
            }
            __m512 acc[RB][RN];
            for (int i = 0; i < RB; ++i)
              for (int j = 0; j < RN; ++j) acc[i][j] = _mm512_setzero_ps();
            if (mr == RB && nr == RN) {
              for (int k = 0; k < kl; k += 32) {
#pragma GCC unroll 8
                for (int j = 0; j < RN; ++j)
#pragma GCC unroll 4
                  for (int i = 0; i < RB; ++i) acc[i][j] = dp32(acc[i][j], xr + i * (long)K + k, wr + j * (long)K + k);
              }
            } else if (nr == RN && mr == 1) {
              // batch 1 (configs[0], opt-125m 1/1): a GEMV -- constant trip counts keep the RN accumulators in registers
              for (int k = 0; k < kl; k += 32) {
#pragma GCC unroll 8
                for (int j = 0; j < RN; ++j) acc[0][j] = dp32(acc[0][j], xr + k, wr + j * (long)K + k);
              }
            } else if (nr == RN && mr == 2) {
              for (int k = 0; k < kl; k += 32) {
#pragma GCC unroll 8
                for (int j = 0; j < RN; ++j)
#pragma GCC unroll 2
                  for (int i = 0; i < 2; ++i) acc[i][j] = dp32(acc[i][j], xr + i * (long)K + k, wr + j * (long)K + k);
              }
            } else {
              for (int k = 0; k < kl; k += 32)
                for (int i = 0; i < mr; ++i)
                  for (int j = 0; j < nr; ++j) acc[i][j] = dp32(acc[i][j], xr + i * (long)K + k, wr + j * (long)K + k);
            }
            // lanes of an edge block beyond mr / nr hold zeros; C's rows are padded so the xmm stores stay inside
            for (int i = 0; i < mr; ++i) {
              float* c = C + (m0 + i) * crow + nb;
              _mm_storeu_ps(c, _mm_add_ps(_mm_loadu_ps(c), reduce4(acc[i][0], acc[i][1], acc[i][2], acc[i][3])));
              if (RN > 4) {
                const __m512 z = _mm512_setzero_ps();
                _mm_storeu_ps(c + 4, _mm_add_ps(_mm_loadu_ps(c + 4), reduce4(acc[i][RN > 4 ? 4 : 0], acc[i][RN > 5 ? 5 : 0], z, z)));
              }
            }
          }
        }
      }
      for (int m = 0; m < M; ++m)
        for (int j = 0; j < ntn; j += 16) {
          const __mmask16 msk = ntn - j >= 16 ? (__mmask16)0xffff : (__mmask16)((1u << (ntn - j)) - 1);
          __m512 t = _mm512_maskz_loadu_ps(msk, C + m * crow + j);
          if (bias) t = _mm512_add_ps(t, load_bf16_x16(bias + nt0 + j, msk));
          t = round_bf16_x16(t);
          if (relu) t = _mm512_mask_blend_ps(_mm512_cmp_ps_mask(t, _mm512_setzero_ps(), _CMP_LT_OQ), t, _mm512_setzero_ps());
          if (residual) t = round_bf16_x16(_mm512_add_ps(load_bf16_x16(residual + (long)m * N + nt0 + j, msk), t));
          store_bf16_x16(y + (long)m * N + nt0 + j, t, msk);
        }
    }
  }
}

// Decode-sized M (<= 256): the weights are streamed once and x (M x K, ~0.9 MB at M = 64, K = 7168) must stay close
// to the core.  Tiles of 64 (96) weight rows x K-chunks of <= 2048: per chunk the x slice (M x 4 KB) and the tile's weight
// slice (256 KB) both sit in L2, a 4 x 4 (4 x 6) zmm block runs over the chunk, and the chunk sums land in a thread-local
// fp32 tile C[M][64] (L1).  With the whole K in one pass x overflows L2 and every 4 weight rows re-read it from L3.
static void host_linear_skinny_team(const lia_bf16* x, const lia_bf16* w, const lia_bf16* bias, const lia_bf16* residual, lia_bf16* y,
                                    int M, int N, int K, int relu) {
  host_linear_skinny_team_t<6>(x, w, bias, residual, y, M, N, K, relu);      // 4 x 6 register blocks: 14.8 -> 14.2 ms per OPT-30B decode layer against 4 x 4
}
static void host_linear_skinny(const lia_bf16* x, const lia_bf16* w, const lia_bf16* bias, const lia_bf16* residual, lia_bf16* y,
                               int M, int N, int K, int relu) {
  HostCall* const call = tl_call;          // (the exported entry point's HostCallScope)
#pragma omp parallel
  {
    HostCall* const prev = team_enter(call);
    host_linear_skinny_team(x, w, bias, residual, y, M, N, K, relu);
    team_leave(prev);
  }
}

// y[M,N] = act(x[M,K] . w[N,K]^T + bias) [+ residual], K % 32 == 0.  Any M: more than 256 rows (policy 1's prefill) go through the
// SAME register-blocked kernel as a decode step, 256 rows at a time inside one parallel region (r05: a separate 4 x 4 kernel for
// M > 256 ran at 2.6 TFLOP/s on 16 Zen 5 cores where the decode kernel reaches 6.3-7.5; a panel's x slice is the 1 MB the kernel's
// K-chunking was tuned for, and the weights' re-read per panel -- 64 x 411 MB for OPT-30B's fc1 at B 64 x T 256 -- is 27 GB/s of
// DRAM traffic beside ~1 s of arithmetic).  Prefill rows and decode rows now see the same arithmetic in the same order.
static void host_linear(const lia_bf16* x, const lia_bf16* w, const lia_bf16* bias, const lia_bf16* residual, lia_bf16* y,
                        long M, int N, int K, int relu) {
  if (M <= 256) { host_linear_skinny(x, w, bias, residual, y, (int)M, N, K, relu); return; }
  HostCall* const call = tl_call;
#pragma omp parallel
  {
    HostCall* const prev = team_enter(call);
    for (long mp = 0; mp < M; mp += 256) {
      const int rows = (int)(M - mp < 256 ? M - mp : 256);
      // (the kernel's worksharing loop is `nowait`: a thread that has finished its tiles of this panel starts on the next one)
      host_linear_skinny_team(x + mp * (long)K, w, bias, residual ? residual + mp * (long)N : nullptr, y + mp * (long)N, rows, N, K, relu);
    }
    team_leave(prev);
  }
}

// The linears are compiled for AVX-512-BF16 (vdpbf16ps); on a host without it the first such instruction would be a
// SIGILL, so the exported entry points check CPUID once and fail with a message instead.
static int host_isa_ok(const char* who) {
#if LIA_HAVE_DPBF16
  static const int ok = __builtin_cpu_supports("avx512bf16") ? 1 : 0;
  if (!ok) {
    lia_set_error("%s: this library was built for AVX-512-BF16 hosts (vdpbf16ps); the CPU does not support it", who);
    return LIA_ERR_INVALID;
  }
#else
  (void)who;
#endif
  return LIA_OK;
}

extern "C" int lia_host_layernorm(const lia_bf16* x, const lia_bf16* g, const lia_bf16* b, lia_bf16* y, long rows, int H,
                                  float eps, int n_threads) {
  if (!x || !g || !b || !y) return LIA_ERR_MISSING;
  if (rows < 0 || H <= 0 || H % 16) { lia_set_error("lia_host_layernorm: H=%d must be a multiple of 16", H); return LIA_ERR_INVALID; }
  if (n_threads > 0) omp_set_num_threads(n_threads);
  host_layernorm(x, g, b, y, rows, H, eps);
  return LIA_OK;
}

extern "C" int lia_host_linear(const lia_bf16* x, const lia_bf16* w, const lia_bf16* bias, const lia_bf16* residual,
                               lia_bf16* y, long M, int N, int K, int relu, int n_threads) {
  if (!x || !w || !y) return LIA_ERR_MISSING;
  if (M < 0 || N <= 0 || K <= 0 || K % 32) { lia_set_error("lia_host_linear: K=%d must be a multiple of 32", K); return LIA_ERR_INVALID; }
  if (int rc = host_isa_ok("lia_host_linear")) return rc;
  if (n_threads > 0) omp_set_num_threads(n_threads);
  HostCallScope scope;
  host_linear(x, w, bias, residual, y, M, N, K, relu);
  return scope.result("lia_host_linear");
}

// the intermediates of a host layer (ln, q, k, v, attention output, h1: M x H each; f1: M x F), one growing block per CALLING thread
struct LayerScratch { lia_bf16* p = nullptr; size_t n = 0; ~LayerScratch() { free(p); } };
static LayerScratch& layer_scratch(size_t need) {
  static thread_local LayerScratch scratch;
  if (scratch.n < need) {
    free(scratch.p);
    scratch.p = (lia_bf16*)aligned_alloc(64, ((need * sizeof(lia_bf16)) + 63) & ~(size_t)63);
    scratch.n = scratch.p ? need : 0;
  }
  return scratch;
}

// one decode-sized layer inside an enclosing parallel region (every thread calls it); sc: 6 * mh + M * F bf16 of scratch
static void host_layer_team(const lia_layer_desc* d, const lia_bf16* const* W, const lia_bf16* x, lia_bf16* y, lia_bf16* kcache,
                            lia_bf16* vcache, int cache_batch, int B, int T, int pos0, int b0, lia_bf16* sc, size_t mh,
                            LiaTeamBarrier& bar, int& sense) {
  const int H = d->hidden, F = d->ffn, heads = d->heads, M = B * T;
  lia_bf16 *ln = sc, *q = ln + mh, *k = q + mh, *v = k + mh, *ao = v + mh, *h1 = ao + mh, *f1 = h1 + mh;
  host_layernorm_team(x, W[0], W[1], ln, M, H, d->ln_eps);
  team_barrier(bar, sense);
  host_linear_skinny_team(ln, W[4], W[5], nullptr, k, M, H, H, 0);        // (nowait loops: the three projections overlap at their tails)
  host_linear_skinny_team(ln, W[6], W[7], nullptr, v, M, H, H, 0);
  host_linear_skinny_team(ln, W[2], W[3], nullptr, q, M, H, H, 0);
  team_barrier(bar, sense);
  host_attention_team(q, k, v, kcache, vcache, ao, B, T, pos0, heads, H / heads, cache_batch, b0);
  team_barrier(bar, sense);
  host_linear_skinny_team(ao, W[8], W[9], x, h1, M, H, H, 0);
  team_barrier(bar, sense);
  host_layernorm_team(h1, W[10], W[11], ln, M, H, d->ln_eps);
  team_barrier(bar, sense);
  host_linear_skinny_team(ln, W[12], W[13], nullptr, f1, M, F, H, 1);
  team_barrier(bar, sense);
  host_linear_skinny_team(f1, W[14], W[15], h1, y, M, H, F, 0);
}

// One decoder layer on the host: OPTDecoderLayer_forward with gpu_linear = gpu_attn = False (decoder.py:191-193,
// 206, 248-250, 276, 286-287, 312-315; attentions.py:365-376, 401-440).  weights: 16 HOST pointers in create_buffer
// order, row-major.  x, y: host [B,T,H]; cache: host [smax][cache_batch][h][d], rows pos0.. appended.
extern "C" int lia_host_layer_forward(const lia_layer_desc* d, const void* const weights[16], const lia_bf16* x, lia_bf16* y,
                                      lia_bf16* kcache, lia_bf16* vcache, int smax, int cache_batch, int B, int T, int pos0,
                                      int b0, int n_threads) {
  if (!d || !weights || !x || !y || !kcache || !vcache) { lia_set_error("lia_host_layer_forward: NULL argument"); return LIA_ERR_MISSING; }
  for (int i = 0; i < 16; ++i)
    if (!weights[i]) { lia_set_error("lia_host_layer_forward: weights[%d] is NULL", i); return LIA_ERR_MISSING; }
  const int H = d->hidden, F = d->ffn, heads = d->heads;
  if (H <= 0 || heads <= 0 || H % heads || (H / heads) % 16 || (H / heads) > 128 || H % 32 || F <= 0 || F % 32 || B <= 0 || T <= 0 || pos0 < 0 ||
      pos0 + T > smax || b0 < 0 || b0 + B > cache_batch) {
    lia_set_error("lia_host_layer_forward: bad shape H=%d heads=%d F=%d B=%d T=%d pos0=%d smax=%d", H, heads, F, B, T, pos0, smax);
    return LIA_ERR_INVALID;
  }
  if (int rc = host_isa_ok("lia_host_layer_forward")) return rc;
  if (n_threads > 0) omp_set_num_threads(n_threads);
  const lia_bf16* const* W = (const lia_bf16* const*)weights;
  const long M = (long)B * T;
  HostCallScope scope;
  // the intermediates live in one scratch block per calling thread that only ever grows: std::vector zero-filled 9 MB per
  // decode call at the OPT-30B shape (single-threaded, ~3 % of the layer) and paid the page faults again after every free
  const size_t mh = ((size_t)M * H + 31) & ~(size_t)31, mf = ((size_t)M * F + 31) & ~(size_t)31, need = 6 * mh + mf;
  LayerScratch& scratch = layer_scratch(need);
  if (!scratch.p) { lia_set_error("lia_host_layer_forward: out of host memory (%zu bytes of scratch)", need * sizeof(lia_bf16)); return LIA_ERR_MEMORY; }
  // (a prefill-sized call -- policy 1 over B * T rows -- gives its block back on return: see the end of the function)
  lia_bf16 *ln = scratch.p, *q = ln + mh, *k = q + mh, *v = k + mh, *ao = v + mh, *h1 = ao + mh, *f1 = h1 + mh;
  if (M <= 256) {
    // decode: the whole layer in ONE parallel region, the ops separated by a spinning barrier (see LiaTeamBarrier)
    LiaTeamBarrier bar;
    lia_bf16* const sc = scratch.p;          // (thread_local: the workers must see the CALLER's block, not their own)
    HostCall* const call = scope.call();
#pragma omp parallel
    {
      HostCall* const prev = team_enter(call);
      int sense = 0;
      host_layer_team(d, W, x, y, kcache, vcache, cache_batch, B, T, pos0, b0, sc, mh, bar, sense);
      team_leave(prev);
    }
    return scope.result("lia_host_layer_forward");
  }
  host_layernorm(x, W[0], W[1], ln, M, H, d->ln_eps);
  host_linear(ln, W[4], W[5], nullptr, k, M, H, H, 0);
  host_linear(ln, W[6], W[7], nullptr, v, M, H, H, 0);
  host_linear(ln, W[2], W[3], nullptr, q, M, H, H, 0);
  int rc = lia_host_attention(q, k, v, kcache, vcache, ao, B, T, pos0, heads, H / heads, cache_batch, b0, n_threads);
  if (rc) return rc;                                                       // (the block stays with the thread; the next call reuses it)
  host_linear(ao, W[8], W[9], x, h1, M, H, H, 0);
  host_layernorm(h1, W[10], W[11], ln, M, H, d->ln_eps);
  host_linear(ln, W[12], W[13], nullptr, f1, M, F, H, 1);
  host_linear(f1, W[14], W[15], h1, y, M, H, F, 0);
  if (scratch.n * sizeof(lia_bf16) > ((size_t)64 << 20)) { free(scratch.p); scratch.p = nullptr; scratch.n = 0; }
  return scope.result("lia_host_layer_forward");
}

// n_layers consecutive decode-sized layers (policy 1's decode step: every layer on the host) in ONE parallel region: the hidden
// state ping-pongs between x and y (the result is in x when n_layers is even, in y when it is odd -- the return value says
// which: 0 = x, 1 = y, negative = error).  With a parked OpenMP team every region costs a futex wake per thread; an opt-125m
// token at batch 1 is twelve 0.1 ms layers (configs[0]).
extern "C" int lia_host_layers_forward(const lia_layer_desc* d, int n_layers, const void* const* weights, lia_bf16* x, lia_bf16* y,
                                       void* const* kcaches, void* const* vcaches, int smax, int cache_batch, int B, int T,
                                       int pos0, int b0, int n_threads) {
  if (!d || !weights || !x || !y || !kcaches || !vcaches || n_layers <= 0) { lia_set_error("lia_host_layers_forward: NULL argument"); return LIA_ERR_MISSING; }
  for (long i = 0; i < 16L * n_layers; ++i)
    if (!weights[i]) { lia_set_error("lia_host_layers_forward: weights[%ld][%ld] is NULL", i / 16, i % 16); return LIA_ERR_MISSING; }
  for (int l = 0; l < n_layers; ++l)
    if (!kcaches[l] || !vcaches[l]) { lia_set_error("lia_host_layers_forward: cache %d is NULL", l); return LIA_ERR_MISSING; }
  const int H = d->hidden, F = d->ffn, heads = d->heads;
  const long M = (long)B * T;
  if (H <= 0 || heads <= 0 || H % heads || (H / heads) % 16 || (H / heads) > 128 || H % 32 || F <= 0 || F % 32 || B <= 0 || T <= 0 || pos0 < 0 ||
      pos0 + T > smax || b0 < 0 || b0 + B > cache_batch || M > 256) {
    lia_set_error("lia_host_layers_forward: bad shape H=%d heads=%d F=%d B=%d T=%d pos0=%d smax=%d (B * T <= 256: decode-sized steps only)",
                  H, heads, F, B, T, pos0, smax);
    return LIA_ERR_INVALID;
  }
  if (int rc = host_isa_ok("lia_host_layers_forward")) return rc;
  if (n_threads > 0) omp_set_num_threads(n_threads);
  const size_t mh = ((size_t)M * H + 31) & ~(size_t)31, mf = ((size_t)M * F + 31) & ~(size_t)31;
  lia_bf16* const sc = layer_scratch(6 * mh + mf).p;     // (fetched by the caller: thread_local)
  if (!sc) { lia_set_error("lia_host_layers_forward: out of host memory"); return LIA_ERR_MEMORY; }
  LiaTeamBarrier bar;
  HostCallScope scope;
  HostCall* const call = scope.call();
#pragma omp parallel
  {
    HostCall* const prev = team_enter(call);
    int sense = 0;
    lia_bf16 *in = x, *out = y;
    for (int l = 0; l < n_layers; ++l) {
      host_layer_team(d, (const lia_bf16* const*)(weights + 16L * l), in, out, (lia_bf16*)kcaches[l], (lia_bf16*)vcaches[l], cache_batch, B, T,
                      pos0, b0, sc, mh, bar, sense);
      team_barrier(bar, sense);
      lia_bf16* t = in; in = out; out = t;
    }
    team_leave(prev);
  }
  if (int rc = scope.result("lia_host_layers_forward")) return rc;
  return n_layers & 1;
}

// memcpy by a team (the streamer's staging of pageable sources, lia_api.hip::staged_copy): 1 MiB pieces dealt out statically;
// n_threads <= 0 = at most 8 (a caller that never said how many CPUs it owns: the box may show 256 and grant 16).  One thread moves
// ~10 GB/s, the link takes 53.
extern "C" void lia_host_parallel_memcpy(void* dst, const void* src, size_t bytes, int n_threads) {
  constexpr size_t PIECE = (size_t)1 << 20;
  const long pieces = (long)((bytes + PIECE - 1) / PIECE);
  if (pieces <= 1) { memcpy(dst, src, bytes); return; }
  int nt = n_threads > 0 ? n_threads : (omp_get_max_threads() < 8 ? omp_get_max_threads() : 8);
  if (nt > pieces) nt = (int)pieces;
#pragma omp parallel for schedule(static) num_threads(nt)
  for (long i = 0; i < pieces; ++i) {
    const size_t off = (size_t)i * PIECE;
    memcpy((char*)dst + off, (const char*)src + off, bytes - off < PIECE ? bytes - off : PIECE);
  }
}

// 1 when the vdpbf16ps inner loops are compiled in AND this CPU executes them
extern "C" int lia_host_has_avx512_bf16(void) {
#if LIA_HAVE_DPBF16
  return __builtin_cpu_supports("avx512bf16") ? 1 : 0;
#else
  return 0;
#endif
}
