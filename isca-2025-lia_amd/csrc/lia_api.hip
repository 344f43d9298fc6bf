// C ABI of liblia_hip.so (include/lia_hip.h): context, the decoder-layer operator, the sub-layer ops,
// the weight streamer and pinned host memory.  Kernels live in lia_gemm.hip / lia_attention.hip /
// lia_elementwise.hip, host-side cooperative code in lia_host.cpp.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "../../include/lia_hip.h"
#include "lia_common.h"

// kernels' host launchers
extern "C" size_t lia_gemm_workspace_bytes(int M, int N);
extern "C" int lia_gemm_launch(const bf16_t* x, long ldx, const bf16_t* W, long ldw, int M, int N, int K,
                               const LiaEpilogue* ep, const LiaOutMap* om, float* workspace, size_t workspace_bytes,
                               LiaGemmOpts* opts, int force_split, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, int* regime,
                               const LiaPost* post, int* post_done);
extern "C" void lia_layernorm_launch(const bf16_t* x, long ldx, const bf16_t* g, const bf16_t* b, bf16_t* y, long ldy,
                                     long rows, int H, float eps, hipStream_t st);
extern "C" void lia_embed_launch(const int64_t* ids, const bf16_t* tok, const bf16_t* pos, bf16_t* y, int B, int T,
                                 int past_len, int H, hipStream_t st);
extern "C" void lia_argmax_launch(const bf16_t* logits, int64_t* out, int B, int vocab, int suppress, hipStream_t st);
extern "C" void lia_blit_launch(void* dst, const void* src, size_t bytes, hipStream_t st);
extern "C" int lia_attn_prefill_launch(const bf16_t* q, long ldq, const bf16_t* kc, const bf16_t* vc, bf16_t* out, long ldo,
                                       int B, int T, int heads, int kv_heads, int d, int Bc, int b0, int post_scale,
                                       hipStream_t st);
extern "C" int lia_attn_decode_launch(const bf16_t* q, long ldq, const bf16_t* kc, const bf16_t* vc, bf16_t* out, long ldo,
                                      int B, int S, int heads, int kv_heads, int d, int Bc, int b0, int post_scale,
                                      hipStream_t st);

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

extern "C" void lia_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* lia_last_error(void) { return g_err; }
extern "C" const char* lia_version(void) { return "lia_hip 0.1 (gfx950)"; }

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      lia_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return _e == hipErrorOutOfMemory ? LIA_ERR_MEMORY : LIA_ERR_HIP;                      \
    }                                                                                        \
  } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct lia_ctx {
  int device;
  hipStream_t compute;
  hipStream_t d2h;           // K/V delivery to the host cache (policy 0)
  char* ws;
  size_t ws_bytes;
  hipEvent_t slab_ready[2];  // recorded on compute when a K/V slab may be copied out
  hipEvent_t slab_done[2];   // recorded on d2h when the slab has been delivered
  bool slab_used[2];
  int slab_next;
  char* host_stage;          // pinned: q|k|v and attention output of the policy-2 round trip
  size_t host_stage_bytes;
  int host_threads;
  long last_rows, last_slab_rows;  // workspace layout of the previous layer call
  // live kernel timing for bench.py's roofline object (lia_prof_*)
  bool prof_on;
  std::vector<hipEvent_t>* prof_events;
  struct ProfRec { int regime; double bytes, flops; };
  std::vector<ProfRec>* prof_recs;
  size_t prof_cap;
  int prof_stride;      // every prof_stride-th GEMM launch is bracketed (an event pair costs ~2 x 6 us of stream idle)
  long prof_seen;
  long prof_host_attn_calls;
  double prof_host_attn_ms;
  LiaGemmOpts gemm_opts;     // per-context switches / counters of the GEMM launcher (lia_ctx_set_option, lia_ctx_get_counter)
  std::vector<hipEvent_t>* deliver_events;   // lia_kv_deliver tickets (events on the d2h stream), recycled round-robin
  std::vector<char>* deliver_pending;
  hipEvent_t deliver_t0, deliver_t1;         // timing events on the d2h stream around one batch of deliveries (lia_kv_deliver_batch_ms)
  bool deliver_batch_open;
  // cross-layer chaining (lia_ctx_chain_next_norm): the combine of a layer call's last GEMM also normalises the output row with
  // the NEXT layer's first-norm weights into the workspace's norm buffer; the next call finds it there and skips its first norm
  const bf16_t *chain_g, *chain_b;
  bool chain_armed;                 // one-shot, consumed by the next layer call
  const void* normed_src;           // the y whose norm sits in normed_buf ...
  const void* normed_buf;
  long normed_rows;                 // ... for this many rows of this width
  int normed_h;
  // LIA_SERIALIZE=1 (debug): ONE stream for everything -- the K/V delivery stream IS the compute stream, and a streamer created
  // on this context copies and decodes on it too, so every event handshake is trivially ordered.  Results must not change; if
  // they do, a cross-stream ordering is missing (the role torch.cuda.synchronize() plays at lia/modeling_opt.py:1339,1528).
  bool serialized;
  char* hidden_tmp;           // a third hidden-state buffer for lia_decode_layers (a layer never writes the buffer it reads)
  size_t hidden_tmp_bytes;
};

extern "C" int lia_ctx_chain_next_norm(lia_ctx* c, const lia_bf16* g, const lia_bf16* b) {
  if (!c || !g) return LIA_ERR_INVALID;
  c->chain_g = g;
  c->chain_b = b;
  c->chain_armed = true;
  return LIA_OK;
}

extern "C" int lia_ctx_create(int device, size_t workspace_bytes, lia_ctx** out) {
  if (!out) return LIA_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) {
    lia_set_error("lia_ctx_create: device %d not present (%d devices)", device, n);
    return LIA_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(device));
  lia_ctx* c = new lia_ctx();
  memset(c, 0, sizeof(*c));
  c->device = device;
  HIP_TRY(hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking));
  {
    const char* ser = getenv("LIA_SERIALIZE");
    c->serialized = ser && ser[0] == '1';
  }
  if (c->serialized) {
    c->d2h = c->compute;
  } else {
    // K/V delivery must not starve behind back-to-back GEMMs when it falls back to a blit kernel (strided case)
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_TRY(hipStreamCreateWithPriority(&c->d2h, hipStreamNonBlocking, hi));
  }
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(hipEventCreateWithFlags(&c->slab_ready[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->slab_done[i], hipEventDisableTiming));
  }
  c->ws_bytes = workspace_bytes;
  if (workspace_bytes) HIP_TRY(hipMalloc((void**)&c->ws, workspace_bytes));
  c->gemm_opts.fuse_combine = 1;
  *out = c;
  return LIA_OK;
}

extern "C" void lia_ctx_destroy(lia_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->compute);
  (void)hipStreamSynchronize(c->d2h);
  for (int i = 0; i < 2; ++i) {
    (void)hipEventDestroy(c->slab_ready[i]);
    (void)hipEventDestroy(c->slab_done[i]);
  }
  if (c->ws) (void)hipFree(c->ws);
  if (c->hidden_tmp) (void)hipFree(c->hidden_tmp);
  if (c->host_stage) (void)hipHostFree(c->host_stage);
  if (c->deliver_t0) { (void)hipEventDestroy(c->deliver_t0); (void)hipEventDestroy(c->deliver_t1); }
  if (c->deliver_events) {
    for (hipEvent_t e : *c->deliver_events) (void)hipEventDestroy(e);
    delete c->deliver_events;
    delete c->deliver_pending;
  }
  if (c->prof_events) {
    for (hipEvent_t e : *c->prof_events) (void)hipEventDestroy(e);
    delete c->prof_events;
    delete c->prof_recs;
  }
  if (c->d2h != c->compute) (void)hipStreamDestroy(c->d2h);
  (void)hipStreamDestroy(c->compute);
  delete c;
}

extern "C" void* lia_ctx_compute_stream(lia_ctx* c) { return c ? (void*)c->compute : nullptr; }
extern "C" int lia_ctx_serialized(lia_ctx* c) { return c && c->serialized ? 1 : 0; }

// hipStreamSynchronize spins on this image (one CPU of the container's CFS quota for as long as the GPU works); a blocking event
// instead (the thread sleeps, +20-50 us to wake) was measured in r03: less throttling, no gain per step (LABNOTES.md) -- spinning stays.
static int ctx_wait(hipStream_t st) {
  HIP_TRY(hipStreamSynchronize(st));
  return LIA_OK;
}

extern "C" int lia_ctx_synchronize(lia_ctx* c) {
  if (!c) return LIA_ERR_INVALID;
  if (int rc = ctx_wait(c->compute)) return rc;
  return ctx_wait(c->d2h);
}

extern "C" int lia_ctx_synchronize_compute(lia_ctx* c) {
  if (!c) return LIA_ERR_INVALID;
  return ctx_wait(c->compute);
}

// Per-context options and counters (r05: the A/B switches used to be process-wide setters -- lia_gemm_set_* -- although the
// library promises that two contexts of one process are independent, include/lia_hip.h).
extern "C" int lia_ctx_set_option(lia_ctx* c, int key, long value) {
  if (!c) return LIA_ERR_INVALID;
  switch (key) {
    case LIA_OPT_FUSE_COMBINE: c->gemm_opts.fuse_combine = value ? 1 : 0; return LIA_OK;
    default: lia_set_error("lia_ctx_set_option: unknown key %d", key); return LIA_ERR_INVALID;
  }
}
extern "C" long lia_ctx_get_counter(lia_ctx* c, int key) {
  if (!c) return -1;
  if (key > LIA_CNT_FUSED_COMBINE && key <= LIA_CNT_FUSED_COMBINE + 4) return c->gemm_opts.fused_combines[key - LIA_CNT_FUSED_COMBINE];
  return -1;
}

extern "C" int lia_ctx_set_host_threads(lia_ctx* c, int n) {
  if (!c || n < 0) return LIA_ERR_INVALID;
  c->host_threads = n;
  return LIA_OK;
}

extern "C" int lia_prof_start(lia_ctx* c, int max_launches) {
  if (!c || max_launches <= 0) return LIA_ERR_INVALID;
  if (!c->prof_events) { c->prof_events = new std::vector<hipEvent_t>(); c->prof_recs = new std::vector<lia_ctx::ProfRec>(); }
  while (c->prof_events->size() < (size_t)2 * max_launches) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    c->prof_events->push_back(e);
  }
  c->prof_recs->clear();
  c->prof_host_attn_calls = 0;
  c->prof_host_attn_ms = 0.0;
  c->prof_cap = max_launches;
  c->prof_seen = 0;
  if (c->prof_stride < 1) c->prof_stride = 1;
  c->prof_on = true;
  return LIA_OK;
}

extern "C" int lia_prof_set_stride(lia_ctx* c, int stride) {
  if (!c || stride < 1) return LIA_ERR_INVALID;
  c->prof_stride = stride;
  return LIA_OK;
}

extern "C" int lia_prof_stop(lia_ctx* c, lia_prof_result* out) {
  if (!c || !out || !c->prof_events) return LIA_ERR_INVALID;
  c->prof_on = false;
  memset(out, 0, sizeof(*out));
  out->host_attention_calls = c->prof_host_attn_calls;
  out->host_attention_ms = c->prof_host_attn_ms;
  HIP_TRY(hipStreamSynchronize(c->compute));
  for (size_t i = 0; i < c->prof_recs->size(); ++i) {
    float ms = 0.f;
    HIP_TRY(hipEventSynchronize((*c->prof_events)[2 * i + 1]));
    HIP_TRY(hipEventElapsedTime(&ms, (*c->prof_events)[2 * i], (*c->prof_events)[2 * i + 1]));
    const auto& r = (*c->prof_recs)[i];
    if (r.regime == 1) { out->skinny_launches++; out->skinny_ms += ms; out->skinny_bytes += r.bytes; out->skinny_flops += r.flops; }
    else { out->tiled_launches++; out->tiled_ms += ms; out->tiled_bytes += r.bytes; out->tiled_flops += r.flops; }
  }
  // calibrate the bracket: 32 empty event pairs on the same stream (rocprofv3's kernel durations carry no such term)
  if (c->prof_events->size() >= 2) {
    double sum = 0.0;
    const int reps = 32;
    for (int i = 0; i < reps; ++i) {
      HIP_TRY(hipEventRecord((*c->prof_events)[0], c->compute));
      HIP_TRY(hipEventRecord((*c->prof_events)[1], c->compute));
      HIP_TRY(hipEventSynchronize((*c->prof_events)[1]));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, (*c->prof_events)[0], (*c->prof_events)[1]));
      sum += ms;
    }
    out->empty_bracket_ms = sum / reps;
  }
  return LIA_OK;
}

extern "C" int lia_ctx_kv_store_wait(lia_ctx* c) {
  if (!c) return LIA_ERR_INVALID;
  HIP_TRY(hipStreamSynchronize(c->d2h));
  return LIA_OK;
}

extern "C" int lia_kv_deliver(lia_ctx* c, const lia_kv* dev, lia_kv* host, int T, int row_elems, int* ticket) {
  if (!c || !dev || !host || !ticket) return LIA_ERR_INVALID;
  if (!dev->k || !dev->v || !host->k || !host->v) { lia_set_error("lia_kv_deliver: NULL cache"); return LIA_ERR_MISSING; }
  if (!dev->on_device || host->on_device || dev->batch != host->batch || T <= 0 || T > dev->smax || T > host->smax || row_elems <= 0) {
    lia_set_error("lia_kv_deliver: dev(on_device=%d batch=%d smax=%d) host(on_device=%d batch=%d smax=%d) T=%d", dev->on_device, dev->batch,
                  dev->smax, host->on_device, host->batch, host->smax, T);
    return LIA_ERR_INVALID;
  }
  if (!c->deliver_events) { c->deliver_events = new std::vector<hipEvent_t>(); c->deliver_pending = new std::vector<char>(); }
  // A ticket is freed by lia_kv_deliver_wait and by nothing else: recycling one whose copy merely happened to be complete
  // (r02) handed the same id to a later layer of the SAME prefill, and waiting for layer i then meant waiting for layer j > i.
  int id = -1;
  for (size_t i = 0; i < c->deliver_pending->size(); ++i)
    if (!(*c->deliver_pending)[i]) { id = (int)i; break; }
  if (id < 0) {
    hipEvent_t e;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->deliver_events->push_back(e);
    c->deliver_pending->push_back(0);
    id = (int)c->deliver_events->size() - 1;
  }
  // behind the compute stream's work so far (the layer that filled the holding cache), on the K/V delivery stream
  HIP_TRY(hipEventRecord((*c->deliver_events)[id], c->compute));
  HIP_TRY(hipStreamWaitEvent(c->d2h, (*c->deliver_events)[id], 0));
  if (!c->deliver_t0) { HIP_TRY(hipEventCreate(&c->deliver_t0)); HIP_TRY(hipEventCreate(&c->deliver_t1)); }
  if (!c->deliver_batch_open) { HIP_TRY(hipEventRecord(c->deliver_t0, c->d2h)); c->deliver_batch_open = true; }
  // the strided 2-D form even though the rows are contiguous: the runtime serves it with a blit kernel, a linear copy would
  // queue on the SDMA engine behind the next step's weight copies (see the policy-0 delivery in lia_layer_forward)
  const size_t width = (size_t)dev->batch * row_elems * 2;
  HIP_TRY(hipMemcpy2DAsync(host->k, width, dev->k, width, width, T, hipMemcpyDeviceToHost, c->d2h));
  HIP_TRY(hipMemcpy2DAsync(host->v, width, dev->v, width, width, T, hipMemcpyDeviceToHost, c->d2h));
  HIP_TRY(hipEventRecord((*c->deliver_events)[id], c->d2h));
  HIP_TRY(hipEventRecord(c->deliver_t1, c->d2h));
  (*c->deliver_pending)[id] = 1;
  *ticket = id;
  return LIA_OK;
}

// Device time of the deliveries issued since the last call of this function (first copy's start to last copy's end on the K/V
// delivery stream); blocks until the last one has landed.  For the benchmark's prefill accounting.
extern "C" int lia_kv_deliver_batch_ms(lia_ctx* c, double* ms) {
  if (!c || !ms) return LIA_ERR_INVALID;
  *ms = 0.0;
  if (!c->deliver_batch_open) return LIA_OK;
  HIP_TRY(hipEventSynchronize(c->deliver_t1));
  float f = 0.f;
  HIP_TRY(hipEventElapsedTime(&f, c->deliver_t0, c->deliver_t1));
  *ms = f;
  c->deliver_batch_open = false;
  return LIA_OK;
}

extern "C" int lia_kv_deliver_wait(lia_ctx* c, int ticket) {
  if (!c || !c->deliver_events || ticket < 0 || ticket >= (int)c->deliver_events->size()) { lia_set_error("lia_kv_deliver_wait: ticket %d", ticket); return LIA_ERR_INVALID; }
  if (!(*c->deliver_pending)[ticket]) return LIA_OK;
  HIP_TRY(hipEventSynchronize((*c->deliver_events)[ticket]));
  (*c->deliver_pending)[ticket] = 0;
  return LIA_OK;
}

// ------------------------------------------------------------------------------------------------
// packed layer layout / workspace sizing
// ------------------------------------------------------------------------------------------------
static int check_desc(const lia_layer_desc* d) {
  if (!d) return LIA_ERR_INVALID;
  if (d->hidden <= 0 || d->heads <= 0 || d->ffn <= 0 || d->hidden % d->heads) {
    lia_set_error("layer desc: hidden=%d heads=%d ffn=%d", d->hidden, d->heads, d->ffn);
    return LIA_ERR_INVALID;
  }
  int hd = d->hidden / d->heads;
  if (!(hd == 32 || hd == 64 || hd == 128) || d->hidden % 64 || d->ffn % 64) {
    lia_set_error("layer desc: head_dim %d must be 32/64/128 and hidden, ffn multiples of 64", hd);
    return LIA_ERR_INVALID;
  }
  return LIA_OK;
}

extern "C" int lia_layer_pack_offsets(const lia_layer_desc* d, size_t off[16], size_t* total) {
  int rc = check_desc(d);
  if (rc) return rc;
  const size_t H = d->hidden, F = d->ffn;
  size_t p = 0;
  auto put = [&](int idx, size_t elems, bool align) {
    if (align) p = align_up(p, 256);
    off[idx] = p;
    p += elems * 2;
  };
  // q.w | k.w | v.w contiguous = one [3H,H] operand; q.b | k.b | v.b contiguous = one [3H] bias
  put(2, H * H, true);  put(4, H * H, false);  put(6, H * H, false);
  put(3, H, true);      put(5, H, false);      put(7, H, false);
  put(8, H * H, true);  put(9, H, true);
  put(12, F * H, true); put(13, F, true);
  put(14, H * F, true); put(15, H, true);
  put(0, H, true);      put(1, H, true);
  put(10, H, true);     put(11, H, true);
  if (total) *total = align_up(p, 2048);   // 1024 bf16 values: the block size of the pack10 wire format
  return LIA_OK;
}

struct WsLayout {
  size_t ln, q, k, v, attn, h1, f1, slab[2], gemm, total;
  size_t gemm_bytes;
};

static WsLayout ws_layout(const lia_layer_desc* d, long rows, long slab_rows) {
  WsLayout w;
  const size_t H = d->hidden, F = d->ffn, R = (size_t)rows;
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t o = p; p = align_up(p + bytes, 256); return o; };
  w.ln = take(R * H * 2);
  w.q = take(R * H * 2);   // q | k | v contiguous: the policy-2 round trip moves them in one copy
  w.k = take(R * H * 2);
  w.v = take(R * H * 2);
  w.attn = take(R * H * 2);
  w.h1 = take(R * H * 2);
  w.f1 = take(R * F * 2);
  w.slab[0] = take(2 * (size_t)slab_rows * H * 2);
  w.slab[1] = take(2 * (size_t)slab_rows * H * 2);
  // split-K slabs of the skinny regime (M <= 256).  Reserved for larger calls too, so that the size is monotonic in `rows`: a
  // context sized for a 292-row call must also serve a 256-row one (r01 sized 16 MB for 292 rows and then needed 82 MB for 256)
  w.gemm_bytes = (size_t)8 * std::min(R, (size_t)256) * std::max(3 * H, F) * 4;
  w.gemm = take(w.gemm_bytes);
  w.total = p;
  return w;
}

extern "C" size_t lia_layer_workspace_bytes(const lia_layer_desc* d, int max_rows) {
  if (check_desc(d) || max_rows <= 0) return 0;
  return ws_layout(d, max_rows, max_rows).total;
}

// ------------------------------------------------------------------------------------------------
// sub-layer ops
// ------------------------------------------------------------------------------------------------
static LiaOutMap plain_out(bf16_t* y, long ldy, int N) {
  LiaOutMap om;
  memset(&om, 0, sizeof(om));
  om.base[0] = y;
  om.ld[0] = ldy;
  om.seg_n = N;
  om.T = 1;
  return om;
}

extern "C" int lia_layernorm(const lia_bf16* x, long ldx, const lia_bf16* g, const lia_bf16* b, lia_bf16* y, long ldy,
                             long rows, int H, float eps, void* stream) {
  if (!x || !g || !b || !y) { lia_set_error("lia_layernorm: NULL tensor"); return LIA_ERR_MISSING; }
  if (rows < 0 || H <= 0 || H % 8) { lia_set_error("lia_layernorm: rows=%ld H=%d (H %% 8)", rows, H); return LIA_ERR_INVALID; }
  lia_layernorm_launch(x, ldx, g, b, y, ldy, rows, H, eps, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return LIA_OK;
}

static int gemm_checked(lia_ctx* ctx, const bf16_t* x, long ldx, const bf16_t* w, int M, int N, int K,
                        const LiaEpilogue& ep, const LiaOutMap& om, float* ws, size_t ws_bytes, int split, hipStream_t st,
                        const LiaPost* post = nullptr, int* post_done = nullptr) {
  if (N % 16 || K % 64) {
    lia_set_error("linear: N=%d must be a multiple of 16 and K=%d of 64", N, K);
    return LIA_ERR_INVALID;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int regime = 0;
  const bool timed = ctx && ctx->prof_on && ctx->prof_recs->size() < ctx->prof_cap && (ctx->prof_seen++ % ctx->prof_stride) == 0;
  if (timed) {
    size_t i = ctx->prof_recs->size();
    e0 = (*ctx->prof_events)[2 * i];
    e1 = (*ctx->prof_events)[2 * i + 1];
  }
  int rc = lia_gemm_launch(x, ldx, w, (long)K, M, N, K, &ep, &om, ws, ws_bytes, ctx ? &ctx->gemm_opts : nullptr, split, st, e0, e1, &regime, post, post_done);
  if (timed && rc == 0 && regime != 0) {
    // algorithmic traffic of the op: the weight once, the activations in and out once
    double bytes = 2.0 * ((double)N * K + (double)M * K + (double)M * N);
    ctx->prof_recs->push_back({regime, bytes, 2.0 * M * (double)N * K});
  }
  if (rc) { lia_set_error("linear: unsupported shape M=%d N=%d K=%d", M, N, K); return LIA_ERR_INVALID; }
  HIP_TRY(hipGetLastError());
  return LIA_OK;
}

extern "C" int lia_linear(lia_ctx* ctx, const lia_bf16* x, long ldx, const lia_bf16* w, const lia_bf16* bias,
                          const lia_bf16* residual, long ldr, lia_bf16* y, long ldy, int M, int N, int K, int relu,
                          int split_k, void* stream) {
  if (!ctx) return LIA_ERR_INVALID;
  if (!x || !w || !y) { lia_set_error("lia_linear: NULL tensor"); return LIA_ERR_MISSING; }
  if (M < 0 || N <= 0 || K <= 0) { lia_set_error("lia_linear: M=%d N=%d K=%d", M, N, K); return LIA_ERR_INVALID; }
  LiaEpilogue ep{bias, residual, ldr, relu};
  LiaOutMap om = plain_out(y, ldy, N);
  size_t need = lia_gemm_workspace_bytes(M, N);          // split-K slabs: 8 for M <= 256, up to 4 for the tiled kernel below M = 2048
  size_t have = std::min(need, ctx->ws_bytes);
  ctx->normed_src = nullptr; ctx->chain_armed = false;   // the workspace is about to be reused
  return gemm_checked(ctx, x, ldx, w, M, N, K, ep, om, (float*)ctx->ws, have, split_k, (hipStream_t)stream);
}

static int qkv_project(lia_ctx* ctx, const bf16_t* x, const bf16_t* w, const bf16_t* bias, bf16_t* qout, bf16_t* kdst,
                       bf16_t* vdst, bool kv_cache_mode, int B, int T, int H, int cache_batch, int b0, int pos0,
                       float* ws, size_t ws_bytes, hipStream_t st) {
  LiaEpilogue ep{bias, nullptr, 0, 0};
  LiaOutMap om;
  memset(&om, 0, sizeof(om));
  om.base[0] = qout; om.base[1] = kdst; om.base[2] = vdst;
  om.ld[0] = om.ld[1] = om.ld[2] = H;
  om.cache_mode[1] = om.cache_mode[2] = kv_cache_mode ? 1 : 0;
  om.seg_n = H; om.T = T; om.Bc = cache_batch; om.b0 = b0; om.pos0 = pos0;
  return gemm_checked(ctx, x, H, w, B * T, 3 * H, H, ep, om, ws, ws_bytes, 0, st);
}

extern "C" int lia_qkv_project(lia_ctx* ctx, const lia_bf16* x, const lia_bf16* w, const lia_bf16* bias, lia_bf16* qout,
                               lia_bf16* kcache, lia_bf16* vcache, int B, int T, int H, int cache_batch, int b0, int pos0,
                               void* stream) {
  if (!ctx) return LIA_ERR_INVALID;
  if (!x || !w || !qout || !kcache || !vcache) { lia_set_error("lia_qkv_project: NULL tensor"); return LIA_ERR_MISSING; }
  if (B <= 0 || T <= 0 || b0 < 0 || b0 + B > cache_batch || pos0 < 0) {
    lia_set_error("lia_qkv_project: B=%d T=%d cache_batch=%d b0=%d pos0=%d", B, T, cache_batch, b0, pos0);
    return LIA_ERR_INVALID;
  }
  size_t need = (size_t)B * T <= 256 ? (size_t)8 * B * T * 3 * H * 4 : 0;
  ctx->normed_src = nullptr; ctx->chain_armed = false;   // the workspace is about to be reused
  return qkv_project(ctx, x, w, bias, qout, kcache, vcache, true, B, T, H, cache_batch, b0, pos0, (float*)ctx->ws,
                     std::min(need, ctx->ws_bytes), (hipStream_t)stream);
}

extern "C" int lia_attention(const lia_bf16* q, long ldq, const lia_bf16* kcache, const lia_bf16* vcache, lia_bf16* out,
                             long ldo, int B, int T, int S, int heads, int head_dim, int cache_batch, int b0, void* stream) {
  if (!q || !kcache || !vcache || !out) { lia_set_error("lia_attention: NULL tensor"); return LIA_ERR_MISSING; }
  if (B <= 0 || T <= 0 || S < T || b0 < 0 || b0 + B > cache_batch) {
    lia_set_error("lia_attention: B=%d T=%d S=%d cache_batch=%d b0=%d", B, T, S, cache_batch, b0);
    return LIA_ERR_INVALID;
  }
  int rc;
  if (T == 1) {
    rc = lia_attn_decode_launch(q, ldq, kcache, vcache, out, ldo, B, S, heads, heads, head_dim, cache_batch, b0, 0, (hipStream_t)stream);
  } else {
    if (S != T) { lia_set_error("lia_attention: multi-token blocks only as a prefill (S == T), got S=%d T=%d", S, T); return LIA_ERR_INVALID; }
    rc = lia_attn_prefill_launch(q, ldq, kcache, vcache, out, ldo, B, T, heads, heads, head_dim, cache_batch, b0, 0, (hipStream_t)stream);
  }
  if (rc) {
    lia_set_error("lia_attention: head_dim %d unsupported (32/64/128), S=%d too long, or a cache of %d batch rows x %d values per key row beyond the d = 128 prefill kernel's 32-bit staging offsets (64 key rows < 4 GiB)",
                  head_dim, S, cache_batch, heads * head_dim);
    return LIA_ERR_INVALID;
  }
  HIP_TRY(hipGetLastError());
  return LIA_OK;
}

extern "C" int lia_embed(const int64_t* ids, const lia_bf16* tok, const lia_bf16* pos, lia_bf16* y, int B, int T,
                         int past_len, int H, void* stream) {
  if (!ids || !tok || !pos || !y) { lia_set_error("lia_embed: NULL tensor"); return LIA_ERR_MISSING; }
  if (B <= 0 || T <= 0 || past_len < 0 || H % 8) { lia_set_error("lia_embed: B=%d T=%d past=%d H=%d", B, T, past_len, H); return LIA_ERR_INVALID; }
  lia_embed_launch(ids, tok, pos, y, B, T, past_len, H, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return LIA_OK;
}

extern "C" int lia_lm_head(lia_ctx* ctx, const lia_bf16* hidden, int B, int T, int H, const lia_bf16* lnw,
                           const lia_bf16* lnb, const lia_bf16* emb, int vocab, float eps, int suppress_token,
                           lia_bf16* logits, int64_t* next_ids, void* stream) {
  if (!ctx) return LIA_ERR_INVALID;
  if (!hidden || !lnw || !lnb || !emb || !logits || !next_ids) { lia_set_error("lia_lm_head: NULL tensor"); return LIA_ERR_MISSING; }
  if (B <= 0 || B > 256 || T <= 0 || vocab % 16) { lia_set_error("lia_lm_head: B=%d (<=256) T=%d vocab=%d (%%16)", B, T, vocab); return LIA_ERR_INVALID; }
  size_t scratch = align_up((size_t)B * H * 2, 256);
  size_t gemm_need = (size_t)8 * B * vocab * 4;
  if (ctx->ws_bytes < scratch) { lia_set_error("lia_lm_head: workspace too small"); return LIA_ERR_MEMORY; }
  hipStream_t st = (hipStream_t)stream;
  ctx->normed_src = nullptr; ctx->chain_armed = false;   // the workspace is about to be reused
  bf16_t* lno = (bf16_t*)ctx->ws;
  // hidden[:, -1, :] -> final LN (lia/modeling_opt.py:1563 applies it to all positions; only the last one feeds
  // lm_head, models.py:424-431, and LN is row-wise, so the other rows are never needed)
  lia_layernorm_launch(hidden + (long)(T - 1) * H, (long)T * H, lnw, lnb, lno, H, B, H, eps, st);
  LiaEpilogue ep{nullptr, nullptr, 0, 0};
  LiaOutMap om = plain_out(logits, vocab, vocab);
  size_t have = ctx->ws_bytes - scratch;
  int rc = gemm_checked(ctx, lno, H, emb, B, vocab, H, ep, om, (float*)(ctx->ws + scratch), std::min(have, gemm_need), 0, st);
  if (rc) return rc;
  lia_argmax_launch(logits, next_ids, B, vocab, suppress_token, st);
  HIP_TRY(hipGetLastError());
  return LIA_OK;
}

// ------------------------------------------------------------------------------------------------
// the decoder-layer operator
// ------------------------------------------------------------------------------------------------
static int ensure_host_stage(lia_ctx* c, size_t bytes) {
  if (c->host_stage_bytes >= bytes) return LIA_OK;
  if (c->host_stage) (void)hipHostFree(c->host_stage);
  c->host_stage = nullptr;
  c->host_stage_bytes = 0;
  HIP_TRY(hipHostMalloc((void**)&c->host_stage, bytes, hipHostMallocDefault));
  c->host_stage_bytes = bytes;
  return LIA_OK;
}

// tail = 1 (lia_layer_forward_last): LN1 and the q | k | v projection run on all B x T rows (the cache needs every K/V row), everything
// behind them -- attention, out-proj, LN2, fc1, fc2 -- only on the LAST position of each row; y is then [B, 1, H].
static int layer_forward_impl(lia_ctx* ctx, const lia_layer_desc* d, int policy, const void* const weights[16], const lia_bf16* x,
                              lia_bf16* y, lia_kv* kv, int B, int T, int pos0, int b0, void* stream, int tail) {
  if (!ctx) return LIA_ERR_INVALID;
  int rc = check_desc(d);
  if (rc) return rc;
  if (!weights || !x || !y || !kv || !kv->k || !kv->v) { lia_set_error("lia_layer_forward: NULL tensor"); return LIA_ERR_MISSING; }
  for (int i = 0; i < 16; ++i)
    if (!weights[i]) { lia_set_error("lia_layer_forward: weights[%d] is NULL", i); return LIA_ERR_MISSING; }
  if (policy != 0 && policy != 2 && policy != 3) {
    lia_set_error("lia_layer_forward: policy %d is not a GPU policy (0, 2, 3)", policy);
    return LIA_ERR_INVALID;
  }
  if (B <= 0 || T <= 0 || pos0 < 0 || b0 < 0 || b0 + B > kv->batch || pos0 + T > kv->smax) {
    lia_set_error("lia_layer_forward: B=%d T=%d pos0=%d b0=%d vs cache batch=%d smax=%d", B, T, pos0, b0, kv->batch, kv->smax);
    return LIA_ERR_INVALID;
  }
  if (T > 1 && pos0 != 0) { lia_set_error("lia_layer_forward: multi-token call must be a prefill (pos0 == 0)"); return LIA_ERR_INVALID; }
  if (tail && (policy == 2 || T < 2)) { lia_set_error("lia_layer_forward_last: a GPU-attention prefill (policy 0 / 3, T > 1) only"); return LIA_ERR_INVALID; }
  if ((policy == 3) != (kv->on_device != 0)) {
    lia_set_error("lia_layer_forward: policy %d needs a %s cache", policy, policy == 3 ? "device" : "host");
    return LIA_ERR_INVALID;
  }
  const int H = d->hidden, F = d->ffn, hd = H / d->heads;
  const long M = (long)B * T;
  // policy-0 decode parks the whole cached prefix [pos0+1][B][H] (K and V) in a slab
  const WsLayout w = ws_layout(d, M, policy == 0 ? (long)(pos0 + T) * B : M);
  if (w.total > ctx->ws_bytes) {
    lia_set_error("lia_layer_forward: workspace %zu < %zu needed for %ld rows", ctx->ws_bytes, w.total, M);
    return LIA_ERR_MEMORY;
  }
  hipStream_t st = (hipStream_t)stream;
  {
    // a K/V delivery still draining from a slab of a DIFFERENT layout could overlap this call's buffers
    const long sr = policy == 0 ? (long)(pos0 + T) * B : M;
    if (ctx->last_rows != M || ctx->last_slab_rows != sr) {
      for (int i = 0; i < 2; ++i)
        if (ctx->slab_used[i]) HIP_TRY(hipStreamWaitEvent(st, ctx->slab_done[i], 0));
      ctx->last_rows = M;
      ctx->last_slab_rows = sr;
    }
  }
  char* ws = ctx->ws;
  bf16_t *ln = (bf16_t*)(ws + w.ln), *qb = (bf16_t*)(ws + w.q), *kb = (bf16_t*)(ws + w.k), *vb = (bf16_t*)(ws + w.v);
  bf16_t *ao = (bf16_t*)(ws + w.attn), *h1 = (bf16_t*)(ws + w.h1), *f1 = (bf16_t*)(ws + w.f1);
  float* gws = (float*)(ws + w.gemm);
  const bf16_t* const* W = (const bf16_t* const*)weights;
  const bool fused_qkv = (W[4] == W[2] + (size_t)H * H) && (W[6] == W[4] + (size_t)H * H) && (W[5] == W[3] + H) &&
                         (W[7] == W[5] + H);
  const float eps = d->ln_eps;
  const size_t S = (size_t)pos0 + T;
  // rows behind the projection: all of them, or the last position of each batch row (row b*T + T-1: stride T*H)
  const long Mt = tail ? (long)B : M;
  const long tail_off = tail ? (long)(T - 1) * H : 0, tail_ld = tail ? (long)T * H : (long)H;
  // chaining state: a hint for THIS call's last GEMM, and whether the previous call already left LN1(x) in `ln`
  const bf16_t *chain_g = ctx->chain_armed ? ctx->chain_g : nullptr, *chain_b = ctx->chain_armed ? ctx->chain_b : nullptr;
  ctx->chain_armed = false;
  const bool have_ln1 = ctx->normed_src == (const void*)x && ctx->normed_buf == (const void*)ln && ctx->normed_rows == M && ctx->normed_h == H;
  ctx->normed_src = nullptr;

  // LN1 (decoder.py:199-206)
  if (!have_ln1) lia_layernorm_launch(x, H, W[0], W[1], ln, H, M, H, eps, st);

  // destinations of the fresh K/V rows
  bf16_t *kdst, *vdst;
  int dst_batch, dst_b0, dst_pos0;
  bool cache_mode = true;
  int slab = -1;
  if (policy == 3) {
    kdst = kv->k; vdst = kv->v; dst_batch = kv->batch; dst_b0 = b0; dst_pos0 = pos0;
  } else if (policy == 0) {
    // device slab [S][B][h][d] shared by K then V; two slabs alternate so the previous delivery may still drain
    slab = ctx->slab_next; ctx->slab_next ^= 1;
    if (ctx->slab_used[slab]) HIP_TRY(hipStreamWaitEvent(st, ctx->slab_done[slab], 0));
    kdst = (bf16_t*)(ws + w.slab[slab]); vdst = kdst + S * B * H;
    dst_batch = B; dst_b0 = 0; dst_pos0 = pos0;
    if (pos0 > 0) {
      // intended semantics of the reference's decode policy 0 (modeling_opt.py:1379-1491; its own
      // implementation is not an oracle): bring the cached rows to the GPU, attend there
      const size_t width = (size_t)B * H * 2, pitch = (size_t)kv->batch * H * 2;
      HIP_TRY(hipMemcpy2DAsync(kdst, width, kv->k + (size_t)b0 * H, pitch, width, pos0, hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpy2DAsync(vdst, width, kv->v + (size_t)b0 * H, pitch, width, pos0, hipMemcpyHostToDevice, st));
    }
  } else {  // policy 2: plain [M,H] buffers that travel to the host
    kdst = kb; vdst = vb; dst_batch = B; dst_b0 = 0; dst_pos0 = 0; cache_mode = false;
  }

  // q | k | v projection (attentions.py:393-394, 418)
  if (fused_qkv) {
    rc = qkv_project(ctx, ln, W[2], W[3], qb, kdst, vdst, cache_mode, B, T, H, dst_batch, dst_b0, dst_pos0, gws, w.gemm_bytes, st);
    if (rc) return rc;
  } else {
    for (int i = 0; i < 3; ++i) {
      LiaEpilogue ep{W[3 + 2 * i], nullptr, 0, 0};
      LiaOutMap om = plain_out(i == 0 ? qb : (i == 1 ? kdst : vdst), H, H);
      if (i > 0 && cache_mode) { om.cache_mode[0] = 1; om.T = T; om.Bc = dst_batch; om.b0 = dst_b0; om.pos0 = dst_pos0; }
      rc = gemm_checked(ctx, ln, H, W[2 + 2 * i], (int)M, H, H, ep, om, gws, w.gemm_bytes, 0, st);
      if (rc) return rc;
    }
  }

  if (policy == 2) {
    // q, k, v -> host; host attention over the host cache; result back (attentions.py:421-440)
    const size_t one = (size_t)M * H * 2;
    rc = ensure_host_stage(ctx, 4 * one);
    if (rc) return rc;
    bf16_t* hq = (bf16_t*)ctx->host_stage;
    bf16_t *hk = hq + (size_t)M * H, *hv = hk + (size_t)M * H, *ha = hv + (size_t)M * H;
    // kernel blits over the mapped pinned buffer, not SDMA copies (see lia_blit_kernel)
    lia_blit_launch(hq, qb, one, st);
    lia_blit_launch(hk, kb, one, st);
    lia_blit_launch(hv, vb, one, st);
    HIP_TRY(hipStreamSynchronize(st));
    const auto ha_t0 = std::chrono::steady_clock::now();
    rc = lia_host_attention(hq, hk, hv, kv->k, kv->v, ha, B, T, pos0, d->heads, hd, kv->batch, b0, ctx->host_threads);
    if (rc) return rc;
    if (ctx->prof_on) {
      ctx->prof_host_attn_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ha_t0).count();
      ctx->prof_host_attn_calls++;
    }
    lia_blit_launch(ao, ha, one, st);
  } else {
    // GPU attention (attentions.py:443-536)
    // (tail: the last query of a causal block attends every key -- one decode-style query per row over the S rows just written)
    int arc = (T == 1 || tail) ? lia_attn_decode_launch(qb + tail_off, tail_ld, kdst, vdst, ao, H, B, (int)S, d->heads, d->heads, hd, dst_batch, dst_b0, 0, st)
                               : lia_attn_prefill_launch(qb, H, kdst, vdst, ao, H, B, T, d->heads, d->heads, hd, dst_batch, dst_b0, 0, st);
    if (arc) { lia_set_error("attention: unsupported head_dim %d / S %zu", hd, S); return LIA_ERR_INVALID; }
    if (policy == 0) {
      // deliver rows [pos0, pos0+T) of batch rows [b0, b0+B) to the host cache
      HIP_TRY(hipEventRecord(ctx->slab_ready[slab], st));
      HIP_TRY(hipStreamWaitEvent(ctx->d2h, ctx->slab_ready[slab], 0));
      const size_t width = (size_t)B * H * 2, pitch = (size_t)kv->batch * H * 2;
      lia_bf16* hk = kv->k + ((size_t)pos0 * kv->batch + b0) * H;
      lia_bf16* hv = kv->v + ((size_t)pos0 * kv->batch + b0) * H;
      // Always the strided 2-D path, even when width == pitch: the runtime serves it with a blit kernel, which
      // shares PCIe with the weight stream gracefully.  A linear copy would go to the SDMA queue behind the
      // 1.2 GB weight copies (measured: prefill 1057 ms -> 1335 ms with four slots).
      HIP_TRY(hipMemcpy2DAsync(hk, pitch, kdst + (size_t)pos0 * B * H, width, width, T, hipMemcpyDeviceToHost, ctx->d2h));
      HIP_TRY(hipMemcpy2DAsync(hv, pitch, vdst + (size_t)pos0 * B * H, width, width, T, hipMemcpyDeviceToHost, ctx->d2h));
      HIP_TRY(hipEventRecord(ctx->slab_done[slab], ctx->d2h));
      ctx->slab_used[slab] = true;
    }
  }

  // out-proj + bias, residual (decoder.py:225-229); LN2 (:268-276) rides in the split-K combine when there is one
  int ln2_done = 0;
  {
    LiaEpilogue ep{W[9], x + tail_off, tail_ld, 0};
    LiaOutMap om = plain_out(h1, H, H);
    LiaPost post{};
    post.kind = LIA_POST_LAYERNORM; post.g = W[10]; post.b = W[11]; post.eps = eps; post.out = ln; post.ldo = H;
    rc = gemm_checked(ctx, ao, H, W[8], (int)Mt, H, H, ep, om, gws, w.gemm_bytes, 0, st, &post, &ln2_done);
    if (rc) return rc;
  }
  // fc1 + relu (:282-285), fc2 + residual (:306-310)
  if (!ln2_done) lia_layernorm_launch(h1, H, W[10], W[11], ln, H, Mt, H, eps, st);
  {
    LiaEpilogue ep{W[13], nullptr, 0, 1};
    LiaOutMap om = plain_out(f1, F, F);
    rc = gemm_checked(ctx, ln, H, W[12], (int)Mt, F, H, ep, om, gws, w.gemm_bytes, 0, st);
    if (rc) return rc;
  }
  {
    LiaEpilogue ep{W[15], h1, H, 0};
    LiaOutMap om = plain_out(y, H, H);
    LiaPost post{};
    int chained = 0;
    if (chain_g && chain_b) { post.kind = LIA_POST_LAYERNORM; post.g = chain_g; post.b = chain_b; post.eps = eps; post.out = ln; post.ldo = H; }
    rc = gemm_checked(ctx, f1, F, W[14], (int)Mt, H, F, ep, om, gws, w.gemm_bytes, 0, st, post.kind ? &post : nullptr, &chained);
    if (rc) return rc;
    if (chained) { ctx->normed_src = y; ctx->normed_buf = ln; ctx->normed_rows = Mt; ctx->normed_h = H; }
  }
  HIP_TRY(hipGetLastError());
  return LIA_OK;
}

extern "C" int lia_layer_forward(lia_ctx* ctx, const lia_layer_desc* d, int policy, const void* const weights[16],
                                 const lia_bf16* x, lia_bf16* y, lia_kv* kv, int B, int T, int pos0, int b0, void* stream) {
  return layer_forward_impl(ctx, d, policy, weights, x, y, kv, B, T, pos0, b0, stream, 0);
}

extern "C" int lia_layer_forward_last(lia_ctx* ctx, const lia_layer_desc* d, int policy, const void* const weights[16],
                                      const lia_bf16* x, lia_bf16* y_last, lia_kv* kv, int B, int T, int pos0, int b0, void* stream) {
  return layer_forward_impl(ctx, d, policy, weights, x, y_last, kv, B, T, pos0, b0, stream, 1);
}

// ------------------------------------------------------------------------------------------------
// Llama-family layer (config 4, build-defined): RMSNorm, RoPE, grouped-query attention, SiLU-gated MLP
// ------------------------------------------------------------------------------------------------
extern "C" void lia_rmsnorm_launch(const bf16_t* x, long ldx, const bf16_t* w, bf16_t* y, long ldy, long rows, int H, float eps,
                                   hipStream_t st);
extern "C" void lia_rope_launch(bf16_t* x, long row_stride, const bf16_t* cosb, const bf16_t* sinb, long rows, int heads, int d,
                                int pos0, int pos_mod, int pos_div, hipStream_t st);
extern "C" void lia_silu_mul_launch(const bf16_t* gu, bf16_t* out, long M, int F, int gu_block, hipStream_t st);
extern "C" void lia_embed_tokens_launch(const int64_t* ids, const bf16_t* tok, bf16_t* y, long rows, int H, hipStream_t st);

static int check_llama_desc(const lia_llama_desc* d) {
  if (!d) return LIA_ERR_INVALID;
  if (d->hidden <= 0 || d->heads <= 0 || d->kv_heads <= 0 || d->ffn <= 0 || d->hidden % d->heads || d->heads % d->kv_heads) {
    lia_set_error("llama desc: hidden=%d heads=%d kv_heads=%d ffn=%d", d->hidden, d->heads, d->kv_heads, d->ffn);
    return LIA_ERR_INVALID;
  }
  int hd = d->hidden / d->heads;
  if (!(hd == 32 || hd == 64 || hd == 128) || d->hidden % 128 || d->ffn % 128 || (d->kv_heads * hd) % 16) {
    lia_set_error("llama desc: head_dim %d must be 32/64/128, hidden and ffn multiples of 128", hd);
    return LIA_ERR_INVALID;
  }
  if (d->gu_block != 0 && d->gu_block != LIA_GU_BLOCK) {
    lia_set_error("llama desc: gu_block %d (0 = gate.w | up.w as they are, %d = interleaved in blocks of %d rows)", d->gu_block, LIA_GU_BLOCK, LIA_GU_BLOCK);
    return LIA_ERR_INVALID;
  }
  return LIA_OK;
}

extern "C" int lia_llama_pack_offsets(const lia_llama_desc* d, size_t off[9], size_t* total) {
  int rc = check_llama_desc(d);
  if (rc) return rc;
  const size_t H = d->hidden, F = d->ffn, KD = (size_t)d->kv_heads * (H / d->heads);
  size_t p = 0;
  auto put = [&](int idx, size_t elems, bool align) { if (align) p = align_up(p, 256); off[idx] = p; p += elems * 2; };
  put(1, H * H, true);                            // q
  put(2, KD * H, true); put(3, KD * H, false);    // k | v adjacent: one [2*KD, H] GEMM
  put(4, H * H, true);                            // o
  put(6, F * H, true); put(7, F * H, false);      // gate | up adjacent: one [2F, H] GEMM
  put(8, H * F, true);                            // down
  put(0, H, true); put(5, H, true);               // the two RMSNorm weights
  if (total) *total = align_up(p, 2048);
  return LIA_OK;
}

struct LlamaWs { size_t ln, q, attn, h1, gu, act, gemm, gemm_bytes, total; };
static LlamaWs llama_ws(const lia_llama_desc* d, long rows) {
  LlamaWs w;
  const size_t H = d->hidden, F = d->ffn, R = (size_t)rows;
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t o = p; p = align_up(p + bytes, 256); return o; };
  w.ln = take(R * H * 2); w.q = take(R * H * 2); w.attn = take(R * H * 2); w.h1 = take(R * H * 2);
  w.gu = take(R * 2 * F * 2); w.act = take(R * F * 2);
  w.gemm_bytes = (size_t)8 * std::min(R, (size_t)256) * 2 * F * 4;   // monotonic in rows, see ws_layout
  w.gemm = take(w.gemm_bytes);
  w.total = p;
  return w;
}

extern "C" size_t lia_llama_workspace_bytes(const lia_llama_desc* d, int max_rows) {
  if (check_llama_desc(d) || max_rows <= 0) return 0;
  return llama_ws(d, max_rows).total;
}

// tail = 1 (lia_llama_layer_forward_last): see layer_forward_impl -- norm, q | k | v projection and RoPE on every row, the rest on the
// last position of each row; y is [B, 1, H]
static int llama_layer_forward_impl(lia_ctx* ctx, const lia_llama_desc* d, const void* const weights[9], const lia_bf16* x,
                                    lia_bf16* y, lia_kv* kv, const lia_bf16* cos_table, const lia_bf16* sin_table, int B, int T,
                                    int pos0, int b0, void* stream, int tail) {
  if (!ctx) return LIA_ERR_INVALID;
  int rc = check_llama_desc(d);
  if (rc) return rc;
  if (!weights || !x || !y || !kv || !kv->k || !kv->v || !cos_table || !sin_table) { lia_set_error("lia_llama_layer_forward: NULL tensor"); return LIA_ERR_MISSING; }
  for (int i = 0; i < 9; ++i)
    if (!weights[i]) { lia_set_error("lia_llama_layer_forward: weights[%d] is NULL", i); return LIA_ERR_MISSING; }
  if (!kv->on_device) { lia_set_error("lia_llama_layer_forward: the KV cache must live on the device"); return LIA_ERR_INVALID; }
  if (B <= 0 || T <= 0 || pos0 < 0 || b0 < 0 || b0 + B > kv->batch || pos0 + T > kv->smax || (T > 1 && pos0 != 0)) {
    lia_set_error("lia_llama_layer_forward: B=%d T=%d pos0=%d b0=%d vs cache batch=%d smax=%d", B, T, pos0, b0, kv->batch, kv->smax);
    return LIA_ERR_INVALID;
  }
  if (tail && T < 2) { lia_set_error("lia_llama_layer_forward_last: a multi-token prefill only"); return LIA_ERR_INVALID; }
  const int H = d->hidden, F = d->ffn, hd = H / d->heads, KD = d->kv_heads * hd;
  const long M = (long)B * T;
  const long Mt = tail ? (long)B : M;
  const long tail_off = tail ? (long)(T - 1) * H : 0, tail_ld = tail ? (long)T * H : (long)H;
  const LlamaWs w = llama_ws(d, M);
  if (w.total > ctx->ws_bytes) { lia_set_error("lia_llama_layer_forward: workspace %zu < %zu", ctx->ws_bytes, w.total); return LIA_ERR_MEMORY; }
  hipStream_t st = (hipStream_t)stream;
  char* ws = ctx->ws;
  bf16_t *ln = (bf16_t*)(ws + w.ln), *qb = (bf16_t*)(ws + w.q), *ao = (bf16_t*)(ws + w.attn), *h1 = (bf16_t*)(ws + w.h1);
  bf16_t *gu = (bf16_t*)(ws + w.gu), *act = (bf16_t*)(ws + w.act);
  float* gws = (float*)(ws + w.gemm);
  const bf16_t* const* W = (const bf16_t* const*)weights;
  const LiaEpilogue none{nullptr, nullptr, 0, 0};
  // chaining state (see lia_layer_forward)
  const bf16_t* chain_g = ctx->chain_armed ? ctx->chain_g : nullptr;
  ctx->chain_armed = false;
  const bool have_norm1 = ctx->normed_src == (const void*)x && ctx->normed_buf == (const void*)ln && ctx->normed_rows == M && ctx->normed_h == H;
  ctx->normed_src = nullptr;

  if (!have_norm1) lia_rmsnorm_launch(x, H, W[0], ln, H, M, H, d->rms_eps, st);
  const bool fused_kv = W[3] == W[2] + (size_t)KD * H;
  const int G = d->heads / d->kv_heads;
  // q | k | v in ONE GEMM when the three weights are adjacent (they are in lia_llama_pack_offsets' layer buffer): the q columns
  // are G segments of the k / v width, the k and v segments scatter into the seq-major cache; in decode the rotation of the q and
  // k heads rides in the split-K combine
  const bool fused_qkv = fused_kv && W[2] == W[1] + (size_t)H * H && G + 2 <= LIA_OUT_SEGS;
  int rope_done = 0;
  if (fused_qkv) {
    LiaOutMap om;
    memset(&om, 0, sizeof(om));
    for (int j = 0; j < G; ++j) { om.base[j] = qb + (size_t)j * KD; om.ld[j] = H; }
    om.base[G] = kv->k; om.base[G + 1] = kv->v; om.ld[G] = om.ld[G + 1] = KD; om.cache_mode[G] = om.cache_mode[G + 1] = 1;
    om.seg_n = KD; om.T = T; om.Bc = kv->batch; om.b0 = b0; om.pos0 = pos0;
    LiaPost post{};
    post.kind = LIA_POST_ROPE; post.cos_t = cos_table; post.sin_t = sin_table; post.rot_heads = d->heads + d->kv_heads; post.hd = hd;
    post.pos0 = pos0; post.T = T;
    rc = gemm_checked(ctx, ln, H, W[1], (int)M, H + 2 * KD, H, none, om, gws, w.gemm_bytes, 0, st, &post, &rope_done);
    if (rc) return rc;
  } else {  // q projection
    LiaOutMap om = plain_out(qb, H, H);
    rc = gemm_checked(ctx, ln, H, W[1], (int)M, H, H, none, om, gws, w.gemm_bytes, 0, st);
    if (rc) return rc;
  }
  for (int part = 0; part < (fused_qkv ? 0 : (fused_kv ? 1 : 2)); ++part) {  // k | v projection, rows scattered into the seq-major cache
    LiaOutMap om;
    memset(&om, 0, sizeof(om));
    if (fused_kv) { om.base[0] = kv->k; om.base[1] = kv->v; om.cache_mode[0] = om.cache_mode[1] = 1; om.ld[0] = om.ld[1] = KD; }
    else { om.base[0] = part == 0 ? kv->k : kv->v; om.cache_mode[0] = 1; om.ld[0] = KD; }
    om.seg_n = KD; om.T = T; om.Bc = kv->batch; om.b0 = b0; om.pos0 = pos0;
    rc = gemm_checked(ctx, ln, H, fused_kv ? W[2] : W[2 + part], (int)M, fused_kv ? 2 * KD : KD, H, none, om, gws, w.gemm_bytes, 0, st);
    if (rc) return rc;
  }
  // RoPE on q (token rows b*T+t) and on the K rows just written (cache rows t*Bc + b); HF caches post-RoPE keys
  if (!rope_done) {
    lia_rope_launch(qb, H, cos_table, sin_table, M, d->heads, hd, pos0, T, 0, st);
    if (b0 == 0 && B == kv->batch) {
      lia_rope_launch(kv->k + (size_t)pos0 * kv->batch * KD, KD, cos_table, sin_table, (long)T * B, d->kv_heads, hd, pos0, 0, kv->batch, st);
    } else {
      for (int t = 0; t < T; ++t)  // minibatch slice of the cache rows: one launch per position
        lia_rope_launch(kv->k + ((size_t)(pos0 + t) * kv->batch + b0) * KD, KD, cos_table, sin_table, B, d->kv_heads, hd, pos0 + t, 0,
                        kv->batch + B, st);
    }
  }
  int arc = (T == 1 || tail) ? lia_attn_decode_launch(qb + tail_off, tail_ld, kv->k, kv->v, ao, H, B, pos0 + T, d->heads, d->kv_heads, hd, kv->batch, b0, 1, st)
                             : lia_attn_prefill_launch(qb, H, kv->k, kv->v, ao, H, B, T, d->heads, d->kv_heads, hd, kv->batch, b0, 1, st);
  if (arc) { lia_set_error("llama attention: unsupported head_dim %d / S %d", hd, pos0 + T); return LIA_ERR_INVALID; }
  int norm2_done = 0, silu_done = 0;
  {  // o_proj + residual; the post-attention RMSNorm rides in the split-K combine when there is one
    LiaEpilogue ep{nullptr, x + tail_off, tail_ld, 0};
    LiaOutMap om = plain_out(h1, H, H);
    LiaPost post{};
    post.kind = LIA_POST_RMSNORM; post.g = W[5]; post.eps = d->rms_eps; post.out = ln; post.ldo = H;
    rc = gemm_checked(ctx, ao, H, W[4], (int)Mt, H, H, ep, om, gws, w.gemm_bytes, 0, st, &post, &norm2_done);
    if (rc) return rc;
  }
  if (!norm2_done) lia_rmsnorm_launch(h1, H, W[5], ln, H, Mt, H, d->rms_eps, st);
  const bool fused_gu = W[7] == W[6] + (size_t)F * H;
  if (fused_gu) {
    LiaOutMap om = plain_out(gu, 2 * F, 2 * F);
    LiaPost post{};
    post.kind = LIA_POST_SILU_MUL; post.out = act; post.ldo = F;     // act = silu(gate) * up straight from the combine / the tiled epilogue
    post.gu_block = d->gu_block;
    rc = gemm_checked(ctx, ln, H, W[6], (int)Mt, 2 * F, H, none, om, gws, w.gemm_bytes, 0, st, &post, &silu_done);
    if (rc) return rc;
  } else {
    for (int part = 0; part < 2; ++part) {
      LiaOutMap om = plain_out(gu + (size_t)part * F, 2 * F, F);
      rc = gemm_checked(ctx, ln, H, W[6 + part], (int)Mt, F, H, none, om, gws, w.gemm_bytes, 0, st);
      if (rc) return rc;
    }
  }
  if (!silu_done) lia_silu_mul_launch(gu, act, Mt, F, fused_gu ? d->gu_block : 0, st);
  {  // down_proj + residual (+ the next layer's input RMSNorm when the caller chained it)
    LiaEpilogue ep{nullptr, h1, H, 0};
    LiaOutMap om = plain_out(y, H, H);
    LiaPost post{};
    int chained = 0;
    if (chain_g) { post.kind = LIA_POST_RMSNORM; post.g = chain_g; post.eps = d->rms_eps; post.out = ln; post.ldo = H; }
    rc = gemm_checked(ctx, act, F, W[8], (int)Mt, H, F, ep, om, gws, w.gemm_bytes, 0, st, post.kind ? &post : nullptr, &chained);
    if (rc) return rc;
    if (chained) { ctx->normed_src = y; ctx->normed_buf = ln; ctx->normed_rows = Mt; ctx->normed_h = H; }
  }
  HIP_TRY(hipGetLastError());
  return LIA_OK;
}

extern "C" int lia_llama_layer_forward(lia_ctx* ctx, const lia_llama_desc* d, const void* const weights[9], const lia_bf16* x,
                                       lia_bf16* y, lia_kv* kv, const lia_bf16* cos_table, const lia_bf16* sin_table, int B, int T,
                                       int pos0, int b0, void* stream) {
  return llama_layer_forward_impl(ctx, d, weights, x, y, kv, cos_table, sin_table, B, T, pos0, b0, stream, 0);
}

extern "C" int lia_llama_layer_forward_last(lia_ctx* ctx, const lia_llama_desc* d, const void* const weights[9], const lia_bf16* x,
                                            lia_bf16* y_last, lia_kv* kv, const lia_bf16* cos_table, const lia_bf16* sin_table, int B,
                                            int T, int pos0, int b0, void* stream) {
  return llama_layer_forward_impl(ctx, d, weights, x, y_last, kv, cos_table, sin_table, B, T, pos0, b0, stream, 1);
}

extern "C" int lia_llama_embed(const int64_t* ids, const lia_bf16* tok, lia_bf16* y, int B, int T, int H, void* stream) {
  if (!ids || !tok || !y) return LIA_ERR_MISSING;
  if (B <= 0 || T <= 0 || H % 8) return LIA_ERR_INVALID;
  lia_embed_tokens_launch(ids, tok, y, (long)B * T, H, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return LIA_OK;
}

extern "C" int lia_llama_lm_head(lia_ctx* ctx, const lia_bf16* hidden, int B, int T, int H, const lia_bf16* normw,
                                 const lia_bf16* lm, int vocab, float eps, int suppress_token, lia_bf16* logits, int64_t* next_ids,
                                 void* stream) {
  if (!ctx) return LIA_ERR_INVALID;
  if (!hidden || !normw || !lm || !logits || !next_ids) { lia_set_error("lia_llama_lm_head: NULL tensor"); return LIA_ERR_MISSING; }
  if (B <= 0 || B > 256 || T <= 0 || vocab % 16) { lia_set_error("lia_llama_lm_head: B=%d T=%d vocab=%d", B, T, vocab); return LIA_ERR_INVALID; }
  size_t scratch = align_up((size_t)B * H * 2, 256);
  if (ctx->ws_bytes < scratch) { lia_set_error("lia_llama_lm_head: workspace too small"); return LIA_ERR_MEMORY; }
  hipStream_t st = (hipStream_t)stream;
  ctx->normed_src = nullptr; ctx->chain_armed = false;   // the workspace is about to be reused
  bf16_t* lno = (bf16_t*)ctx->ws;
  lia_rmsnorm_launch(hidden + (long)(T - 1) * H, (long)T * H, normw, lno, H, B, H, eps, st);
  LiaEpilogue ep{nullptr, nullptr, 0, 0};
  LiaOutMap om = plain_out(logits, vocab, vocab);
  size_t have = ctx->ws_bytes - scratch;
  int rc = gemm_checked(ctx, lno, H, lm, B, vocab, H, ep, om, (float*)(ctx->ws + scratch), std::min(have, (size_t)8 * B * vocab * 4), 0, st);
  if (rc) return rc;
  lia_argmax_launch(logits, next_ids, B, vocab, suppress_token, st);
  HIP_TRY(hipGetLastError());
  return LIA_OK;
}

// ------------------------------------------------------------------------------------------------
// decode step over a run of HBM-resident layers in one call, layer by layer through the per-op route, each layer's last combine
// also computing the next layer's first norm (lia_ctx_chain_next_norm).  (r04 also had a persistent per-layer "chain" kernel
// behind a switch here; bit-identical, measured 4-9 % slower per step -- a seam inside a launch costs what a kernel boundary
// costs -- and retired in r05: LABNOTES.md, git history.)
// ------------------------------------------------------------------------------------------------
static int hidden_tmp_buffer(lia_ctx* ctx, size_t bytes, bf16_t** out) {
  if (ctx->hidden_tmp_bytes < bytes) {
    HIP_TRY(hipStreamSynchronize(ctx->compute));
    if (ctx->hidden_tmp) (void)hipFree(ctx->hidden_tmp);
    ctx->hidden_tmp = nullptr; ctx->hidden_tmp_bytes = 0;
    HIP_TRY(hipMalloc((void**)&ctx->hidden_tmp, bytes));
    ctx->hidden_tmp_bytes = bytes;
  }
  *out = (bf16_t*)ctx->hidden_tmp;
  return LIA_OK;
}

// hidden state h_l (the input of layer l; h_n = the result): h_0 = x, never written; h_n = y; in between y and the context's
// third buffer alternate so that a layer never writes the buffer it reads
static inline bf16_t* hidden_of(int l, int n, const bf16_t* x, bf16_t* y, bf16_t* t) { return l == 0 ? (bf16_t*)x : (((n - l) & 1) ? t : y); }

extern "C" int lia_llama_decode_layers(lia_ctx* ctx, const lia_llama_desc* d, int n_layers, const void* const* weights, const lia_bf16* x,
                                       lia_bf16* y, lia_kv* const* kv, const lia_bf16* cos_table, const lia_bf16* sin_table, int B, int pos0,
                                       void* stream) {
  if (!ctx) return LIA_ERR_INVALID;
  int rc = check_llama_desc(d);
  if (rc) return rc;
  if (n_layers <= 0 || !weights || !x || !y || !kv || !cos_table || !sin_table) { lia_set_error("lia_llama_decode_layers: NULL argument"); return LIA_ERR_MISSING; }
  for (int l = 0; l < n_layers; ++l) {
    if (!kv[l] || !kv[l]->k || !kv[l]->v) { lia_set_error("lia_llama_decode_layers: kv[%d] is NULL", l); return LIA_ERR_MISSING; }
    if (!kv[l]->on_device || B <= 0 || B > kv[l]->batch || pos0 < 0 || pos0 + 1 > kv[l]->smax) {
      lia_set_error("lia_llama_decode_layers: B=%d pos0=%d vs cache %d batch=%d smax=%d on_device=%d", B, pos0, l, kv[l]->batch, kv[l]->smax, kv[l]->on_device);
      return LIA_ERR_INVALID;
    }
    for (int i = 0; i < 9; ++i)
      if (!weights[l * 9 + i]) { lia_set_error("lia_llama_decode_layers: weights[%d][%d] is NULL", l, i); return LIA_ERR_MISSING; }
  }
  const int H = d->hidden, M = B;
  bf16_t* tmp = nullptr;
  rc = hidden_tmp_buffer(ctx, (size_t)M * H * 2, &tmp);
  if (rc) return rc;
  const bf16_t* const* Wall = (const bf16_t* const*)weights;

  for (int l = 0; l < n_layers; ++l) {
    if (l + 1 < n_layers) lia_ctx_chain_next_norm(ctx, Wall[(l + 1) * 9], nullptr);
    rc = llama_layer_forward_impl(ctx, d, (const void* const*)(Wall + l * 9), hidden_of(l, n_layers, x, y, tmp), hidden_of(l + 1, n_layers, x, y, tmp), kv[l],
                                  cos_table, sin_table, B, 1, pos0, 0, stream, 0);
    if (rc) return rc;
  }
  return LIA_OK;
}

// the OPT counterpart: policy 3 (everything on the GPU, KV cache in HBM) for n_layers consecutive resident layers, T == 1 --
// the loop over resident layers of OPTDecoder.forward (lia/modeling_opt.py:1246-1260) with decoder.py:172-335 /
// attentions.py:393-529 per layer
extern "C" int lia_decode_layers(lia_ctx* ctx, const lia_layer_desc* d, int n_layers, const void* const* weights, const lia_bf16* x,
                                 lia_bf16* y, lia_kv* const* kv, int B, int pos0, void* stream) {
  if (!ctx) return LIA_ERR_INVALID;
  int rc = check_desc(d);
  if (rc) return rc;
  if (n_layers <= 0 || !weights || !x || !y || !kv) { lia_set_error("lia_decode_layers: NULL argument"); return LIA_ERR_MISSING; }
  for (int l = 0; l < n_layers; ++l) {
    if (!kv[l] || !kv[l]->k || !kv[l]->v) { lia_set_error("lia_decode_layers: kv[%d] is NULL", l); return LIA_ERR_MISSING; }
    if (!kv[l]->on_device || B <= 0 || B > kv[l]->batch || pos0 < 0 || pos0 + 1 > kv[l]->smax) {
      lia_set_error("lia_decode_layers: B=%d pos0=%d vs cache %d batch=%d smax=%d on_device=%d", B, pos0, l, kv[l]->batch, kv[l]->smax, kv[l]->on_device);
      return LIA_ERR_INVALID;
    }
    for (int i = 0; i < 16; ++i)
      if (!weights[l * 16 + i]) { lia_set_error("lia_decode_layers: weights[%d][%d] is NULL", l, i); return LIA_ERR_MISSING; }
  }
  const int H = d->hidden, M = B;
  bf16_t* tmp = nullptr;
  rc = hidden_tmp_buffer(ctx, (size_t)M * H * 2, &tmp);
  if (rc) return rc;
  const bf16_t* const* Wall = (const bf16_t* const*)weights;
  for (int l = 0; l < n_layers; ++l) {
    if (l + 1 < n_layers) lia_ctx_chain_next_norm(ctx, Wall[(l + 1) * 16], Wall[(l + 1) * 16 + 1]);
    rc = layer_forward_impl(ctx, d, 3, (const void* const*)(Wall + l * 16), hidden_of(l, n_layers, x, y, tmp), hidden_of(l + 1, n_layers, x, y, tmp), kv[l], B, 1,
                            pos0, 0, stream, 0);
    if (rc) return rc;
  }
  return LIA_OK;
}

// ------------------------------------------------------------------------------------------------
// weight streamer
// ------------------------------------------------------------------------------------------------
extern "C" size_t lia_pack10_bound(size_t n_values);
extern "C" void lia_packed_decode_launch(const char* src, bf16_t* dst, size_t n_values, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);

struct lia_streamer {
  lia_ctx* ctx;
  int n_slots;
  size_t slot_bytes;
  char* slots;
  char* staging;          // per-slot landing area of pack10-encoded layers (lazily allocated)
  size_t staging_bytes;
  hipStream_t decode;     // the wire-format decode kernels run here, ordered after the slot's copy by an event
  std::vector<hipEvent_t> landed;
  std::vector<char> decoded_on_side;
  // live timing of the wire-format decode kernel (lia_stream_decode_stats): one event pair per slot around the MAIN decode kernel
  std::vector<hipEvent_t> d0, d1;
  std::vector<char> decode_pending;
  std::vector<double> decode_in, decode_out;      // bytes of the pending launch: encoded layer read, bf16 layer written
  double dec_ms, dec_in, dec_out;
  long dec_launches;
  hipStream_t copy;
  std::vector<hipEvent_t> copied, released, t0, t1;
  std::vector<char> has_release, timing_pending, was_marked;
  std::vector<size_t> pending_bytes;
  // pageable sources (the reference's un-pinned numa_alloc tensors, a layer kept without --pin-weight) are staged through two pinned
  // bounce buffers of LIA_BOUNCE_BYTES: a team memcpy fills one while the copy engine drains the other (staged_copy)
  char* bounce[2];
  hipEvent_t bounce_done[2];
  int bounce_next;
  double bytes, busy_ms;
};

static void streamer_collect_decode(lia_streamer* s, int slot, bool wait) {
  if (s->decode_pending.empty() || !s->decode_pending[slot]) return;
  if (!wait && hipEventQuery(s->d1[slot]) != hipSuccess) return;
  if (hipEventSynchronize(s->d1[slot]) == hipSuccess) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->d0[slot], s->d1[slot]) == hipSuccess) {
      s->dec_ms += ms; s->dec_in += s->decode_in[slot]; s->dec_out += s->decode_out[slot]; s->dec_launches++;
    }
  }
  s->decode_pending[slot] = 0;
}

static void streamer_collect(lia_streamer* s, int slot) {
  if (!s->timing_pending[slot]) return;
  if (hipEventSynchronize(s->t1[slot]) == hipSuccess) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->t0[slot], s->t1[slot]) == hipSuccess) {
      s->busy_ms += ms;
      s->bytes += (double)s->pending_bytes[slot];
    }
  }
  s->timing_pending[slot] = 0;
}

extern "C" int lia_stream_create(lia_ctx* ctx, int n_slots, size_t slot_bytes, lia_streamer** out) {
  if (!ctx || !out || n_slots < 1 || n_slots > 16 || slot_bytes == 0) { lia_set_error("lia_stream_create: n_slots=%d slot_bytes=%zu", n_slots, slot_bytes); return LIA_ERR_INVALID; }
  *out = nullptr;
  lia_streamer* s = new lia_streamer();
  s->ctx = ctx; s->n_slots = n_slots; s->slot_bytes = align_up(slot_bytes, 256);
  s->bounce[0] = s->bounce[1] = nullptr; s->bounce_done[0] = s->bounce_done[1] = nullptr; s->bounce_next = 0;
  s->bytes = 0; s->busy_ms = 0; s->slots = nullptr; s->staging = nullptr; s->staging_bytes = 0;
  s->dec_ms = s->dec_in = s->dec_out = 0; s->dec_launches = 0;
  HIP_TRY(hipMalloc((void**)&s->slots, s->slot_bytes * n_slots));
  if (ctx->serialized) s->copy = ctx->compute;
  else HIP_TRY(hipStreamCreateWithFlags(&s->copy, hipStreamNonBlocking));
  s->copied.resize(n_slots); s->released.resize(n_slots); s->t0.resize(n_slots); s->t1.resize(n_slots);
  s->has_release.assign(n_slots, 0); s->timing_pending.assign(n_slots, 0); s->pending_bytes.assign(n_slots, 0);
  s->was_marked.assign(n_slots, 0);
  for (int i = 0; i < n_slots; ++i) {
    HIP_TRY(hipEventCreateWithFlags(&s->copied[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s->released[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&s->t0[i]));
    HIP_TRY(hipEventCreate(&s->t1[i]));
  }
  *out = s;
  return LIA_OK;
}

extern "C" void lia_stream_destroy(lia_streamer* s) {
  if (!s) return;
  (void)hipStreamSynchronize(s->copy);
  for (int i = 0; i < s->n_slots; ++i) {
    (void)hipEventDestroy(s->copied[i]); (void)hipEventDestroy(s->released[i]);
    (void)hipEventDestroy(s->t0[i]); (void)hipEventDestroy(s->t1[i]);
  }
  for (int i = 0; i < 2; ++i) {
    if (s->bounce[i]) (void)hipHostFree(s->bounce[i]);
    if (s->bounce_done[i]) (void)hipEventDestroy(s->bounce_done[i]);
  }
  if (s->slots) (void)hipFree(s->slots);
  if (s->staging) {
    (void)hipStreamSynchronize(s->decode);
    for (hipEvent_t e : s->landed) (void)hipEventDestroy(e);
    for (hipEvent_t e : s->d0) (void)hipEventDestroy(e);
    for (hipEvent_t e : s->d1) (void)hipEventDestroy(e);
    if (s->decode != s->ctx->compute) (void)hipStreamDestroy(s->decode);
    (void)hipFree(s->staging);
  }
  if (s->copy != s->ctx->compute) (void)hipStreamDestroy(s->copy);
  delete s;
}

extern "C" void* lia_stream_slot_ptr(lia_streamer* s, int slot) {
  if (!s || slot < 0 || slot >= s->n_slots) return nullptr;
  return s->slots + (size_t)slot * s->slot_bytes;
}

extern "C" void* lia_stream_copy_stream(lia_streamer* s) { return s ? (void*)s->copy : nullptr; }

// prefetch = begin + copy_chunk(whole layer) + mark_ready.  The three-step form lets a data-parallel caller
// interleave RCCL broadcasts of the chunks (on streams ordered after the copy stream) before the slot is
// declared ready.
extern "C" int lia_stream_begin(lia_streamer* s, int slot) {
  if (!s || slot < 0 || slot >= s->n_slots) { lia_set_error("lia_stream_begin: slot=%d", slot); return LIA_ERR_INVALID; }
  streamer_collect(s, slot);
  if (s->has_release[slot]) HIP_TRY(hipStreamWaitEvent(s->copy, s->released[slot], 0));
  // A prefetch the caller dropped (never waited for, never released) may still have its decode kernel running on the side
  // stream, reading this slot's staging area and writing the slot: the new copy must land behind it as well.
  if (s->was_marked[slot]) HIP_TRY(hipStreamWaitEvent(s->copy, s->copied[slot], 0));
  HIP_TRY(hipEventRecord(s->t0[slot], s->copy));
  s->pending_bytes[slot] = 0;
  return LIA_OK;
}

// Pageable source -> device through the two pinned bounce buffers (the reference's cpu_buff, modeling_opt.py:1219-1220, and what
// `copy_(non_blocking=True)` from its un-pinned numa_alloc tensors does inside torch).  r01-r04: one slot-sized bounce buffer, a
// single-threaded memcpy behind a stream synchronize -- 16.6 GB/s in the H2D microbenchmark twin where torch's own staging reaches
// 51.  Now 64 MiB pieces, a team memcpy (lia_host_parallel_memcpy, the context's host threads) into one buffer while the copy engine
// drains the other; the host thread waits only for the buffer it is about to overwrite.
constexpr size_t LIA_BOUNCE_BYTES = (size_t)64 << 20;
extern "C" void lia_host_parallel_memcpy(void* dst, const void* src, size_t bytes, int n_threads);
static int staged_copy(lia_streamer* s, char* dst_device, const void* host_ptr, size_t bytes) {
  for (int i = 0; i < 2; ++i)
    if (!s->bounce[i]) {
      HIP_TRY(hipHostMalloc((void**)&s->bounce[i], LIA_BOUNCE_BYTES, hipHostMallocDefault));
      HIP_TRY(hipEventCreateWithFlags(&s->bounce_done[i], hipEventDisableTiming));
      HIP_TRY(hipEventRecord(s->bounce_done[i], s->copy));
    }
  for (size_t off = 0; off < bytes; off += LIA_BOUNCE_BYTES) {
    const size_t n = std::min(LIA_BOUNCE_BYTES, bytes - off);
    const int b = s->bounce_next;
    s->bounce_next ^= 1;
    HIP_TRY(hipEventSynchronize(s->bounce_done[b]));                 // the DMA that last read this buffer has finished
    lia_host_parallel_memcpy(s->bounce[b], (const char*)host_ptr + off, n, s->ctx->host_threads);
    HIP_TRY(hipMemcpyAsync(dst_device + off, s->bounce[b], n, hipMemcpyHostToDevice, s->copy));
    HIP_TRY(hipEventRecord(s->bounce_done[b], s->copy));
  }
  return LIA_OK;
}

extern "C" int lia_stream_copy_chunk(lia_streamer* s, int slot, size_t offset, const void* host_ptr, size_t bytes, int pinned) {
  if (!s || slot < 0 || slot >= s->n_slots || !host_ptr || offset + bytes > s->slot_bytes) {
    lia_set_error("lia_stream_copy_chunk: slot=%d offset=%zu bytes=%zu (slot holds %zu)", slot, offset, bytes, s ? s->slot_bytes : 0);
    return LIA_ERR_INVALID;
  }
  char* const dst = s->slots + (size_t)slot * s->slot_bytes + offset;
  if (pinned) HIP_TRY(hipMemcpyAsync(dst, host_ptr, bytes, hipMemcpyHostToDevice, s->copy));
  else if (int rc = staged_copy(s, dst, host_ptr, bytes)) return rc;
  s->pending_bytes[slot] += bytes;
  return LIA_OK;
}

// packed path: the encoded layer lands in the slot's staging area, then a kernel on a side stream rebuilds the
// bf16 layer in the slot itself (lia_pack10.hip).  begin -> copy_chunk_packed* -> decode_packed -> mark_ready.
static int ensure_staging(lia_streamer* s) {
  if (s->staging) return LIA_OK;
  s->staging_bytes = lia_pack10_bound((s->slot_bytes / 2 + 1023) / 1024 * 1024);
  HIP_TRY(hipMalloc((void**)&s->staging, s->staging_bytes * s->n_slots));
  // The wire-format decode runs on a stream of its own, ordered by events, so the copy engine moves on to the next layer at once.
  // (r02-r04 could confine that stream to a few compute units with a CU mask: measured level or worse at every width -- k masked CUs
  // for 256 / k times as long cost the prefill GEMM the same CU-milliseconds -- and removed in r05; what did pay was making the
  // decode kernel itself cheaper, lia_pack10.hip.  LABNOTES.md r04 / r05.)
  if (s->ctx->serialized) s->decode = s->ctx->compute;
  else HIP_TRY(hipStreamCreateWithFlags(&s->decode, hipStreamNonBlocking));
  s->landed.resize(s->n_slots);
  s->decoded_on_side.assign(s->n_slots, 0);
  s->d0.resize(s->n_slots); s->d1.resize(s->n_slots);
  s->decode_pending.assign(s->n_slots, 0); s->decode_in.assign(s->n_slots, 0.0); s->decode_out.assign(s->n_slots, 0.0);
  for (int i = 0; i < s->n_slots; ++i) {
    HIP_TRY(hipEventCreateWithFlags(&s->landed[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&s->d0[i]));
    HIP_TRY(hipEventCreate(&s->d1[i]));
  }
  return LIA_OK;
}

extern "C" void* lia_stream_staging_ptr(lia_streamer* s, int slot) {
  if (!s || slot < 0 || slot >= s->n_slots || ensure_staging(s)) return nullptr;
  return s->staging + (size_t)slot * s->staging_bytes;
}

extern "C" int lia_stream_copy_chunk_packed(lia_streamer* s, int slot, size_t offset, const void* host_ptr, size_t bytes, int pinned) {
  if (!s || slot < 0 || slot >= s->n_slots || !host_ptr) return LIA_ERR_INVALID;
  int rc = ensure_staging(s);
  if (rc) return rc;
  if (offset + bytes > s->staging_bytes) { lia_set_error("lia_stream_copy_chunk_packed: %zu + %zu > staging %zu", offset, bytes, s->staging_bytes); return LIA_ERR_INVALID; }
  char* const dst = s->staging + (size_t)slot * s->staging_bytes + offset;
  if (pinned) HIP_TRY(hipMemcpyAsync(dst, host_ptr, bytes, hipMemcpyHostToDevice, s->copy));
  else if (int rc = staged_copy(s, dst, host_ptr, bytes)) return rc;
  s->pending_bytes[slot] += bytes;
  return LIA_OK;
}

extern "C" int lia_stream_decode_packed(lia_streamer* s, int slot, size_t n_values, int format) {
  if (!s || slot < 0 || slot >= s->n_slots || !s->staging || n_values * 2 > s->slot_bytes || format != 10 || (n_values % 1024)) {
    lia_set_error("lia_stream_decode_packed: slot=%d n_values=%zu format=%d", slot, n_values, format);
    return LIA_ERR_INVALID;
  }
  // The decode runs on its own stream behind an event, so the copy engine moves on to the next layer at once.
  // It must not start before the slot's previous consumer released it: begin() made the COPY stream wait for that,
  // and `landed` is recorded on the copy stream after the copy, so the order is inherited.
  HIP_TRY(hipEventRecord(s->t1[slot], s->copy));                       // copy-engine busy time ends here
  HIP_TRY(hipEventRecord(s->landed[slot], s->copy));
  HIP_TRY(hipStreamWaitEvent(s->decode, s->landed[slot], 0));
  // the timing bracket re-uses the slot's event pair: collect the previous launch first -- WITHOUT waiting (this runs inside the
  // token loop).  The slot's previous decode finished long ago (its consumer released the slot); should it not have -- a dropped
  // prefetch whose decode is still in flight -- this launch simply goes untimed.
  streamer_collect_decode(s, slot, false);
  (void)hipGetLastError();                                             // hipErrorNotReady from the query is an answer, not an error
  const bool timed = !s->decode_pending[slot];
  lia_packed_decode_launch(s->staging + (size_t)slot * s->staging_bytes, (bf16_t*)(s->slots + (size_t)slot * s->slot_bytes), n_values, s->decode,
                           timed ? s->d0[slot] : nullptr, timed ? s->d1[slot] : nullptr);
  HIP_TRY(hipGetLastError());
  if (timed) {
    s->decode_in[slot] = (double)s->pending_bytes[slot];
    s->decode_out[slot] = 2.0 * (double)n_values;
    s->decode_pending[slot] = 1;
  }
  s->decoded_on_side[slot] = 1;
  return LIA_OK;
}

extern "C" int lia_pack_decode(const char* src_device, lia_bf16* dst_device, size_t n_values, int format, void* stream) {
  if (!src_device || !dst_device) { lia_set_error("lia_pack_decode: NULL buffer"); return LIA_ERR_MISSING; }
  if (format != 10 || (n_values % 1024)) {
    lia_set_error("lia_pack_decode: n_values=%zu format=%d", n_values, format);
    return LIA_ERR_INVALID;
  }
  lia_packed_decode_launch(src_device, dst_device, n_values, (hipStream_t)stream, nullptr, nullptr);
  HIP_TRY(hipGetLastError());
  return LIA_OK;
}

extern "C" int lia_stream_prefetch_packed(lia_streamer* s, int slot, const void* host_ptr, size_t packed_bytes, size_t n_values, int format,
                                          int pinned) {
  int rc = lia_stream_begin(s, slot);
  if (rc) return rc;
  rc = lia_stream_copy_chunk_packed(s, slot, 0, host_ptr, packed_bytes, pinned);
  if (rc) return rc;
  rc = lia_stream_decode_packed(s, slot, n_values, format);
  if (rc) return rc;
  return lia_stream_mark_ready(s, slot);
}

extern "C" int lia_stream_mark_ready(lia_streamer* s, int slot) {
  if (!s || slot < 0 || slot >= s->n_slots) return LIA_ERR_INVALID;
  if (s->staging && s->decoded_on_side[slot]) {
    HIP_TRY(hipEventRecord(s->copied[slot], s->decode));              // ready = decoded
    s->decoded_on_side[slot] = 0;
    s->timing_pending[slot] = 1;
    s->was_marked[slot] = 1;
    return LIA_OK;
  }
  HIP_TRY(hipEventRecord(s->t1[slot], s->copy));
  HIP_TRY(hipEventRecord(s->copied[slot], s->copy));
  s->timing_pending[slot] = 1;
  s->was_marked[slot] = 1;
  return LIA_OK;
}

extern "C" int lia_stream_prefetch(lia_streamer* s, int slot, const void* host_ptr, size_t bytes, int pinned) {
  if (!s || slot < 0 || slot >= s->n_slots || !host_ptr || bytes > s->slot_bytes) {
    lia_set_error("lia_stream_prefetch: slot=%d bytes=%zu (slot holds %zu)", slot, bytes, s ? s->slot_bytes : 0);
    return LIA_ERR_INVALID;
  }
  int rc = lia_stream_begin(s, slot);
  if (rc) return rc;
  rc = lia_stream_copy_chunk(s, slot, 0, host_ptr, bytes, pinned);
  if (rc) return rc;
  return lia_stream_mark_ready(s, slot);
}

extern "C" int lia_stream_wait(lia_streamer* s, int slot, void* compute_stream) {
  if (!s || slot < 0 || slot >= s->n_slots) return LIA_ERR_INVALID;
  HIP_TRY(hipStreamWaitEvent((hipStream_t)compute_stream, s->copied[slot], 0));
  return LIA_OK;
}

extern "C" int lia_stream_release(lia_streamer* s, int slot, void* compute_stream) {
  if (!s || slot < 0 || slot >= s->n_slots) return LIA_ERR_INVALID;
  HIP_TRY(hipEventRecord(s->released[slot], (hipStream_t)compute_stream));
  s->has_release[slot] = 1;
  return LIA_OK;
}

extern "C" int lia_stream_stats(lia_streamer* s, double* bytes, double* busy_ms, int reset) {
  if (!s) return LIA_ERR_INVALID;
  for (int i = 0; i < s->n_slots; ++i) streamer_collect(s, i);
  if (bytes) *bytes = s->bytes;
  if (busy_ms) *busy_ms = s->busy_ms;
  if (reset) { s->bytes = 0; s->busy_ms = 0; }
  return LIA_OK;
}

// Live timing of the wire-format decode kernel (the largest GPU consumer of a link-bound decode step; bench.py's
// roofline.dominant_kernel): HIP events on the decode stream around the main decode kernel of every packed layer since the last
// reset.  bytes_in = encoded bytes the kernel read, bytes_out = bf16 bytes it wrote (its algorithmic traffic: each once).
extern "C" int lia_stream_decode_stats(lia_streamer* s, long* launches, double* ms, double* bytes_in, double* bytes_out, int reset) {
  if (!s) return LIA_ERR_INVALID;
  for (int i = 0; i < s->n_slots; ++i) streamer_collect_decode(s, i, true);
  if (launches) *launches = s->dec_launches;
  if (ms) *ms = s->dec_ms;
  if (bytes_in) *bytes_in = s->dec_in;
  if (bytes_out) *bytes_out = s->dec_out;
  if (reset) { s->dec_ms = s->dec_in = s->dec_out = 0; s->dec_launches = 0; }
  return LIA_OK;
}

// The non-blocking form: only copies whose end event has already completed are counted (a queued copy stays pending and is
// counted by a later call).  For callers inside the token loop, which must never wait for the prefetched layers.
extern "C" int lia_stream_poll_decode_stats(lia_streamer* s, long* launches, double* ms, double* bytes_in, double* bytes_out, int reset) {
  if (!s) return LIA_ERR_INVALID;
  for (int i = 0; i < s->n_slots; ++i) streamer_collect_decode(s, i, false);
  (void)hipGetLastError();   // hipErrorNotReady from the query is an answer, not an error
  if (launches) *launches = s->dec_launches;
  if (ms) *ms = s->dec_ms;
  if (bytes_in) *bytes_in = s->dec_in;
  if (bytes_out) *bytes_out = s->dec_out;
  if (reset) { s->dec_ms = s->dec_in = s->dec_out = 0; s->dec_launches = 0; }
  return LIA_OK;
}

extern "C" int lia_stream_poll_stats(lia_streamer* s, double* bytes, double* busy_ms) {
  if (!s) return LIA_ERR_INVALID;
  for (int i = 0; i < s->n_slots; ++i)
    if (s->timing_pending[i] && hipEventQuery(s->t1[i]) == hipSuccess) streamer_collect(s, i);
  (void)hipGetLastError();   // hipErrorNotReady from the query is an answer, not an error
  if (bytes) *bytes = s->bytes;
  if (busy_ms) *busy_ms = s->busy_ms;
  return LIA_OK;
}

// ------------------------------------------------------------------------------------------------
// pinned / registered host memory
// ------------------------------------------------------------------------------------------------
extern "C" void* lia_host_alloc_pinned(size_t size) {
  void* p = nullptr;
  hipError_t e = hipHostMalloc(&p, size, hipHostMallocDefault);
  if (e != hipSuccess) {
    lia_set_error("hipHostMalloc(%zu) failed: %s", size, hipGetErrorString(e));
    return nullptr;
  }
  return p;
}
extern "C" void lia_host_free_pinned(void* p) { if (p) (void)hipHostFree(p); }

extern "C" int lia_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes) {
  if (!dst_device || !src_host) return LIA_ERR_MISSING;
  HIP_TRY(hipMemcpy(dst_device, src_host, bytes, hipMemcpyHostToDevice));
  return LIA_OK;
}
// Small transfer between device memory and MAPPED pinned host memory by a kernel on `stream` (16-byte multiples):
// the copy engines belong to the weight stream, a hipMemcpy would queue behind a 0.8-1.2 GB layer copy.
extern "C" int lia_blit(void* dst, const void* src, size_t bytes, void* stream) {
  if (!dst || !src) return LIA_ERR_MISSING;
  if (bytes % 16) { lia_set_error("lia_blit: %zu bytes is not a multiple of 16", bytes); return LIA_ERR_INVALID; }
  lia_blit_launch(dst, src, bytes, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return LIA_OK;
}

extern "C" int lia_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes) {
  if (!dst_host || !src_device) return LIA_ERR_MISSING;
  HIP_TRY(hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost));
  return LIA_OK;
}

extern "C" int lia_numa_register(void* ptr, size_t size) {
  if (!ptr || !size) return LIA_ERR_INVALID;
  hipError_t e = hipHostRegister(ptr, size, hipHostRegisterDefault);
  if (e != hipSuccess) {
    lia_set_error("hipHostRegister(%p, %zu) failed: %s", ptr, size, hipGetErrorString(e));
    return LIA_ERR_MEMORY;
  }
  return LIA_OK;
}
// Register a READ-ONLY mapping (a shared, PROT_READ mmap of a checkpoint file): the pages are pinned without write intent, so
// they stay the page-cache pages -- no private copy of the file appears in anonymous memory.
extern "C" int lia_numa_register_readonly(void* ptr, size_t size) {
  if (!ptr || !size) return LIA_ERR_INVALID;
  hipError_t e = hipHostRegister(ptr, size, hipHostRegisterReadOnly);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    lia_set_error("hipHostRegister(%p, %zu, read-only) failed: %s", ptr, size, hipGetErrorString(e));
    return LIA_ERR_MEMORY;
  }
  return LIA_OK;
}
extern "C" int lia_numa_unregister(void* ptr) {
  if (!ptr) return LIA_ERR_INVALID;
  HIP_TRY(hipHostUnregister(ptr));
  return LIA_OK;
}
