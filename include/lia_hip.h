/*
 * lia_hip.h -- C ABI of liblia_hip.so: the MI355X-native replacement for the GPU side and the
 * host-memory tiers of LIA's weight-offloaded cooperative decoder.
 *
 * Plain pointers and sizes only; no torch / HIP types in any signature (streams travel as void*,
 * which is a hipStream_t).  Every function returns 0 (LIA_OK) or a negative LIA_ERR_* code and never
 * throws; lia_last_error() gives the text.  The caller owns every buffer it passes in; the library
 * owns only its context, streams, events and workspace.  A context is re-entrant across contexts, not
 * thread-safe within one (the reference drives everything from one Python thread,
 * lia/modeling_opt.py:1208-1212).
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference tree;
 *   decoder.py    = intel_extension_for_pytorch/transformers/models/reference/modules/decoder.py
 *   attentions.py = intel_extension_for_pytorch/transformers/models/reference/modules/attentions.py
 *   modeling_opt.py = lia/modeling_opt.py).
 * INTEGRATION.md shows the ctypes stubs a reference maintainer would add.
 *
 * All activations / weights are bf16 (uint16 bit patterns).  Linear weights are ROW-MAJOR [N, K]
 * (the reference's TPP-blocked [N/16,K/64,32,16,2] wire format, _weight_prepack.py:19-63, is converted
 * once on the host by lia_tpp_unblock, not on every use as attentions.py:381-382 does).
 */
#ifndef LIA_HIP_H
#define LIA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIA_OK 0
#define LIA_ERR_INVALID (-1) /* bad shape / argument           -> Python ValueError  (attentions.py:503,516,532) */
#define LIA_ERR_MEMORY (-2)  /* allocation / pinning failed    -> Python MemoryError (modeling_opt.py:175)       */
#define LIA_ERR_HIP (-3)     /* HIP runtime error              -> Python RuntimeError                            */
#define LIA_ERR_MISSING (-4) /* a required tensor is NULL      -> Python AttributeError (modeling_opt.py:110,126) */

typedef uint16_t lia_bf16;
typedef struct lia_ctx lia_ctx;
typedef struct lia_streamer lia_streamer;

const char* lia_last_error(void);
const char* lia_version(void);

/* ---- context ----------------------------------------------------------------------------------
 * Replaces the per-forward stream/buffer setup of OPTDecoder.forward (modeling_opt.py:1178-1220).
 * workspace_bytes: device scratch for one layer call (see lia_layer_workspace_bytes). */
int lia_ctx_create(int device, size_t workspace_bytes, lia_ctx** out);
void lia_ctx_destroy(lia_ctx* ctx);
void* lia_ctx_compute_stream(lia_ctx* ctx); /* hipStream_t created by the context */
int lia_ctx_synchronize(lia_ctx* ctx);          /* compute AND K/V delivery streams */
int lia_ctx_synchronize_compute(lia_ctx* ctx);  /* the compute stream only (deferred K/V deliveries keep running) */
int lia_ctx_set_host_threads(lia_ctx* ctx, int n); /* OpenMP threads of the policy-2 host attention */
/* 1 when the context was created under LIA_SERIALIZE=1 (debug): the K/V delivery stream, and the copy / wire-decode streams of
 * every streamer created on it, ARE the compute stream, so no ordering depends on an event.  The reference serialises with
 * torch.cuda.synchronize() after every minibatch / layer (modeling_opt.py:1298,1339,1506,1528); a result that differs between
 * the two modes is a missing cross-stream dependency. */
int lia_ctx_serialized(lia_ctx* ctx);
/* Cross-layer chaining for decode (build-defined; the reference launches every LayerNorm as its own kernel, decoder.py:199-206).
 * One-shot promise about the NEXT lia_layer_forward / lia_llama_layer_forward call on this context: its output y will be the
 * input x of the call after it, unchanged, and that layer's first norm has weights g (and b; NULL for RMSNorm).  The layer then
 * normalises y inside the split-K combine of its last GEMM, and the following call (same x pointer, rows and width; anything
 * else falls back to the stand-alone norm kernel) skips its first norm.  g / b must be readable on the compute stream when the
 * hinted call is issued: chain resident layers, not a streamed layer whose copy may still be in flight. */
int lia_ctx_chain_next_norm(lia_ctx* ctx, const lia_bf16* g, const lia_bf16* b);

/* Live kernel timing for the benchmark's roofline report: while enabled, every GEMM main-kernel launch made
 * through this context is bracketed by HIP events on the stream it is launched on. */
typedef struct {
  long skinny_launches; double skinny_ms, skinny_bytes, skinny_flops; /* decode regime (M <= 256)  */
  long tiled_launches;  double tiled_ms, tiled_bytes, tiled_flops;    /* prefill regime            */
  double empty_bracket_ms; /* what an event pair around NOTHING reads on this stream: the bracket's own cost, measured at stop */
  long host_attention_calls; double host_attention_ms; /* policy-2 host attention (wall clock of lia_host_attention inside lia_layer_forward) */
} lia_prof_result;
int lia_prof_start(lia_ctx* ctx, int max_launches);
/* Bracket every stride-th GEMM launch only (default 1).  An event pair on the compute stream costs two ~6 us idle gaps, which is
 * the whole step's slack in the all-resident configurations; a stride co-prime to the launches per step samples every shape. */
int lia_prof_set_stride(lia_ctx* ctx, int stride);
int lia_prof_stop(lia_ctx* ctx, lia_prof_result* out); /* synchronises the compute stream */

/* ---- layer description and the 16-tensor weight set --------------------------------------------
 * Order fixed by create_buffer (modeling_opt.py:90-126) and consumed by index in decoder.py /
 * attentions.py:  0 ln1.w 1 ln1.b 2 q.w 3 q.b 4 k.w 5 k.b 6 v.w 7 v.b 8 out.w 9 out.b
 *                10 ln2.w 11 ln2.b 12 fc1.w 13 fc1.b 14 fc2.w 15 fc2.b                        */
typedef struct {
  int hidden; /* H */
  int heads;  /* h, head_dim = H / h in {32, 64, 128} */
  int ffn;    /* F */
  float ln_eps;
} lia_layer_desc;

/* Byte offsets of the 16 tensors inside one packed, 256-byte-aligned flat layer buffer -- the unit
 * the weight streamer moves (replaces 16 separate copy_ calls of load_layer, modeling_opt.py:270-293).
 * q.w|k.w|v.w and q.b|k.b|v.b are adjacent so the projection runs as one [3H,H] GEMM. */
int lia_layer_pack_offsets(const lia_layer_desc* d, size_t offsets[16], size_t* total_bytes);
size_t lia_layer_workspace_bytes(const lia_layer_desc* d, int max_rows /* B*T of one call */);

/* KV cache of one layer: seq-major [smax][batch][heads][head_dim] exactly like the reference's
 * past_key_value[1], [2] (attentions.py:462-476); the length is an explicit int instead of the shape of
 * a dummy tensor (attentions.py:464-470). */
typedef struct {
  lia_bf16* k;
  lia_bf16* v;
  int smax;
  int batch;     /* batch rows of the cache (row pitch) */
  int on_device; /* 1: k/v are device pointers (policy 3); 0: host pointers (policy 0 / 2)  */
} lia_kv;

/* ---- the operator boundary ----------------------------------------------------------------------
 * One decoder layer = decoder_layer(hidden, attention_mask=, past_key_value=, gpu_layer=, policy=,
 * max_new_tokens=) i.e. _IPEXDecoderLayerRef.forward -> OPTDecoderLayer_forward (decoder.py:172-335)
 * + _OPTAttention_forward (attentions.py:312-557).
 *
 *   policy 0 : everything on the GPU, K/V rows delivered to a HOST cache (prefill: rows [0,T) of batch
 *              rows [b0, b0+B) via the context's D2H stream; completion = lia_ctx_kv_store_wait)
 *   policy 3 : everything on the GPU, cache on the device (resident layers)
 *   policy 2 : LN + linears on the GPU, attention on the host over the host cache (decode only)
 *   policy 1 : not a GPU policy -- LIA_ERR_INVALID (the all-CPU path lives in lia_host_*)
 *
 * weights[16]: DEVICE pointers (resident copy or a streamer slot).  x, y: device [B, T, H], may not
 * alias.  pos0 = tokens already in the cache; rows pos0..pos0+T-1 are written, 0..pos0+T-1 attended
 * (causal inside the new block).  b0 = first cache batch row served by this call (minibatch offset,
 * modeling_opt.py:1283-1339).  Asynchronous on `stream` except policy 2, which blocks the caller while
 * the host attends. */
int lia_layer_forward(lia_ctx* ctx, const lia_layer_desc* d, int policy, const void* const weights[16],
                      const lia_bf16* x, lia_bf16* y, lia_kv* kv, int B, int T, int pos0, int b0, void* stream);
/* The LAST decoder layer of a prefill (build-defined; the reference computes every position of every layer, decoder.py:172-335):
 * only hidden[:, -1, :] of the last layer feeds the final LayerNorm and lm_head (models.py:424-431), but its K/V rows of EVERY
 * position feed the decode steps.  So: LN1 and the q | k | v projection on all B x T rows (cache written exactly as by
 * lia_layer_forward), then attention -- one query per row over the T keys --, out-proj, LN2, fc1 and fc2 on the last position
 * of each row only.  y_last: device [B, 1, H].  policy 0 or 3, T > 1, pos0 == 0.  Saves 9/12 of the layer's GEMM flops. */
int lia_layer_forward_last(lia_ctx* ctx, const lia_layer_desc* d, int policy, const void* const weights[16],
                           const lia_bf16* x, lia_bf16* y_last, lia_kv* kv, int B, int T, int pos0, int b0, void* stream);
/* dst / src: device memory or pinned (mapped) host memory; bytes % 16 == 0; asynchronous on `stream`.  Kernel copy used for
 * the small activation hops of the cooperative policies (modeling_opt.py:320-355 load_activation / store_hidden). */
int lia_blit(void* dst, const void* src, size_t bytes, void* stream);
int lia_ctx_kv_store_wait(lia_ctx* ctx); /* host-blocks until every policy-0 K/V delivery has landed */
/* Deferred form of the policy-0 K/V delivery (store_cache, lia/modeling_opt.py:334-345): the prefill of a streamed layer writes
 * its K/V rows into a DEVICE holding cache (policy 3 arithmetic, `dev`: on_device = 1, same batch as `host`), and this call
 * enqueues the copy of rows [0, T) into the host cache behind everything already queued on the compute stream.  The scheduler
 * issues the deliveries after the prefill's last layer, so the 23 GB of K/V do not share the host link with the weight stream
 * of the prefill (H2D runs 8 % slower beside them); the first decode step waits per layer with lia_kv_deliver_wait(ticket)
 * right before that layer's host attention.  row_elems = heads * head_dim. */
int lia_kv_deliver(lia_ctx* ctx, const lia_kv* dev, lia_kv* host, int T, int row_elems, int* ticket);
int lia_kv_deliver_wait(lia_ctx* ctx, int ticket); /* host-blocks until that delivery has landed; frees the ticket */
/* device time (first copy's start -> last copy's end, on the delivery stream) of the deliveries issued since the last call; blocks
 * until they have landed.  What the reference's first-token latency contains and this build's does not (bench.py kv_delivery). */
int lia_kv_deliver_batch_ms(lia_ctx* ctx, double* ms);

/* ---- individual sub-layer ops (same kernels the layer call uses; exposed for parity tests) ------ */
/* F.layer_norm, decoder.py:107-119 */
int lia_layernorm(const lia_bf16* x, long ldx, const lia_bf16* g, const lia_bf16* b, lia_bf16* y, long ldy, long rows,
                  int H, float eps, void* stream);
/* torch.matmul(x, w.t()) + bias [relu] [residual + .], decoder.py:79-105,229,310; attentions.py:393-394,418.
 * y[M,N] row-major with leading dimension ldy.  bias / residual may be NULL.  split_k: 0 = heuristic. */
int lia_linear(lia_ctx* ctx, const lia_bf16* x, long ldx, const lia_bf16* w, const lia_bf16* bias,
               const lia_bf16* residual, long ldr, lia_bf16* y, long ldy, int M, int N, int K, int relu, int split_k,
               void* stream);
/* fused q|k|v projection: q -> qout [M,H]; k,v rows scattered into the seq-major cache slab
 * (attentions.py:393-394,418,457-458,475-476,490-491).  w = [3H,H], bias = [3H]. */
int lia_qkv_project(lia_ctx* ctx, const lia_bf16* x, const lia_bf16* w, const lia_bf16* bias, lia_bf16* qout,
                    lia_bf16* kcache, lia_bf16* vcache, int B, int T, int H, int cache_batch, int b0, int pos0,
                    void* stream);
/* GPU attention, attentions.py:443-536.  q [B,T,H] (ldq elements per token row); cache [S][cache_batch][h][d].
 * head_dim 32 / 64 / 128.  LIA_ERR_INVALID (by name in lia_last_error) for any other head_dim and, for a head_dim-128 prefill
 * (T > 1), for a cache whose key row -- cache_batch x heads x head_dim values -- exceeds 3.3e7 values (64 key rows = 4 GiB:
 * the kernel's 32-bit staging offsets; OPT-175B at a cache batch of 2048 is 2.5e7). */
int lia_attention(const lia_bf16* q, long ldq, const lia_bf16* kcache, const lia_bf16* vcache, lia_bf16* out, long ldo,
                  int B, int T, int S, int heads, int head_dim, int cache_batch, int b0, void* stream);
/* embed_tokens + embed_positions, modeling_opt.py:1108,357-378,1142 */
int lia_embed(const int64_t* ids, const lia_bf16* tok, const lia_bf16* pos, lia_bf16* y, int B, int T, int past_len, int H,
              void* stream);
/* final LN on the last position + tied lm_head + greedy argmax: modeling_opt.py:1563, models.py:424-431,
 * greedy_search.py:367,395.  logits [B,vocab], next_ids [B] (device).  suppress_token >= 0: that token's score
 * counts as -inf (EOS while min_new_tokens is unmet, run_generation.py:173,179-182); -1: none. */
int lia_lm_head(lia_ctx* ctx, const lia_bf16* hidden, int B, int T, int H, const lia_bf16* lnw, const lia_bf16* lnb,
                const lia_bf16* emb, int vocab, float eps, int suppress_token, lia_bf16* logits, int64_t* next_ids,
                void* stream);

/* ---- Llama-family layer (BASELINE.json config 4; build-defined: the reference's LlamaDecoderLayer_forward,
 * decoder.py:121-169, has no policy plumbing -- SURVEY.md quirk 3).  Arithmetic = HF transformers' eager bf16 Llama.
 * weights[9]: 0 input_norm.w  1 q.w [h*d,H]  2 k.w [kvh*d,H]  3 v.w  4 o.w [H,h*d]  5 post_norm.w  6 gate.w [F,H]
 * 7 up.w [F,H]  8 down.w [H,F]; q|k|v and gate|up adjacent in the packed layout.  KV cache on the device, seq-major
 * [smax][batch][kv_heads][head_dim], post-RoPE keys.  cos/sin: [max_pos][head_dim] bf16 tables.
 * gu_block: 0 = gate.w and up.w as they are; 32 (and weights[7] == weights[6] + F*H, ffn % 32 == 0) = the 2F rows of the
 * gate|up block are interleaved, 32 gate rows then the matching 32 up rows -- a 64-column tile of the projection then holds
 * both factors of 32 outputs and the prefill GEMM's epilogue writes silu(gate) * up itself. */
typedef struct {
  int hidden, heads, kv_heads, ffn;
  float rms_eps;
  int gu_block;
} lia_llama_desc;
int lia_llama_pack_offsets(const lia_llama_desc* d, size_t offsets[9], size_t* total_bytes);
size_t lia_llama_workspace_bytes(const lia_llama_desc* d, int max_rows);
int lia_llama_layer_forward(lia_ctx* ctx, const lia_llama_desc* d, const void* const weights[9], const lia_bf16* x, lia_bf16* y,
                            lia_kv* kv, const lia_bf16* cos_table, const lia_bf16* sin_table, int B, int T, int pos0, int b0,
                            void* stream);
/* the Llama counterpart of lia_layer_forward_last: the LAST layer of a prefill -- norm, q | k | v projection and RoPE on every row
 * (the cache holds every position), attention / o / MLP on the last position of each row; y_last: [B, 1, H]; T > 1, pos0 == 0 */
int lia_llama_layer_forward_last(lia_ctx* ctx, const lia_llama_desc* d, const void* const weights[9], const lia_bf16* x,
                                 lia_bf16* y_last, lia_kv* kv, const lia_bf16* cos_table, const lia_bf16* sin_table, int B, int T,
                                 int pos0, int b0, void* stream);
int lia_llama_embed(const int64_t* ids, const lia_bf16* tok, lia_bf16* y, int B, int T, int H, void* stream);
int lia_llama_lm_head(lia_ctx* ctx, const lia_bf16* hidden, int B, int T, int H, const lia_bf16* normw, const lia_bf16* lm,
                      int vocab, float eps, int suppress_token, lia_bf16* logits, int64_t* next_ids, void* stream);

/* ---- decode step over a run of HBM-resident layers (build-defined fast path of the per-layer loop) ---------------------
 * The reference runs its resident layers one torch op at a time (lia/modeling_opt.py:1246-1260 -> decoder.py:172-335,
 * attentions.py:393-529: ~10 launches per layer).  These two entry points run n_layers CONSECUTIVE resident layers of one decode
 * step (T == 1, KV cache in HBM = policy 3 arithmetic) in ONE library call: lia_layer_forward(policy 3) /
 * lia_llama_layer_forward layer by layer, every layer's closing norm chained into the next (lia_ctx_chain_next_norm), the
 * hidden state alternating between y and a context-owned buffer so that x is never written.  (r04 also offered a persistent
 * one-kernel-per-layer route behind a switch: bit-identical, measured 4-9 % slower per step, retired in r05 -- LABNOTES.md.)
 *   weights: n_layers x 16 (OPT) / n_layers x 9 (Llama) device pointers, layer-major;  kv: n_layers pointers to device caches
 *   x: [B, 1, H] input (never written);  y: [B, 1, H] result;  B <= cache batch, rows [0, B) of the caches are served */
int lia_decode_layers(lia_ctx* ctx, const lia_layer_desc* d, int n_layers, const void* const* weights, const lia_bf16* x,
                      lia_bf16* y, lia_kv* const* kv, int B, int pos0, void* stream);
int lia_llama_decode_layers(lia_ctx* ctx, const lia_llama_desc* d, int n_layers, const void* const* weights, const lia_bf16* x,
                            lia_bf16* y, lia_kv* const* kv, const lia_bf16* cos_table, const lia_bf16* sin_table, int B, int pos0,
                            void* stream);
/* Per-context options and counters.  The library keeps no process-wide setting: two contexts of one process may differ.
 *   LIA_OPT_FUSE_COMBINE (default 1): the split-K combine of a decode GEMM also runs the op behind it (LayerNorm / RMSNorm of the
 *     finished row, SiLU(gate) * up, RoPE); 0: every such op is a kernel of its own -- the same device functions on the same
 *     values, so the two routes agree bit for bit (tests/test_gpu_fused_combine.py).
 *   lia_ctx_get_counter(ctx, LIA_CNT_FUSED_COMBINE + kind), kind = 1 LayerNorm, 2 RMSNorm, 3 SiLU*up, 4 RoPE: fused combines
 *     launched through this context (tests assert the route was taken); -1 for an unknown key. */
enum { LIA_OPT_FUSE_COMBINE = 1 };
enum { LIA_CNT_FUSED_COMBINE = 100 };
int lia_ctx_set_option(lia_ctx* ctx, int key, long value);
long lia_ctx_get_counter(lia_ctx* ctx, int key);

/* ---- host side of the cooperative policies -------------------------------------------------------
 * Indirect-access-KV masked MHA, csrc/cpu/aten/kernels/MaskedMultiHeadAttentionKrnl.cpp:513-842: fp32
 * scores / softmax / weighted sum over a host cache, new K/V rows written in place.  q,k,v: [B,T,h*d]
 * host; cache [smax][cache_batch][h][d] host; rows pos0.. written, 0..pos0+T-1 attended. */
int lia_host_attention(const lia_bf16* q, const lia_bf16* k, const lia_bf16* v, lia_bf16* kcache, lia_bf16* vcache,
                       lia_bf16* out, int B, int T, int pos0, int heads, int head_dim, int cache_batch, int b0,
                       int n_threads);

/* policy 1, "compute everything on CPU" (modeling_opt.py:1168): the same layer entirely on the host cores.  The
 * reference routes it through IPEX (tpp_linear_bias/_relu/_add, TPPGEMMKrnl.h:89-176,671-765,858-951 + the masked
 * MHA kernel); here AVX-512(-BF16) code in the same library.  All pointers are HOST pointers; weights row-major. */
int lia_host_layer_forward(const lia_layer_desc* d, const void* const weights[16], const lia_bf16* x, lia_bf16* y,
                           lia_bf16* kcache, lia_bf16* vcache, int smax, int cache_batch, int B, int T, int pos0, int b0,
                           int n_threads);
/* The decode step of policy 1 over n_layers CONSECUTIVE layers (the reference's loop over decoder layers with every layer on the
 * CPU, modeling_opt.py:1545-1555) in one OpenMP region: weights = n_layers x 16 host pointers, kcaches / vcaches = n_layers host
 * caches; the hidden state ping-pongs between x and y (both are written).  B * T <= 256.  Returns 0 if the result is in x,
 * 1 if it is in y, a negative LIA_ERR_* code on error. */
int lia_host_layers_forward(const lia_layer_desc* d, int n_layers, const void* const* weights, lia_bf16* x, lia_bf16* y,
                            void* const* kcaches, void* const* vcaches, int smax, int cache_batch, int B, int T, int pos0,
                            int b0, int n_threads);
int lia_host_layernorm(const lia_bf16* x, const lia_bf16* g, const lia_bf16* b, lia_bf16* y, long rows, int H, float eps,
                       int n_threads);
int lia_host_linear(const lia_bf16* x, const lia_bf16* w, const lia_bf16* bias, const lia_bf16* residual, lia_bf16* y, long M,
                    int N, int K, int relu, int n_threads);
int lia_host_has_avx512_bf16(void);
/* Per-thread scratch of the host kernels (fp32 C tiles of the linears, score rows of the attention) is refused above this many
 * bytes (0 = no limit, the default) in lia_host_* calls made BY THE CALLING THREAD from now on: the call that needed it returns
 * LIA_ERR_MEMORY -- as it does when the allocation itself fails in a worker thread -- instead of running the box out of memory.
 * Thread-local, like lia_last_error: callers on other threads are not affected. */
void lia_host_thread_scratch_limit(size_t bytes_per_thread);

/* ---- weight streamer ------------------------------------------------------------------------------
 * Replaces load_layer / layer_copy under torch.cuda.stream(load_weight_stream) + device-wide syncs
 * (modeling_opt.py:270-318,1287-1312,1508-1515): n_slots HBM slots, one pinned hipMemcpyAsync per packed
 * layer on a dedicated copy stream, event handshakes with the compute stream. */
int lia_stream_create(lia_ctx* ctx, int n_slots, size_t slot_bytes, lia_streamer** out);
void lia_stream_destroy(lia_streamer* s);
void* lia_stream_slot_ptr(lia_streamer* s, int slot);
/* enqueue host -> slot; first waits (on the copy stream) until the slot's last consumer released it.
 * pinned = 0 stages through the streamer's two pinned 64 MiB bounce buffers -- a team memcpy fills one while the copy engine drains
 * the other -- (the reference's cpu_buff path,
 * modeling_opt.py:1219-1220, 1289-1292). */
int lia_stream_prefetch(lia_streamer* s, int slot, const void* host_ptr, size_t bytes, int pinned);
/* the same in three steps, for callers that put work of their own (an RCCL broadcast of each chunk to the
 * data-parallel peers) between the host copy and the moment the slot is declared ready */
int lia_stream_begin(lia_streamer* s, int slot);
int lia_stream_copy_chunk(lia_streamer* s, int slot, size_t offset, const void* host_ptr, size_t bytes, int pinned);
int lia_stream_mark_ready(lia_streamer* s, int slot);
/* pack10: a lossless wire format for the streamed bf16 layers (lia_pack10.hip; build-defined: the reference ships raw bf16).  One
 * byte sign|mantissa per value + a three-level exponent code (2-bit level-1 planes for the three most frequent exponents of a
 * 65536-value region, a compacted 2-bit level 2 for the next three, a 4-bit level 3; two per-1024-value offset tables): 10.8 bits
 * per value for N(0,sigma) weights, against an exponent-entropy bound of 10.55.  The host keeps / ships the encoded bytes, a kernel
 * on a side stream rebuilds the exact bf16 layer in the slot.  n_values must be a multiple of 1024.  encode returns 0, or 1 when
 * the layer does not fit the format and must travel raw. */
size_t lia_pack10_bound(size_t n_values);
int lia_pack10_encode(const lia_bf16* src_device, size_t n_values, char* dst_device, size_t dst_capacity, size_t* out_bytes);
/* Host-side consistency check of an encoded buffer's 256-byte header (magic, version, value count == the slot's, every offset
 * inside `staged_bytes`, counts within their capacities) -- run when a packed layer is mapped from a checkpoint file or pinned,
 * because the decode kernel takes its loop bounds and offsets from the header.  0 = consistent, negative = the failed check. */
int lia_pack10_validate(const void* header_host, size_t staged_bytes, size_t n_values);
/* Rebuild the raw bf16 values of an encoded buffer (device -> device), asynchronous on `stream` (NULL = the default
 * stream): the kernels the streamer runs, exposed for re-tiering a layer that the host holds in a packed format
 * (model placement, lia/modeling_opt.py:229-268) and for the round-trip tests.  format: 10 (the argument is kept so that a later
 * format needs no new entry point). */
int lia_pack_decode(const char* src_device, lia_bf16* dst_device, size_t n_values, int format, void* stream);
/* format: 10 */
int lia_stream_prefetch_packed(lia_streamer* s, int slot, const void* host_ptr, size_t packed_bytes, size_t n_values, int format,
                               int pinned);
int lia_stream_copy_chunk_packed(lia_streamer* s, int slot, size_t offset, const void* host_ptr, size_t bytes, int pinned);
int lia_stream_decode_packed(lia_streamer* s, int slot, size_t n_values, int format);
void* lia_stream_staging_ptr(lia_streamer* s, int slot);
int lia_stream_wait(lia_streamer* s, int slot, void* compute_stream);    /* compute waits for the copy  */
int lia_stream_release(lia_streamer* s, int slot, void* compute_stream); /* slot reusable after this    */
/* bytes copied and copy-engine busy milliseconds since the last reset (hipEvent timing on the copy stream) */
int lia_stream_stats(lia_streamer* s, double* bytes, double* busy_ms, int reset);
/* the same totals without ever blocking: copies still in flight are left for a later call (used inside the token loop by the
 * online cooperative-split controller, which must not wait for the prefetched layers) */
int lia_stream_poll_stats(lia_streamer* s, double* bytes, double* busy_ms);
/* Live timing of the wire-format decode kernel (bench.py's roofline.dominant_kernel; the role rocprofv3's per-kernel average plays
 * in profiles/): launches, summed HIP-event milliseconds around the MAIN decode kernel on the decode stream, encoded bytes read and
 * bf16 bytes written by them since the last reset.  Waits for the decodes still in flight. */
int lia_stream_decode_stats(lia_streamer* s, long* launches, double* ms, double* bytes_in, double* bytes_out, int reset);
/* the same totals without blocking: decodes still in flight are left for a later call (counted into the window they finish in) --
 * what bench.py calls at the edge of its timed region, so that reading the counters does not drain the prefetched layers' decodes
 * right before the first timed step.  (busy_ms of lia_stream_stats includes the staging memcpy for PAGEABLE sources.) */
int lia_stream_poll_decode_stats(lia_streamer* s, long* launches, double* ms, double* bytes_in, double* bytes_out, int reset);
void* lia_stream_copy_stream(lia_streamer* s);

/* ---- host memory tiers ------------------------------------------------------------------------------
 * The four exports of the reference's libnuma shim, lia/cxl/numa_alloc.c:7,25,66,108 (same names, same
 * NULL-on-failure + stderr behaviour), with the interleave node set configurable instead of hard-coded
 * {2,3} (numa_alloc.c:80-81): lia_numa_set_interleave_nodes or env LIA_CXL_NODES="2,3". */
void* numa_alloc_node(size_t size, int node);
void* numa_alloc_interleave(size_t size);
void numa_free_node(void* memory, size_t size);
void check_memory_node(void* memory, int num);
int lia_numa_set_interleave_nodes(const int* nodes, int n);
int lia_numa_available(void);
/* make a host range DMA-able by the copy engine (hipHostRegister); the reference leaves CXL tensors
 * pageable (numa_alloc.py:49) so its copies degrade to staged synchronous ones */
int lia_numa_register(void* ptr, size_t size);
/* the same for a read-only mapping (PROT_READ, MAP_SHARED mmap of a checkpoint file): pinned without write intent, so the
 * page-cache pages are used in place */
int lia_numa_register_readonly(void* ptr, size_t size);
int lia_numa_unregister(void* ptr);
/* pinned host memory (Tensor.pin_memory(), modeling_opt.py:207-227) */
void* lia_host_alloc_pinned(size_t size);
void lia_host_free_pinned(void* p);

/* blocking copies between a host range and device memory (model placement, lia/modeling_opt.py:229-268) */
int lia_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes);
int lia_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes);

/* TPP-blocked [N/16,K/64,32,16,2] <-> row-major [N,K] on the host (_weight_prepack.py:19-63;
 * attentions.py:381-382) */
int lia_tpp_unblock(const lia_bf16* blocked, lia_bf16* plain, int N, int K);
int lia_tpp_block(const lia_bf16* plain, lia_bf16* blocked, int N, int K);

#ifdef __cplusplus
}
#endif
#endif /* LIA_HIP_H */
