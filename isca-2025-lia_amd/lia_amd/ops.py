"""torch-tensor front end of the C ABI: the GPU sub-layer ops of decoder.py / attentions.py.

torch provides device memory and stream handles only; all arithmetic happens in liblia_hip.so.
Every function takes contiguous bf16 CUDA tensors (int64 for ids) and launches on `stream`
(a raw hipStream_t as int; default = the context's compute stream).
"""
import ctypes

import torch

from . import _native as N

LAYER_TENSORS = (
    "ln1_w", "ln1_b", "q_w", "q_b", "k_w", "k_b", "v_w", "v_b",
    "out_w", "out_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk(t, dtype=torch.bfloat16):
    if t is None:
        return
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"expected a contiguous {dtype} CUDA tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")


class Context:
    """lia_ctx: compute stream + D2H stream + per-layer workspace (include/lia_hip.h)."""

    def __init__(self, device=0, workspace_bytes=0):
        self.lib = N.lib()
        h = ctypes.c_void_p()
        N.check(self.lib.lia_ctx_create(device, workspace_bytes, ctypes.byref(h)), "lia_ctx_create")
        self.handle = h
        self.device = device
        self.workspace_bytes = workspace_bytes
        self.stream = self.lib.lia_ctx_compute_stream(h)

    def close(self):
        if self.handle:
            self.lib.lia_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        N.check(self.lib.lia_ctx_synchronize(self.handle), "lia_ctx_synchronize")

    def kv_store_wait(self):
        N.check(self.lib.lia_ctx_kv_store_wait(self.handle), "lia_ctx_kv_store_wait")

    def set_host_threads(self, n):
        N.check(self.lib.lia_ctx_set_host_threads(self.handle, n))

    def set_option(self, key, value):
        """lia_ctx_set_option: a switch of THIS context (N.LIA_OPT_*); other contexts of the process are not affected"""
        N.check(self.lib.lia_ctx_set_option(self.handle, key, value), "lia_ctx_set_option")

    def fused_combines(self, kind):
        """fused split-K combines of LIA_POST_* kind (1 LayerNorm, 2 RMSNorm, 3 SiLU*up, 4 RoPE) launched through this context"""
        return self.lib.lia_ctx_get_counter(self.handle, N.LIA_CNT_FUSED_COMBINE + kind)

    def chain_next_norm(self, g_ptr, b_ptr=None):
        """lia_ctx_chain_next_norm: the next layer call also computes the first norm (weights g, b) of the layer after it."""
        N.check(self.lib.lia_ctx_chain_next_norm(self.handle, ctypes.c_void_p(g_ptr), ctypes.c_void_p(b_ptr) if b_ptr else None),
                "lia_ctx_chain_next_norm")

    def prof_start(self, max_launches=16384, stride=1):
        """HIP-event brackets around every stride-th GEMM launch from here on (lia_prof_start / lia_prof_set_stride)."""
        N.check(self.lib.lia_prof_set_stride(self.handle, stride), "lia_prof_set_stride")
        N.check(self.lib.lia_prof_start(self.handle, max_launches), "lia_prof_start")

    def prof_stop(self):
        r = N.ProfResult()
        N.check(self.lib.lia_prof_stop(self.handle, ctypes.byref(r)), "lia_prof_stop")
        return {k: getattr(r, k) for k, _ in r._fields_}

    def _st(self, stream):
        return ctypes.c_void_p(self.stream if stream is None else stream)

    # ---- sub-layer ops --------------------------------------------------------------------------
    def layernorm(self, x, g, b, eps=1e-5, stream=None):
        for t in (x, g, b):
            _chk(t)
        y = torch.empty_like(x)
        H = x.shape[-1]
        rows = x.numel() // H
        N.check(self.lib.lia_layernorm(_ptr(x), H, _ptr(g), _ptr(b), _ptr(y), H, rows, H, eps, self._st(stream)), "lia_layernorm")
        return y

    def linear(self, x, w, bias=None, residual=None, relu=False, split_k=0, stream=None):
        for t in (x, w, bias, residual):
            _chk(t)
        Nn, K = w.shape
        M = x.numel() // K
        y = torch.empty(x.shape[:-1] + (Nn,), dtype=torch.bfloat16, device=x.device)
        N.check(self.lib.lia_linear(self.handle, _ptr(x), K, _ptr(w), _ptr(bias), _ptr(residual), Nn, _ptr(y), Nn, M, Nn, K,
                                    int(relu), split_k, self._st(stream)), "lia_linear")
        return y

    def qkv_project(self, x, w, bias, kcache, vcache, b0, pos0, stream=None):
        for t in (x, w, bias, kcache, vcache):
            _chk(t)
        B, T, H = x.shape
        q = torch.empty_like(x)
        N.check(self.lib.lia_qkv_project(self.handle, _ptr(x), _ptr(w), _ptr(bias), _ptr(q), _ptr(kcache), _ptr(vcache), B, T, H,
                                         kcache.shape[1], b0, pos0, self._st(stream)), "lia_qkv_project")
        return q

    def attention(self, q, kcache, vcache, S, heads, b0=0, stream=None):
        for t in (q, kcache, vcache):
            _chk(t)
        B, T, H = q.shape
        out = torch.empty_like(q)
        N.check(self.lib.lia_attention(_ptr(q), H, _ptr(kcache), _ptr(vcache), _ptr(out), H, B, T, S, heads, H // heads,
                                       kcache.shape[1], b0, self._st(stream)), "lia_attention")
        return out

    def embed(self, ids, tok, pos, past_len, stream=None):
        _chk(ids, torch.int64)
        _chk(tok)
        _chk(pos)
        B, T = ids.shape
        H = tok.shape[1]
        y = torch.empty((B, T, H), dtype=torch.bfloat16, device=tok.device)
        N.check(self.lib.lia_embed(_ptr(ids), _ptr(tok), _ptr(pos), _ptr(y), B, T, past_len, H, self._st(stream)), "lia_embed")
        return y

    LM_HEAD_ROWS = 256       # rows per lia_lm_head call (its argmax scratch is sized for that many)

    def lm_head(self, hidden, lnw, lnb, emb, eps=1e-5, suppress_token=-1, stream=None):
        """final LN + tied lm_head on the LAST position + argmax (models.py:424-431, greedy_search.py:367).  A batch beyond 256 rows
        (the reference's --batch-size 900 ... 1580 lines, llm/scripts/cxl_offloading.sh:13-39) goes through in chunks of 256 rows:
        rows are independent, the results are those of one call."""
        for t in (hidden, lnw, lnb, emb):
            _chk(t)
        B, T, H = hidden.shape
        vocab = emb.shape[0]
        logits = torch.empty((B, vocab), dtype=torch.bfloat16, device=hidden.device)
        nxt = torch.empty((B,), dtype=torch.int64, device=hidden.device)
        for b0 in range(0, B, self.LM_HEAD_ROWS):
            nb = min(self.LM_HEAD_ROWS, B - b0)
            N.check(self.lib.lia_lm_head(self.handle, _ptr(hidden[b0:b0 + nb]), nb, T, H, _ptr(lnw), _ptr(lnb), _ptr(emb), vocab, eps, suppress_token,
                                         _ptr(logits[b0:b0 + nb]), _ptr(nxt[b0:b0 + nb]), self._st(stream)), "lia_lm_head")
        return logits, nxt

    # ---- the operator boundary ------------------------------------------------------------------
    def layer_forward(self, desc, policy, weight_ptrs, x, y, kv, B, T, pos0, b0=0, stream=None):
        """weight_ptrs: (c_void_p * 16) of device pointers; kv: _native.KV.  x, y: [B,T,H] device."""
        N.check(self.lib.lia_layer_forward(self.handle, ctypes.byref(desc), policy, ctypes.byref(weight_ptrs), _ptr(x), _ptr(y),
                                           ctypes.byref(kv), B, T, pos0, b0, self._st(stream)), "lia_layer_forward")


    def layer_forward_last(self, desc, policy, weight_ptrs, x, y_last, kv, B, T, pos0, b0=0, stream=None):
        """lia_layer_forward_last: the last layer of a prefill -- K/V of every position, the rest on the last position only;
        y_last: [B,1,H] device."""
        N.check(self.lib.lia_layer_forward_last(self.handle, ctypes.byref(desc), policy, ctypes.byref(weight_ptrs), _ptr(x), _ptr(y_last),
                                                ctypes.byref(kv), B, T, pos0, b0, self._st(stream)), "lia_layer_forward_last")


def make_desc(hidden, heads, ffn, eps=1e-5):
    return N.LayerDesc(hidden, heads, ffn, eps)


def pack_offsets(desc):
    offs = (ctypes.c_size_t * 16)()
    total = ctypes.c_size_t()
    N.check(N.lib().lia_layer_pack_offsets(ctypes.byref(desc), ctypes.byref(offs), ctypes.byref(total)), "lia_layer_pack_offsets")
    return list(offs), total.value


def workspace_bytes(desc, max_rows):
    return N.lib().lia_layer_workspace_bytes(ctypes.byref(desc), max_rows)


def weight_ptr_array(base_ptr, offsets):
    return (ctypes.c_void_p * 16)(*[base_ptr + o for o in offsets])
