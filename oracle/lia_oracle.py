"""ctypes binding + model-level driver for the CPU oracle (oracle/lia_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Parity status: pinned against outputs of the reference's own functions executed
in the build container (tests/golden/), the reference holds no vectors for this path.

Tensors are numpy uint16 arrays holding bf16 bit patterns.
"""
import ctypes
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

LAYER_TENSORS = (
    "ln1_w", "ln1_b", "q_w", "q_b", "k_w", "k_b", "v_w", "v_b",
    "out_w", "out_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
)
EPS = 1e-5


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liblia_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        vp, i, l, f = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
        L.lia_oracle_layernorm.argtypes = [vp, vp, vp, vp, l, i, f]
        L.lia_oracle_linear.argtypes = [vp, vp, vp, vp, vp, l, i, i, i, i]
        L.lia_oracle_kv_store.argtypes = [vp, vp, i, i, i, i]
        L.lia_oracle_attn_gpu.argtypes = [vp, vp, vp, vp, i, i, i, i, i, f, i]
        L.lia_oracle_attn_cpu.argtypes = [vp, vp, vp, vp, i, i, i, i, i, f, i]
        L.lia_oracle_layer_forward.argtypes = [i, vp, vp, vp, vp, vp, i, i, i, i, i, i, f]
        L.lia_oracle_embed.argtypes = [vp, vp, vp, vp, i, i, i, i]
        L.lia_oracle_lm_head.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, f]
        L.lia_oracle_set_fast.argtypes = [i]
        L.lia_oracle_set_fast.restype = None
        L.lia_oracle_set_attn_twin.argtypes = [i]
        L.lia_oracle_set_attn_twin.restype = None
        L.lia_oracle_fast_available.restype = i
        L.lia_oracle_num_threads.restype = i
        L.lia_oracle_set_threads.argtypes = [i]
        for fn in ("layernorm", "linear", "kv_store", "attn_gpu", "attn_cpu", "layer_forward", "embed", "lm_head",
                   "set_threads"):
            getattr(L, "lia_oracle_" + fn).restype = None
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint16)


def layernorm(x, g, b, eps=EPS):
    x = _c(x)
    y = np.empty_like(x)
    lib().lia_oracle_layernorm(_p(x), _p(_c(g)), _p(_c(b)), _p(y), x.size // x.shape[-1], x.shape[-1], eps)
    return y


def linear(x, w, bias=None, residual=None, relu=False, split_bias=True):
    x, w = _c(x), _c(w)
    N, K = w.shape
    M = x.size // K
    y = np.empty(x.shape[:-1] + (N,), dtype=np.uint16)
    b = None if bias is None else _c(bias)
    r = None if residual is None else _c(residual)
    lib().lia_oracle_linear(_p(x), _p(w), _p(b), _p(r), _p(y), M, N, K, int(relu), int(split_bias))
    return y


def attention(q, kc, vc, S, heads, policy_gpu=True, causal=None):
    """q [B,T,H]; kc/vc [Smax,B,h,d] with rows < S valid."""
    q = _c(q)
    B, T, H = q.shape
    d = H // heads
    out = np.empty_like(q)
    causal = (T > 1) if causal is None else causal
    if policy_gpu:
        lib().lia_oracle_attn_gpu(_p(q), _p(kc), _p(vc), _p(out), B, T, S, heads, d, float(d) ** -0.5, int(causal))
    else:
        lib().lia_oracle_attn_cpu(_p(q), _p(kc), _p(vc), _p(out), B, T, S, heads, d, float(d) ** 0.5, int(causal))
    return out


def layer_forward(policy, W, x, kc, vc, pos0, heads):
    """W: dict name -> uint16 array (row-major linears).  x [B,T,H].  kc/vc [Smax,B,h,d] updated in place."""
    x = _c(x)
    B, T, H = x.shape
    F = W["fc1_w"].shape[0]
    ws = [_c(W[n]) for n in LAYER_TENSORS]
    arr = (ctypes.c_void_p * 16)(*[w.ctypes.data for w in ws])
    y = np.empty_like(x)
    assert kc.flags.c_contiguous and vc.flags.c_contiguous and kc.dtype == np.uint16
    lib().lia_oracle_layer_forward(policy, arr, _p(x), _p(y), _p(kc), _p(vc), B, T, pos0, H, heads, F, EPS)
    return y


def embed(ids, tok, pos, past_len):
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    B, T = ids.shape
    H = tok.shape[1]
    y = np.empty((B, T, H), dtype=np.uint16)
    lib().lia_oracle_embed(_p(ids), _p(_c(tok)), _p(_c(pos)), _p(y), B, T, past_len, H)
    return y


def lm_head(hidden, lnw, lnb, emb):
    hidden = _c(hidden)
    B, T, H = hidden.shape
    vocab = emb.shape[0]
    logits = np.empty((B, vocab), dtype=np.uint16)
    nxt = np.empty((B,), dtype=np.int64)
    lib().lia_oracle_lm_head(_p(hidden), _p(_c(lnw)), _p(_c(lnb)), _p(_c(emb)), _p(logits), _p(nxt), B, T, H, vocab,
                             EPS)
    return logits, nxt


def tpp_block(w):
    """[N,K] -> [N/16,K/64,32,16,2]: the host/wire layout of every streamed Linear weight in the
    reference (intel_extension_for_pytorch/nn/utils/_weight_prepack.py:19-63; bk=16, bc=64, VNNI=2)."""
    N, K = w.shape
    return np.ascontiguousarray(w.reshape(N // 16, 16, K // 64, 32, 2).transpose(0, 2, 3, 1, 4))


def tpp_unblock(wb):
    """Inverse, as the reference does on the GPU at every use: permute([0,3,1,2,4]).view(N,K)
    (attentions.py:381-382,412; decoder.py:25-58)."""
    n16, k64, _, _, _ = wb.shape
    return np.ascontiguousarray(wb.transpose(0, 3, 1, 2, 4)).reshape(n16 * 16, k64 * 64)


def generate(model, input_ids, max_new_tokens, heads, prefill_policy=1, decoding_policy=1, gpu_percentage=0,
             return_logits=False, return_kv=False):
    """Greedy loop of the reference: greedy_search.py:144-424 over OPTDecoder.forward's layer loop
    (lia/modeling_opt.py:1222-1558): layers [0, n_gpu) run policy 3, the rest the phase's policy;
    n_gpu = int(L * gpu% / 100) (:1182).  Returns (ids [B,T+new], latency_list) like config.token_latency."""
    ids = np.ascontiguousarray(input_ids, dtype=np.int64)
    B, T = ids.shape
    L = len(model["layers"])
    H = model["embed_tokens"].shape[1]
    d = H // heads
    n_gpu = int(L * gpu_percentage / 100)
    Smax = T + max_new_tokens
    kcs = [np.zeros((Smax, B, heads, d), dtype=np.uint16) for _ in range(L)]
    vcs = [np.zeros((Smax, B, heads, d), dtype=np.uint16) for _ in range(L)]
    lat, all_logits = [], []
    past = 0
    cur = ids
    for step in range(max_new_tokens):
        tic = time.time()
        hid = embed(cur, model["embed_tokens"], model["embed_positions"], past)
        for li, W in enumerate(model["layers"]):
            pol = 3 if li < n_gpu else (prefill_policy if step == 0 else decoding_policy)
            hid = layer_forward(pol, W, hid, kcs[li], vcs[li], past, heads)
        logits, nxt = lm_head(hid, model["final_ln_w"], model["final_ln_b"], model["embed_tokens"])
        past += cur.shape[1]
        ids = np.concatenate([ids, nxt[:, None]], axis=1)
        cur = nxt[:, None]
        lat.append(time.time() - tic)
        if return_logits:
            all_logits.append(logits)
    if return_kv:          # + the per-layer caches [Smax, B, h, d] (rows [0, T + new - 1) written), for the K/V-row parity tests
        return (ids, lat, all_logits, kcs, vcs) if return_logits else (ids, lat, kcs, vcs)
    if return_logits:
        return ids, lat, all_logits
    return ids, lat


# ---------------------------------------------------------------------------------------------------------
# Llama family (config 4, build-defined): HF transformers eager Llama in bf16 is what is restated
# ---------------------------------------------------------------------------------------------------------
LLAMA_TENSORS = ("in_norm_w", "q_w", "k_w", "v_w", "o_w", "post_norm_w", "gate_w", "up_w", "down_w")
_llama_bound = False


def _llama_lib():
    global _llama_bound
    L = lib()
    if not _llama_bound:
        vp, i, l, f = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
        L.lia_oracle_rmsnorm.argtypes = [vp, vp, vp, l, i, f]
        L.lia_oracle_rope.argtypes = [vp, vp, vp, i, i, i, i, i]
        L.lia_oracle_attn_gqa.argtypes = [vp, vp, vp, vp, i, i, i, i, i, i, f]
        L.lia_oracle_silu_mul.argtypes = [vp, vp, vp, l]
        L.lia_oracle_llama_layer_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, f]
        L.lia_oracle_llama_lm_head.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, f]
        for n in ("rmsnorm", "rope", "attn_gqa", "silu_mul", "llama_layer_forward", "llama_lm_head"):
            getattr(L, "lia_oracle_" + n).restype = None
        _llama_bound = True
    return L


def rope_tables(max_pos, d, theta):
    """cos/sin [max_pos, d] as bf16 bits, computed like LlamaRotaryEmbedding.forward (fp32, then cast)."""
    import torch
    inv_freq = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
    freqs = torch.outer(torch.arange(max_pos, dtype=torch.float32), inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    to_bits = lambda t: t.to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)  # noqa: E731
    return to_bits(emb.cos()), to_bits(emb.sin())


def rmsnorm(x, w, eps=1e-5):
    x = _c(x)
    y = np.empty_like(x)
    _llama_lib().lia_oracle_rmsnorm(_p(x), _p(_c(w)), _p(y), x.size // x.shape[-1], x.shape[-1], eps)
    return y


def rope(x, cosb, sinb, heads, pos0):
    """x [B,T,heads*d] rotated in a copy."""
    x = _c(x).copy()
    B, T, HD = x.shape
    _llama_lib().lia_oracle_rope(_p(x), _p(_c(cosb)), _p(_c(sinb)), B, T, heads, HD // heads, pos0)
    return x


def attention_gqa(q, kc, vc, S, heads, kv_heads):
    q = _c(q)
    B, T, H = q.shape
    d = H // heads
    out = np.empty_like(q)
    _llama_lib().lia_oracle_attn_gqa(_p(q), _p(kc), _p(vc), _p(out), B, T, S, heads, kv_heads, d, float(d) ** -0.5)
    return out


def silu_mul(g, u):
    g, u = _c(g), _c(u)
    y = np.empty_like(g)
    _llama_lib().lia_oracle_silu_mul(_p(g), _p(u), _p(y), g.size)
    return y


def llama_layer_forward(W, x, kc, vc, cosb, sinb, pos0, heads, kv_heads, eps=1e-5):
    x = _c(x)
    B, T, H = x.shape
    F = W["gate_w"].shape[0]
    ws = [_c(W[n]) for n in LLAMA_TENSORS]
    arr = (ctypes.c_void_p * 9)(*[w.ctypes.data for w in ws])
    y = np.empty_like(x)
    _llama_lib().lia_oracle_llama_layer_forward(arr, _p(x), _p(y), _p(kc), _p(vc), _p(_c(cosb)), _p(_c(sinb)), B, T, pos0, H, heads,
                                                kv_heads, F, eps)
    return y


def llama_generate(model, input_ids, max_new_tokens, heads, kv_heads, theta, eps=1e-5, return_logits=False):
    """Greedy loop over a Llama-family model dict (synth.make_llama_model layout)."""
    ids = np.ascontiguousarray(input_ids, dtype=np.int64)
    B, T = ids.shape
    L = len(model["layers"])
    H = model["embed_tokens"].shape[1]
    d = H // heads
    vocab = model["lm_head"].shape[0]
    Smax = T + max_new_tokens
    cosb, sinb = rope_tables(Smax, d, theta)
    kcs = [np.zeros((Smax, B, kv_heads, d), dtype=np.uint16) for _ in range(L)]
    vcs = [np.zeros((Smax, B, kv_heads, d), dtype=np.uint16) for _ in range(L)]
    lat, all_logits, past, cur = [], [], 0, ids
    for step in range(max_new_tokens):
        tic = time.time()
        hid = np.ascontiguousarray(model["embed_tokens"][cur])          # no position embedding, no scaling
        for li, W in enumerate(model["layers"]):
            hid = llama_layer_forward(W, hid, kcs[li], vcs[li], cosb, sinb, past, heads, kv_heads, eps)
        logits = np.empty((B, vocab), dtype=np.uint16)
        nxt = np.empty((B,), dtype=np.int64)
        _llama_lib().lia_oracle_llama_lm_head(_p(_c(hid)), _p(_c(model["final_norm_w"])), _p(_c(model["lm_head"])), _p(logits), _p(nxt),
                                              B, hid.shape[1], H, vocab, eps)
        past += cur.shape[1]
        ids = np.concatenate([ids, nxt[:, None]], axis=1)
        cur = nxt[:, None]
        lat.append(time.time() - tic)
        all_logits.append(logits)
    return (ids, lat, all_logits) if return_logits else (ids, lat)
