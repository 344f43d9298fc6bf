"""ctypes binding of liblia_hip.so (include/lia_hip.h).

The library is the product; this module only marshals pointers.  There is NO fallback: if the shared
object is missing or a call fails, the error propagates (a GPU box that cannot load the HIP extension
must fail loudly, never compute on the CPU instead).

C return codes are re-raised as the exception types the reference raises at the same places:
LIA_ERR_INVALID -> ValueError (attentions.py:503,516,532), LIA_ERR_MEMORY -> MemoryError
(lia/modeling_opt.py:175), LIA_ERR_MISSING -> AttributeError (lia/modeling_opt.py:110,126).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "liblia_hip.so")

LIA_OK, LIA_ERR_INVALID, LIA_ERR_MEMORY, LIA_ERR_HIP, LIA_ERR_MISSING = 0, -1, -2, -3, -4
LIA_OPT_FUSE_COMBINE = 1           # lia_ctx_set_option keys (include/lia_hip.h)
LIA_CNT_FUSED_COMBINE = 100        # lia_ctx_get_counter(ctx, LIA_CNT_FUSED_COMBINE + kind), kind = 1 LayerNorm, 2 RMSNorm, 3 SiLU*up, 4 RoPE

c_void_p, c_int, c_long, c_size_t, c_float, c_double = (ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_size_t,
                                                         ctypes.c_float, ctypes.c_double)


class LayerDesc(ctypes.Structure):
    _fields_ = [("hidden", c_int), ("heads", c_int), ("ffn", c_int), ("ln_eps", c_float)]


class ProfResult(ctypes.Structure):
    _fields_ = [("skinny_launches", c_long), ("skinny_ms", c_double), ("skinny_bytes", c_double), ("skinny_flops", c_double),
                ("tiled_launches", c_long), ("tiled_ms", c_double), ("tiled_bytes", c_double), ("tiled_flops", c_double),
                ("empty_bracket_ms", c_double), ("host_attention_calls", c_long), ("host_attention_ms", c_double)]


class LlamaDesc(ctypes.Structure):
    _fields_ = [("hidden", c_int), ("heads", c_int), ("kv_heads", c_int), ("ffn", c_int), ("rms_eps", c_float), ("gu_block", c_int)]


class KV(ctypes.Structure):
    _fields_ = [("k", c_void_p), ("v", c_void_p), ("smax", c_int), ("batch", c_int), ("on_device", c_int)]


# name -> (restype, argtypes).  Every symbol include/lia_hip.h declares is listed here; the CPU test
# suite checks the shared object exports all of them.
SIGNATURES = {
    "lia_last_error": (ctypes.c_char_p, []),
    "lia_version": (ctypes.c_char_p, []),
    "lia_ctx_create": (c_int, [c_int, c_size_t, ctypes.POINTER(c_void_p)]),
    "lia_ctx_destroy": (None, [c_void_p]),
    "lia_ctx_compute_stream": (c_void_p, [c_void_p]),
    "lia_ctx_synchronize": (c_int, [c_void_p]),
    "lia_ctx_synchronize_compute": (c_int, [c_void_p]),
    "lia_ctx_set_host_threads": (c_int, [c_void_p, c_int]),
    "lia_ctx_serialized": (c_int, [c_void_p]),
    "lia_ctx_chain_next_norm": (c_int, [c_void_p, c_void_p, c_void_p]),
    "lia_prof_start": (c_int, [c_void_p, c_int]),
    "lia_prof_set_stride": (c_int, [c_void_p, c_int]),
    "lia_prof_stop": (c_int, [c_void_p, ctypes.POINTER(ProfResult)]),
    "lia_layer_pack_offsets": (c_int, [ctypes.POINTER(LayerDesc), ctypes.POINTER(c_size_t * 16), ctypes.POINTER(c_size_t)]),
    "lia_layer_workspace_bytes": (c_size_t, [ctypes.POINTER(LayerDesc), c_int]),
    "lia_layer_forward": (c_int, [c_void_p, ctypes.POINTER(LayerDesc), c_int, ctypes.POINTER(c_void_p * 16), c_void_p,
                                  c_void_p, ctypes.POINTER(KV), c_int, c_int, c_int, c_int, c_void_p]),
    "lia_layer_forward_last": (c_int, [c_void_p, ctypes.POINTER(LayerDesc), c_int, ctypes.POINTER(c_void_p * 16), c_void_p,
                                       c_void_p, ctypes.POINTER(KV), c_int, c_int, c_int, c_int, c_void_p]),
    "lia_ctx_kv_store_wait": (c_int, [c_void_p]),
    "lia_kv_deliver": (c_int, [c_void_p, ctypes.POINTER(KV), ctypes.POINTER(KV), c_int, c_int, ctypes.POINTER(c_int)]),
    "lia_kv_deliver_wait": (c_int, [c_void_p, c_int]),
    "lia_kv_deliver_batch_ms": (c_int, [c_void_p, ctypes.POINTER(c_double)]),
    "lia_layernorm": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long, c_long, c_int, c_float, c_void_p]),
    "lia_linear": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_long, c_int,
                           c_int, c_int, c_int, c_int, c_void_p]),
    "lia_qkv_project": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                c_int, c_int, c_int, c_int, c_void_p]),
    "lia_attention": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int, c_int,
                              c_int, c_int, c_void_p]),
    "lia_embed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "lia_lm_head": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_float,
                            c_int, c_void_p, c_void_p, c_void_p]),
    "lia_host_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_int, c_int, c_int, c_int, c_int]),
    "lia_llama_pack_offsets": (c_int, [ctypes.POINTER(LlamaDesc), ctypes.POINTER(c_size_t * 9), ctypes.POINTER(c_size_t)]),
    "lia_llama_workspace_bytes": (c_size_t, [ctypes.POINTER(LlamaDesc), c_int]),
    "lia_llama_layer_forward": (c_int, [c_void_p, ctypes.POINTER(LlamaDesc), ctypes.POINTER(c_void_p * 9), c_void_p, c_void_p,
                                        ctypes.POINTER(KV), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "lia_llama_layer_forward_last": (c_int, [c_void_p, ctypes.POINTER(LlamaDesc), ctypes.POINTER(c_void_p * 9), c_void_p, c_void_p,
                                             ctypes.POINTER(KV), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "lia_llama_embed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "lia_llama_lm_head": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_int, c_void_p,
                                  c_void_p, c_void_p]),
    "lia_decode_layers": (c_int, [c_void_p, ctypes.POINTER(LayerDesc), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "lia_llama_decode_layers": (c_int, [c_void_p, ctypes.POINTER(LlamaDesc), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int, c_void_p]),
    "lia_ctx_set_option": (c_int, [c_void_p, c_int, c_long]),
    "lia_ctx_get_counter": (c_long, [c_void_p, c_int]),
    "lia_host_layer_forward": (c_int, [ctypes.POINTER(LayerDesc), ctypes.POINTER(c_void_p * 16), c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "lia_host_layers_forward": (c_int, [ctypes.POINTER(LayerDesc), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "lia_host_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_float, c_int]),
    "lia_host_thread_scratch_limit": (None, [ctypes.c_size_t]),
    "lia_host_linear": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int]),
    "lia_host_has_avx512_bf16": (c_int, []),
    "lia_stream_create": (c_int, [c_void_p, c_int, c_size_t, ctypes.POINTER(c_void_p)]),
    "lia_stream_destroy": (None, [c_void_p]),
    "lia_stream_slot_ptr": (c_void_p, [c_void_p, c_int]),
    "lia_stream_prefetch": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_int]),
    "lia_stream_begin": (c_int, [c_void_p, c_int]),
    "lia_stream_copy_chunk": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_size_t, c_int]),
    "lia_stream_mark_ready": (c_int, [c_void_p, c_int]),
    "lia_blit": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "lia_pack10_bound": (c_size_t, [c_size_t]),
    "lia_pack10_encode": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, ctypes.POINTER(c_size_t)]),
    "lia_pack_decode": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "lia_stream_prefetch_packed": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_size_t, c_int, c_int]),
    "lia_stream_copy_chunk_packed": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_size_t, c_int]),
    "lia_stream_decode_packed": (c_int, [c_void_p, c_int, c_size_t, c_int]),
    "lia_stream_staging_ptr": (c_void_p, [c_void_p, c_int]),
    "lia_stream_wait": (c_int, [c_void_p, c_int, c_void_p]),
    "lia_stream_release": (c_int, [c_void_p, c_int, c_void_p]),
    "lia_stream_stats": (c_int, [c_void_p, ctypes.POINTER(c_double), ctypes.POINTER(c_double), c_int]),
    "lia_stream_poll_stats": (c_int, [c_void_p, ctypes.POINTER(c_double), ctypes.POINTER(c_double)]),
    "lia_stream_decode_stats": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(c_double), ctypes.POINTER(c_double),
                                        ctypes.POINTER(c_double), c_int]),
    "lia_stream_poll_decode_stats": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(c_double), ctypes.POINTER(c_double),
                                             ctypes.POINTER(c_double), c_int]),
    "lia_pack10_validate": (c_int, [c_void_p, c_size_t, c_size_t]),
    "lia_stream_copy_stream": (c_void_p, [c_void_p]),
    "numa_alloc_node": (c_void_p, [c_size_t, c_int]),
    "numa_alloc_interleave": (c_void_p, [c_size_t]),
    "numa_free_node": (None, [c_void_p, c_size_t]),
    "check_memory_node": (None, [c_void_p, c_int]),
    "lia_numa_set_interleave_nodes": (c_int, [ctypes.POINTER(c_int), c_int]),
    "lia_numa_available": (c_int, []),
    "lia_numa_register": (c_int, [c_void_p, c_size_t]),
    "lia_numa_register_readonly": (c_int, [c_void_p, c_size_t]),
    "lia_numa_unregister": (c_int, [c_void_p]),
    "lia_host_alloc_pinned": (c_void_p, [c_size_t]),
    "lia_host_free_pinned": (None, [c_void_p]),
    "lia_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lia_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lia_tpp_unblock": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "lia_tpp_block": (c_int, [c_void_p, c_void_p, c_int, c_int]),
}

_lib = None


def lib():
    """Load liblia_hip.so (after torch, so that both share one HIP runtime: torch bundles a
    libamdhip64.so.7 and the loader resolves our NEEDED libamdhip64.so.7 to the copy already mapped)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C isca-2025-lia_amd/csrc` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`.  There is no CPU fallback.")
        import torch  # noqa: F401  (maps torch's HIP runtime first)
        L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class LiaHipError(RuntimeError):
    pass


def check(rc, what=""):
    if rc == LIA_OK:
        return
    msg = lib().lia_last_error().decode(errors="replace")
    text = f"{what}: {msg}" if what else msg
    if rc == LIA_ERR_INVALID:
        raise ValueError(text)
    if rc == LIA_ERR_MEMORY:
        raise MemoryError(text)
    if rc == LIA_ERR_MISSING:
        raise AttributeError(text)
    raise LiaHipError(text)
