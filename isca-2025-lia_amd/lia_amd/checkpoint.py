"""Checkpoint -> streaming format: HF OPT / Llama checkpoints (safetensors or .bin, plain or IPEX/TPP-blocked
linears) to the packed row-major per-layer buffers the streamer moves.

Reference counterparts: AutoModelForCausalLM.from_pretrained(torch_dtype=bf16, device_map='cpu')
(run_generation.py:159-166) followed by the TPP blocking of every Linear (optimize.py:1098,1116) and
move_gpu_layer's un-blocking (lia/modeling_opt.py:229-268); here the blocked layout, if present, is undone
ONCE on the host (lia_tpp_unblock) and never again.

The conversion STREAMS (r06): tensors are fetched one decoder layer at a time -- `safe_open` per tensor for
safetensors, shard by shard through `pytorch_model.bin.index.json` for .bin directories (the layout of the
reference's own OPT-175B dummy directory, llm/utils/opt-weight-gen.py:61-69), a memory-mapped `torch.load`
for a single .bin -- packed, and handed straight to the tier the flags ask for.  Peak host memory is one
layer (+ one .bin shard) beside the packed output; r05 held every shard in one dict and a second copy of
every tensor (2 x the checkpoint: impossible for a 350 GB OPT-175B directory on a 300 GiB box).
"""
import glob
import json
import os

import numpy as np
import torch

from . import _native as N
from .model import LiaOPTModel, OPTShape

_HF = {"ln1": "self_attn_layer_norm", "q": "self_attn.q_proj", "k": "self_attn.k_proj", "v": "self_attn.v_proj",
       "out": "self_attn.out_proj", "ln2": "final_layer_norm", "fc1": "fc1", "fc2": "fc2"}
_HF_LLAMA = {"in_norm_w": "input_layernorm", "q_w": "self_attn.q_proj", "k_w": "self_attn.k_proj", "v_w": "self_attn.v_proj",
             "o_w": "self_attn.o_proj", "post_norm_w": "post_attention_layernorm", "gate_w": "mlp.gate_proj", "up_w": "mlp.up_proj",
             "down_w": "mlp.down_proj"}


def _bits(t):
    return t.detach().to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)


def _unblock_if_needed(a, n, k):
    """A TPP-blocked [N/16,K/64,32,16,2] linear (_weight_prepack.py:19-63) comes back row-major [N,K]."""
    if a.ndim == 5:
        out = np.empty((n, k), np.uint16)
        a = np.ascontiguousarray(a)
        N.check(N.lib().lia_tpp_unblock(a.ctypes.data, out.ctypes.data, n, k), "lia_tpp_unblock")
        return out
    return a


class TensorSource:
    """name -> tensor of a HF checkpoint directory, one tensor at a time.

    safetensors: the header of every file is read once (names only), a tensor is read when asked for.
    .bin with an index: `weight_map` names the shard of every tensor; ONE shard is resident at a time (the layers of a HF
    checkpoint are shard-contiguous, so a layer-by-layer walk loads every shard once -- `shard_loads` counts them).
    a single .bin: torch.load(mmap=True), the tensors page in as they are read."""

    def __init__(self, path):
        self.path = path
        self.where = {}                   # name -> file
        self.kind = None
        self._open = {}                   # file -> safe_open handle
        self._shard = (None, None)        # (file, its state dict)
        self.shard_loads = 0
        st_files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        bin_files = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
        if st_files:
            from safetensors import safe_open
            self.kind = "safetensors"
            for f in st_files:
                h = safe_open(f, framework="pt", device="cpu")
                self._open[f] = h
                for k in h.keys():
                    self.where[k] = f
        elif bin_files:
            self.kind = "bin"
            idx = os.path.join(path, "pytorch_model.bin.index.json")
            if os.path.exists(idx):
                for k, f in json.load(open(idx))["weight_map"].items():
                    self.where[k] = os.path.join(path, f)
            else:
                for f in bin_files:        # no index: the names come from the files themselves (mapped, not read)
                    for k in self._load_bin(f):
                        self.where[k] = f
        else:
            raise FileNotFoundError(f"no *.safetensors / pytorch_model*.bin under {path}")

    def _load_bin(self, f):
        if self._shard[0] != f:
            self._shard = (None, None)     # drop the previous shard BEFORE the next one is read: never two in memory
            try:
                sd = torch.load(f, map_location="cpu", weights_only=True, mmap=True)
            except (RuntimeError, ValueError, TypeError):       # a legacy (non-zipfile) checkpoint cannot be mapped
                sd = torch.load(f, map_location="cpu", weights_only=True)
            self._shard = (f, sd)
            self.shard_loads += 1
        return self._shard[1]

    def __contains__(self, name):
        return name in self.where

    def names(self):
        return self.where.keys()

    def get(self, name):
        f = self.where[name]
        if self.kind == "safetensors":
            return self._open[f].get_tensor(name)
        return self._load_bin(f)[name]

    def close(self):
        self._open, self._shard = {}, (None, None)


def _opt_prefix(src):
    return "model.decoder." if any(k.startswith("model.decoder.") for k in src.names()) else "decoder."


def opt_head_numpy(src, pre):
    return {"embed_tokens": _bits(src.get(pre + "embed_tokens.weight")), "embed_positions": _bits(src.get(pre + "embed_positions.weight")),
            "final_ln_w": _bits(src.get(pre + "final_layer_norm.weight")), "final_ln_b": _bits(src.get(pre + "final_layer_norm.bias"))}


def opt_layer_numpy(src, pre, i, H, F):
    """layer i of a HF OPT checkpoint -> the dict of tests/golden/synth.make_layer (uint16 bf16 bits, row-major linears)"""
    dims = {"q": (H, H), "k": (H, H), "v": (H, H), "out": (H, H), "fc1": (F, H), "fc2": (H, F)}
    lw = {}
    for short, hf in _HF.items():
        w = _bits(src.get(f"{pre}layers.{i}.{hf}.weight"))
        if short in dims:
            w = _unblock_if_needed(w, *dims[short])
        lw[short + "_w"] = w
        lw[short + "_b"] = _bits(src.get(f"{pre}layers.{i}.{hf}.bias"))
    return lw


def iter_hf_opt_layers(path, cfg=None):
    """-> (TensorSource, prefix, generator of (i, layer dict)): the streaming walk load_hf_opt and the CPU memory test share"""
    cfg = cfg or json.load(open(os.path.join(path, "config.json")))
    src = TensorSource(path)
    pre = _opt_prefix(src)
    H, F, L = cfg["hidden_size"], cfg["ffn_dim"], cfg["num_hidden_layers"]
    return src, pre, ((i, opt_layer_numpy(src, pre, i, H, F)) for i in range(L))


def state_dict_to_numpy(sd, cfg):
    """HF OPT state dict (tensors, all in memory) -> the dict layout of tests/golden/synth.make_model (uint16 bf16 bits).
    Kept for callers that already hold a state dict; load_hf_opt streams instead."""
    class _Dict:
        def __init__(self, d):
            self.d = d

        def names(self):
            return self.d.keys()

        def get(self, k):
            return self.d[k]
    src = _Dict(sd)
    pre = _opt_prefix(src)
    m = opt_head_numpy(src, pre)
    m["layers"] = [opt_layer_numpy(src, pre, i, cfg["hidden_size"], cfg["ffn_dim"]) for i in range(cfg["num_hidden_layers"])]
    return m


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(torch.bfloat16).cuda()


def _place_layer(st, i, placement):
    """hand a freshly packed (pageable) layer to its tier at once, so the conversion never holds the whole model in plain host
    memory beside the tier'd copy; placement = None: leave it pageable (the first forward places it)"""
    if placement is None:
        return
    n_gpu, pin_weight, enable_cxl, wire, raw_layers = placement
    if i < n_gpu:
        st.to_device()
    elif enable_cxl and pin_weight:
        st.to_cxl(0 if i in raw_layers else wire)
    elif pin_weight:
        st.to_pinned(wire, keep_raw=(i in raw_layers))


def load_hf_opt(path, n_gpu_layers=None, pin_weight=False, enable_cxl=False, wire=0, raw_layers=()):
    """HF OPT directory -> LiaOPTModel, one layer at a time.  n_gpu_layers given: every layer goes straight to the tier the flags
    name (HBM / pinned in the wire format / the NUMA tier) and the model comes back placed; None: layers stay in plain host
    memory and the scheduler's first forward places them (LiaOPTModel.place)."""
    cfg = json.load(open(os.path.join(path, "config.json")))
    if not cfg.get("do_layer_norm_before", True) or cfg.get("word_embed_proj_dim", cfg["hidden_size"]) != cfg["hidden_size"]:
        raise ValueError("post-LN / projected-embedding OPT variants (opt-350m) are not supported")
    shape = OPTShape(os.path.basename(path.rstrip("/")), cfg["hidden_size"], cfg["num_attention_heads"], cfg["ffn_dim"],
                     cfg["num_hidden_layers"], vocab=cfg["vocab_size"], max_pos=cfg["max_position_embeddings"])
    src, pre, layers = iter_hf_opt_layers(path, cfg)
    model = LiaOPTModel(shape)
    head = opt_head_numpy(src, pre)
    model.embed_tokens, model.embed_positions = _dev(head["embed_tokens"]), _dev(head["embed_positions"])
    model.final_ln_w, model.final_ln_b = _dev(head["final_ln_w"]), _dev(head["final_ln_b"])
    del head
    from .model import LayerStore
    placement = None if n_gpu_layers is None else (int(n_gpu_layers), bool(pin_weight), bool(enable_cxl), LayerStore._fmt_of(wire), frozenset(raw_layers))
    for i, lw in layers:
        model.layers[i].set_from_numpy(lw)
        del lw
        _place_layer(model.layers[i], i, placement)
    src.close()
    if placement is not None:
        torch.cuda.synchronize()
        if pin_weight:          # (without --pin-weight the layers stay pageable: place() has nothing to move either)
            model.placed_for = model._place_key(placement[0], pin_weight, enable_cxl, placement[3], placement[4], None)
    return model


def is_llama_dir(path):
    """a HF checkpoint directory whose config.json names a Llama architecture"""
    try:
        cfg = json.load(open(os.path.join(path, "config.json")))
    except (OSError, ValueError):
        return False
    return cfg.get("model_type") == "llama" or any("Llama" in a for a in cfg.get("architectures", []))


def llama_shape_of(path):
    from .llama import LlamaShape
    cfg = json.load(open(os.path.join(path, "config.json")))
    heads = cfg["num_attention_heads"]
    if cfg.get("head_dim") not in (None, cfg["hidden_size"] // heads):
        raise ValueError("Llama variants whose head_dim is not hidden_size / num_attention_heads are not supported")
    if cfg.get("rope_scaling") not in (None, {}):
        # (Llama-3.1's scaled frequencies are a different table, not this kernel's plain theta ** (-2i / d))
        raise ValueError(f"rope_scaling = {cfg['rope_scaling']!r} is not supported (plain rotary tables only)")
    return LlamaShape(os.path.basename(path.rstrip("/")), cfg["hidden_size"], heads, cfg.get("num_key_value_heads", heads), cfg["intermediate_size"],
                      cfg["num_hidden_layers"], cfg["vocab_size"], max_pos=cfg.get("max_position_embeddings", 8192),
                      rope_theta=float(cfg.get("rope_theta", 10000.0)), rms_eps=float(cfg.get("rms_norm_eps", 1e-5)))


def llama_layer_numpy(src, i):
    return {short: _bits(src.get(f"model.layers.{i}.{hf}.weight")) for short, hf in _HF_LLAMA.items()}


def load_hf_llama(path, n_gpu_layers=None, pin_weight=True, wire=0):
    """HF Llama directory (LlamaForCausalLM: run_generation.py:159-166 loads it through the same AutoModel call) -> LiaLlamaModel,
    one layer at a time like load_hf_opt.  lm_head.weight is the embedding when the checkpoint ties them."""
    from .llama import LiaLlamaModel
    from .model import LayerStore
    shape = llama_shape_of(path)
    src = TensorSource(path)
    model = LiaLlamaModel(shape)
    model.embed_tokens = _dev(_bits(src.get("model.embed_tokens.weight")))
    model.lm_head = _dev(_bits(src.get("lm_head.weight"))) if "lm_head.weight" in src else model.embed_tokens
    model.final_norm_w = _dev(_bits(src.get("model.norm.weight")))
    fmt = LayerStore._fmt_of(wire)
    for i, st in enumerate(model.layers):
        model._pack_numpy(st, llama_layer_numpy(src, i))
        if n_gpu_layers is not None:
            if i < n_gpu_layers:
                st.to_device()
            elif pin_weight:
                st.to_pinned(fmt)
    src.close()
    if n_gpu_layers is not None and pin_weight:
        torch.cuda.synchronize()
        model.placed_for = (int(n_gpu_layers), True, False, fmt)
    return model
