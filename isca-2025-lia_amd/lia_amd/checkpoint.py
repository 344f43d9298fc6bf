"""Checkpoint -> streaming format: HF OPT checkpoints (safetensors or .bin, plain or IPEX/TPP-blocked
linears) to the packed row-major per-layer buffers the streamer moves.

Reference counterparts: AutoModelForCausalLM.from_pretrained(torch_dtype=bf16, device_map='cpu')
(run_generation.py:159-166) followed by the TPP blocking of every Linear (optimize.py:1098,1116) and
move_gpu_layer's un-blocking (lia/modeling_opt.py:229-268); here the blocked layout, if present, is undone
ONCE on the host (lia_tpp_unblock) and never again.
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


def state_dict_to_numpy(sd, cfg):
    """HF OPT state dict (tensors) -> the dict layout of tests/golden/synth.make_model (uint16 bf16 bits)."""
    H, F, L = cfg["hidden_size"], cfg["ffn_dim"], cfg["num_hidden_layers"]
    pre = "model.decoder." if any(k.startswith("model.decoder.") for k in sd) else "decoder."
    m = {"embed_tokens": _bits(sd[pre + "embed_tokens.weight"]), "embed_positions": _bits(sd[pre + "embed_positions.weight"]),
         "final_ln_w": _bits(sd[pre + "final_layer_norm.weight"]), "final_ln_b": _bits(sd[pre + "final_layer_norm.bias"]),
         "layers": []}
    dims = {"q": (H, H), "k": (H, H), "v": (H, H), "out": (H, H), "fc1": (F, H), "fc2": (H, F)}
    for i in range(L):
        lw = {}
        for short, hf in _HF.items():
            w = _bits(sd[f"{pre}layers.{i}.{hf}.weight"])
            if short in dims:
                w = _unblock_if_needed(w, *dims[short])
            lw[short + "_w"] = w
            lw[short + "_b"] = _bits(sd[f"{pre}layers.{i}.{hf}.bias"])
        m["layers"].append(lw)
    return m


def load_hf_opt(path):
    cfg = json.load(open(os.path.join(path, "config.json")))
    if not cfg.get("do_layer_norm_before", True) or cfg.get("word_embed_proj_dim", cfg["hidden_size"]) != cfg["hidden_size"]:
        raise ValueError("post-LN / projected-embedding OPT variants (opt-350m) are not supported")
    sd = {}
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if files:
        from safetensors.torch import load_file
        for f in files:
            sd.update(load_file(f))
    else:
        for f in sorted(glob.glob(os.path.join(path, "pytorch_model*.bin"))):
            sd.update(torch.load(f, map_location="cpu", weights_only=True))
    if not sd:
        raise FileNotFoundError(f"no *.safetensors / pytorch_model*.bin under {path}")
    shape = OPTShape(os.path.basename(path.rstrip("/")), cfg["hidden_size"], cfg["num_attention_heads"], cfg["ffn_dim"],
                     cfg["num_hidden_layers"], vocab=cfg["vocab_size"], max_pos=cfg["max_position_embeddings"])
    return LiaOPTModel.from_numpy(shape, state_dict_to_numpy(sd, cfg))
