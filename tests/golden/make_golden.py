#!/usr/bin/env python3
"""Generate golden vectors by EXECUTING the reference's own per-layer functions on CPU.

Runs only in the build container (needs /root/reference); the GPU box never sees it. What is
committed is this script plus the small .npz fixtures it writes -- inputs are re-derived from
seeds (tests/golden/synth.py), outputs are the reference's.

Reference entry points executed (not restated):
  * OPTDecoderLayer_forward   intel_extension_for_pytorch/transformers/models/reference/modules/decoder.py:172
  * _OPTAttention_forward     .../reference/modules/attentions.py:312
  * _IPEXScaleDotProductRef   .../reference/fusions/mha_fusion.py:532-566 (OPT branch; the pure-torch
                              semantic twin of the C++ masked-MHA kernel used by policy 1/2)
  * _IPEXlinearAddRef / _IPEXlinearReluRef   .../reference/fusions/linear_fusion.py:17-24,73-80 over nn.Linear,
                              and nn.LayerNorm: the modules the CPU branch of OPTDecoderLayer_forward (policy 1,
                              decoder.py:207,231,276,287,312; attentions.py:363-374,402-408,421-440) calls
  * OPTLearnedPositionalEmbedding  lia/modeling_opt.py:357-378
  * TPP blocked layout        intel_extension_for_pytorch/nn/utils/_weight_prepack.py:19-63 (restated
                              as a 3-line view/permute; the reference's own inverse,
                              permute([0,3,1,2,4]).view(N,K) at attentions.py:381, is what is executed)

The reference functions address the GPU as device 'cuda'; there is none here, so a
TorchFunctionMode rewrites device='cuda' / .to('cuda') to CPU. Arithmetic is therefore torch-CPU
(fp32 accumulate, bf16 round after every op) -- the same rounding points as the CUDA ops, not the
same summation order. SURVEY.md section 8(c) documents this recipe.
"""
import importlib
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402

REF = "/root/reference"
IPEX = os.path.join(REF, "intel_extension_for_pytorch")


def _stub_pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    sys.modules[name] = m
    return m


def import_reference():
    import transformers  # noqa: F401  (must precede the deepspeed stub)

    ds = _stub_pkg("deepspeed")
    dsc = _stub_pkg("deepspeed.comm")
    ds.comm = dsc
    _stub_pkg("intel_extension_for_pytorch", IPEX)
    nn_pkg = _stub_pkg("intel_extension_for_pytorch.nn")
    nn_mod = _stub_pkg("intel_extension_for_pytorch.nn.modules")
    nn_mod.WeightOnlyQuantizedLinear = type("WeightOnlyQuantizedLinear", (torch.nn.Module,), {})
    nn_pkg.modules = nn_mod
    for sub in ("utils", "transformers", "transformers.models", "transformers.models.reference",
                "transformers.models.reference.modules", "transformers.models.reference.fusions"):
        _stub_pkg("intel_extension_for_pytorch." + sub, os.path.join(IPEX, *sub.split(".")))
    dec = importlib.import_module("intel_extension_for_pytorch.transformers.models.reference.modules.decoder")
    att = importlib.import_module("intel_extension_for_pytorch.transformers.models.reference.modules.attentions")
    mha = importlib.import_module("intel_extension_for_pytorch.transformers.models.reference.fusions.mha_fusion")
    mha.linear_fusion = importlib.import_module("intel_extension_for_pytorch.transformers.models.reference.fusions.linear_fusion")
    return dec, att, mha


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if kwargs.get("device") is not None and str(kwargs["device"]).startswith("cuda"):
            kwargs["device"] = "cpu"
        args = tuple("cpu" if (isinstance(a, str) and a.startswith("cuda")) else a for a in args)
        if func is torch.Tensor.view:
            # _IPEXScaleDotProductRef's OPT branch views the PERMUTED q/k/v as [B*h, T, d] (mha_fusion.py:533-537), which
            # torch refuses for T > 1 ("Use .reshape(...) instead"): same elements in the same order, one copy more
            try:
                return func(*args, **kwargs)
            except RuntimeError:
                return args[0].reshape(*args[1:])
        return func(*args, **kwargs)


def bits_to_torch(b):
    return torch.from_numpy(np.ascontiguousarray(b).view(np.int16)).view(torch.bfloat16)


def torch_to_bits(t):
    return t.detach().contiguous().to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16).copy()


def tpp_block(w):
    """[N,K] -> [N/16, K/64, 32, 16, 2]   (_weight_prepack.py:19-63: bk=16, bc=64, VNNI=2)."""
    N, K = w.shape
    return w.view(N // 16, 16, K // 64, 32, 2).permute(0, 2, 3, 1, 4).contiguous()


class _Mod:
    pass


def fake_layer(dec, att, mha, W, H, heads, policy):
    """A stand-in `self` carrying exactly the attributes the two reference functions read."""
    d = H // heads
    attn = _Mod()
    attn.num_heads, attn.head_dim, attn.embed_dim = heads, d, H
    attn.scaling = d ** -0.5
    attn.is_decoder = True
    sdp_src = _Mod()
    sdp_src.__class__ = type("OPTAttention", (), {})
    cfg = _Mod()
    cfg.architectures = ["OPTForCausalLM"]
    sdp = mha._IPEXScaleDotProductRef.__new__(mha._IPEXScaleDotProductRef)
    torch.nn.Module.__init__(sdp)
    sdp.model_backbone = "OPTForCausalLM"
    sdp.num_heads, sdp.head_dim = heads, d
    attn._IPEXScaleDotProduct = sdp
    t = {k: bits_to_torch(v) for k, v in W.items()}

    def lin(w, b):
        m = _Mod()
        m.weight, m.bias = w, b
        return m

    attn.q_proj, attn.k_proj, attn.v_proj = lin(t["q_w"], t["q_b"]), lin(t["k_w"], t["k_b"]), lin(t["v_w"], t["v_b"])

    layer = _Mod()
    layer.distributed = False
    layer.do_layer_norm_before = True
    ln1, ln2 = _Mod(), _Mod()
    ln1.normalized_shape, ln1.eps, ln1.weight, ln1.bias = (H,), 1e-5, t["ln1_w"], t["ln1_b"]
    ln2.normalized_shape, ln2.eps, ln2.weight, ln2.bias = (H,), 1e-5, t["ln2_w"], t["ln2_b"]
    layer.self_attn_layer_norm, layer.final_layer_norm = ln1, ln2
    layer.mha_linear_add = lin(t["out_w"], t["out_b"])
    lr = _Mod()
    lr.linear = lin(t["fc1_w"], t["fc1_b"])
    layer.linear_relu = lr
    layer.mlp_linear_add = lin(t["fc2_w"], t["fc2_b"])
    layer.self_attn = lambda **kw: att._OPTAttention_forward(attn, **kw)
    gpu_layer = None
    if policy == 1:
        # the CPU branch calls MODULES: nn.LayerNorm, nn.Linear and the reference's own pure-torch fusion wrappers
        lf = mha.linear_fusion

        def nn_lin(w, b):
            m = torch.nn.Linear(w.shape[1], w.shape[0], bias=True, dtype=torch.bfloat16)
            m.weight = torch.nn.Parameter(w.clone(), requires_grad=False)
            m.bias = torch.nn.Parameter(b.clone(), requires_grad=False)
            return m

        def nn_ln(w, b):
            m = torch.nn.LayerNorm(H, eps=1e-5, dtype=torch.bfloat16)
            m.weight = torch.nn.Parameter(w.clone(), requires_grad=False)
            m.bias = torch.nn.Parameter(b.clone(), requires_grad=False)
            return m

        attn.q_proj, attn.k_proj, attn.v_proj = nn_lin(t["q_w"], t["q_b"]), nn_lin(t["k_w"], t["k_b"]), nn_lin(t["v_w"], t["v_b"])
        layer.self_attn_layer_norm, layer.final_layer_norm = nn_ln(t["ln1_w"], t["ln1_b"]), nn_ln(t["ln2_w"], t["ln2_b"])
        layer.mha_linear_add = lf._IPEXlinearAddRef(nn_lin(t["out_w"], t["out_b"]))
        layer.linear_relu = lf._IPEXlinearReluRef(nn_lin(t["fc1_w"], t["fc1_b"]))
        layer.mlp_linear_add = lf._IPEXlinearAddRef(nn_lin(t["fc2_w"], t["fc2_b"]))
        return layer, None
    if policy != 3:
        gpu_layer = [t[n] if not n.endswith("_w") or n.startswith("ln") else tpp_block(t[n])
                     for n in synth.LAYER_TENSORS]
    return layer, gpu_layer


def run_layer_case(dec, att, mha, name, H, heads, F, B, T, new, seed, w_std, identical_rows, keep_existing=False):
    """policy 0 prefill (blocked streamed weights) + policy 3 prefill and `new` decode steps + policy 2 decode + policy 1
    prefill and decode steps.  keep_existing: vectors already in the fixture file are kept, only new keys are added."""
    W = synth.make_layer(seed, H, F, w_std)
    x = bits_to_torch(synth.make_hidden(seed + 1, B, T, H, identical_rows))
    out = {"cfg": np.array([H, heads, F, B, T, new, seed, int(identical_rows)], dtype=np.int64),
           "w_std": np.array([w_std], dtype=np.float64)}
    boot = (torch.zeros(1, 0, 0, 1, dtype=torch.long), torch.zeros(1, 1, 1, 1), torch.zeros(1, 1, 1, 1),
            torch.zeros(2048, B, dtype=torch.long))
    mask = torch.zeros(B, 1, T, T)  # only "is not None" is consulted (attentions.py:444)
    with CudaToCpu(), torch.no_grad():
        layer, gl = fake_layer(dec, att, mha, W, H, heads, 0)
        o = dec.OPTDecoderLayer_forward(layer, x.clone(), attention_mask=mask, past_key_value=None,
                                        use_cache=True, gpu_layer=gl, policy=0, max_new_tokens=new)
        assert o[1] is None
        out["p0_hidden"], out["p0_key"], out["p0_value"] = torch_to_bits(o[0]), torch_to_bits(o[2]), torch_to_bits(o[3])

        layer, _ = fake_layer(dec, att, mha, W, H, heads, 3)
        o = dec.OPTDecoderLayer_forward(layer, x.clone(), attention_mask=mask, past_key_value=boot,
                                        use_cache=True, policy=3, max_new_tokens=new)
        out["p3_hidden"] = torch_to_bits(o[0])
        past = o[1]
        assert past[0].shape == (1, T, T, 1) and past[1].shape == (T + new, B, heads, H // heads)
        for s in range(new):
            xs = bits_to_torch(synth.make_hidden(seed + 100 + s, B, 1, H, identical_rows))
            o = dec.OPTDecoderLayer_forward(layer, xs, attention_mask=None, past_key_value=past,
                                            use_cache=True, policy=3, max_new_tokens=new)
            out[f"p3_dec{s}_hidden"] = torch_to_bits(o[0])
            past = o[1]
            assert past[0].shape == (1, T + s + 1, T + s + 1, 1)
        out["p3_kcache"], out["p3_vcache"] = torch_to_bits(past[1]), torch_to_bits(past[2])

        # policy 2 decode: GPU linears + host attention. The C++ kernel cannot be built (libxsmm /
        # IPEX unavailable); its pure-torch twin _IPEXScaleDotProductRef is executed instead, fed the
        # [B,h,S,d] past it expects (built from the policy-3 cache rows, identical values).
        layer2, gl2 = fake_layer(dec, att, mha, W, H, heads, 2)
        S = T
        k_past = past[1][:S].permute(1, 2, 0, 3).contiguous()
        v_past = past[2][:S].permute(1, 2, 0, 3).contiguous()
        xs = bits_to_torch(synth.make_hidden(seed + 100, B, 1, H, identical_rows))
        o = dec.OPTDecoderLayer_forward(layer2, xs, attention_mask=None, past_key_value=(k_past, v_past),
                                        use_cache=True, gpu_layer=gl2, policy=2, max_new_tokens=new)
        out["p2_dec0_hidden"] = torch_to_bits(o[0])

        # policy 1 (r05): the CPU branch of the SAME reference function -- nn.LayerNorm, nn.Linear q/k/v,
        # _IPEXScaleDotProductRef, _IPEXlinearAddRef(out_proj), _IPEXlinearReluRef(fc1), _IPEXlinearAddRef(fc2) --
        # prefill then `new` decode steps on the cache it returns ((key, value) as [B,h,S,d], mha_fusion.py:486-492).
        # The masks are what OPTDecoder.forward hands the layer (modeling_opt.py:1133: the HF 4-D additive causal
        # mask in the hidden dtype for the prefill, zeros [B,1,1,S+1] for a decode step).
        layer1, _ = fake_layer(dec, att, mha, W, H, heads, 1)
        neg = torch.finfo(torch.bfloat16).min
        cm = torch.triu(torch.full((T, T), neg, dtype=torch.bfloat16), 1)[None, None].expand(B, 1, T, T).contiguous()
        o = dec.OPTDecoderLayer_forward(layer1, x.clone(), attention_mask=cm, past_key_value=None,
                                        use_cache=True, policy=1, max_new_tokens=new)
        assert len(o) == 2
        out["p1_hidden"] = torch_to_bits(o[0])
        past1 = o[1]
        assert past1[0].shape == (B, heads, T, H // heads)
        for s in range(new):
            xs = bits_to_torch(synth.make_hidden(seed + 100 + s, B, 1, H, identical_rows))
            dm = torch.zeros(B, 1, 1, T + s + 1, dtype=torch.bfloat16)
            o = dec.OPTDecoderLayer_forward(layer1, xs, attention_mask=dm, past_key_value=past1,
                                            use_cache=True, policy=1, max_new_tokens=new)
            out[f"p1_dec{s}_hidden"] = torch_to_bits(o[0])
            past1 = o[1]
            assert past1[0].shape == (B, heads, T + s + 1, H // heads)
        # stored seq-major [S,B,h,d] like every other cache of the fixtures
        out["p1_kcache"] = torch_to_bits(past1[0].permute(2, 0, 1, 3))
        out["p1_vcache"] = torch_to_bits(past1[1].permute(2, 0, 1, 3))
    path = os.path.join(HERE, name + ".npz")
    if keep_existing and os.path.exists(path):
        old = np.load(path)
        out.update({k: old[k] for k in old.files})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items() if k not in ("cfg", "w_std")})


def run_embed_case(name, vocab, max_pos, H, B, T, past_len, seed):
    """Token + learned position embedding via the reference class (lia/modeling_opt.py:357-378)."""
    spec = importlib.util.spec_from_file_location("transformers.models.opt.lia_ref_modeling_opt",
                                                  os.path.join(REF, "lia", "modeling_opt.py"))
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "transformers.models.opt"
    na = types.ModuleType("transformers.models.opt.numa_alloc")
    na.numa_alloc_tensor = na.numa_free_tensor = None
    sys.modules["transformers.models.opt.numa_alloc"] = na
    spec.loader.exec_module(mod)
    m = synth.make_model(seed, vocab, max_pos, H, 4 * H, 0)
    pe = mod.OPTLearnedPositionalEmbedding(max_pos, H)
    pe.weight = torch.nn.Parameter(bits_to_torch(m["embed_positions"]), requires_grad=False)
    ids = torch.from_numpy(synth.make_prompt_ids(seed + 1, B, T, vocab))
    with torch.no_grad():
        tok = torch.nn.functional.embedding(ids, bits_to_torch(m["embed_tokens"]))
        pos = pe(torch.ones(B, past_len + T), past_len) if past_len == 0 else pe(torch.ones(B, past_len + T), past_len)
        hid = tok[:, -T:] + pos
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        cfg=np.array([vocab, max_pos, H, B, T, past_len, seed], dtype=np.int64),
                        hidden=torch_to_bits(hid))
    print("wrote", name)


def run_generate_case(name, vocab, max_pos, H, heads, F, L, B, T, new, seed, w_std, min_gap=0.0):
    """End-to-end greedy token IDs from stock HF OPTForCausalLM (transformers in this container),
    random-init from synth.make_model, bf16 and fp32 -- the form of the reference's own generate
    parity test (tests/cpu/test_ipex_optimize_transformers.py:403-446)."""
    from transformers import OPTConfig, OPTForCausalLM
    m = synth.make_model(seed, vocab, max_pos, H, F, L, w_std)
    cfg = OPTConfig(vocab_size=vocab, hidden_size=H, num_hidden_layers=L, ffn_dim=F, num_attention_heads=heads,
                    max_position_embeddings=max_pos, word_embed_proj_dim=H, do_layer_norm_before=True,
                    dropout=0.0, attention_dropout=0.0, activation_function="relu", bos_token_id=2, eos_token_id=2,
                    pad_token_id=1)
    ids = torch.from_numpy(synth.make_prompt_ids(seed + 1, B, T, vocab))
    res = {"cfg": np.array([vocab, max_pos, H, heads, F, L, B, T, new, seed], dtype=np.int64),
           "w_std": np.array([w_std], dtype=np.float64)}
    for dt, tag in ((torch.bfloat16, "bf16"), (torch.float32, "fp32")):
        model = OPTForCausalLM(cfg).to(dt).eval()
        sd = {"model.decoder.embed_tokens.weight": bits_to_torch(m["embed_tokens"]),
              "model.decoder.embed_positions.weight": bits_to_torch(m["embed_positions"]),
              "model.decoder.final_layer_norm.weight": bits_to_torch(m["final_ln_w"]),
              "model.decoder.final_layer_norm.bias": bits_to_torch(m["final_ln_b"]),
              "lm_head.weight": bits_to_torch(m["embed_tokens"])}
        hf = {"ln1": "self_attn_layer_norm", "q": "self_attn.q_proj", "k": "self_attn.k_proj", "v": "self_attn.v_proj",
              "out": "self_attn.out_proj", "ln2": "final_layer_norm", "fc1": "fc1", "fc2": "fc2"}
        for i, lw in enumerate(m["layers"]):
            for n, v in lw.items():
                base, kind = n.rsplit("_", 1)
                sd[f"model.decoder.layers.{i}.{hf[base]}.{'weight' if kind == 'w' else 'bias'}"] = bits_to_torch(v)
        missing = model.load_state_dict({k: v.to(dt) for k, v in sd.items()}, strict=False)
        assert not missing.unexpected_keys, missing
        with torch.no_grad():
            out = model.generate(ids, attention_mask=torch.ones_like(ids), do_sample=False, num_beams=1,
                                 max_new_tokens=new, min_new_tokens=new, output_scores=True,
                                 return_dict_in_generate=True)
        res[f"ids_{tag}"] = out.sequences.numpy().astype(np.int64)
        sc = torch.stack(out.scores, 1).float()  # [B,new,vocab]
        top2 = sc.topk(2, -1).values
        res[f"gap_{tag}"] = (top2[..., 0] - top2[..., 1]).numpy()
        if tag == "bf16":
            res["logits0_bf16"] = torch_to_bits(out.scores[0])
    ok = bool((res["ids_bf16"] == res["ids_fp32"]).all()) and res["gap_bf16"].min() >= min_gap
    if not ok:
        return False
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **res)
    print("wrote", name, "seed", seed, "min top-2 gap bf16/fp32:", res["gap_bf16"].min(), res["gap_fp32"].min())
    return True


def search_generate_case(name, *a, seed0, w_std, min_gap=0.12):
    """Greedy IDs are only a stable golden when no step is a near-tie (bf16 logits of a 50k-word
    vocabulary tie often): scan seeds for a case whose smallest top-2 logit gap is
    several bf16 ulps and whose bf16 and fp32 IDs agree, and record that seed in the fixture."""
    for seed in range(seed0, seed0 + 200):
        if run_generate_case(name, *a, seed, w_std, min_gap=min_gap):
            return
    raise RuntimeError("no stable seed found for " + name)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    dec, att, mha = import_reference()
    # name, H, heads, F, B, T, new, seed, w_std, identical_rows
    run_layer_case(dec, att, mha, "layer_h128", 128, 4, 512, 2, 8, 3, 11, 0.12, False)
    run_layer_case(dec, att, mha, "layer_h256_d64", 256, 4, 1024, 4, 32, 2, 12, 0.08, False)
    run_layer_case(dec, att, mha, "layer_h512_d128", 512, 4, 2048, 2, 40, 2, 13, 0.06, True)
    run_layer_case(dec, att, mha, "layer_h256_b1", 256, 4, 1024, 1, 17, 2, 14, 0.08, False)
    # r03: the reference's own functions at the HEADLINE layer shape (OPT-30B: 7168 / 56 heads / 28672, N(0, 0.02) weights): 1.2 GB
    # of weights re-derived from the seed, 1.2 MB of outputs kept.  Not named layer_* on purpose: at this width two correct
    # implementations agree to one bf16 quantum, not bit for bit (tests/test_gpu_fullsize_oracle.py), so it has tests of its own.
    # Unlike the small cases this one is not bit-reproducible run to run: torch's CPU bf16 GEMM at K = 7168 / 28672 lands 0.1-0.2 % of
    # the decode outputs (and a handful of prefill outputs) one ulp apart between two runs of this script on the same machine.  The
    # committed fixture keeps the r03 run's vectors byte for byte; the p1_* vectors were appended from an r05 run (`keep_existing`).
    run_layer_case(dec, att, mha, "fullsize_layer_opt30b", 7168, 56, 28672, 2, 8, 1, 15, 0.02, False, keep_existing=True)
    run_embed_case("embed_prefill", 512, 64, 128, 2, 9, 0, 21)
    run_embed_case("embed_decode", 512, 64, 128, 2, 1, 9, 21)
    search_generate_case("generate_tiny", 512, 64, 128, 4, 512, 3, 2, 8, 6, seed0=31, w_std=0.12)
    search_generate_case("generate_h256", 1024, 128, 256, 4, 1024, 4, 4, 16, 8, seed0=300, w_std=0.08)
    search_generate_case("generate_h512_d128", 2048, 128, 512, 4, 2048, 2, 2, 24, 6, seed0=500, w_std=0.06)


if __name__ == "__main__":
    main()
