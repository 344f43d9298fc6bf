"""-m gpu: the BASELINE.json configurations and entry points that round 1 left without a GPU test.

  * configs[2] "opt-175b dummy weights ... --enable-cxl --pin-weight": the reference's dummy recipe (torch.rand_like,
    llm/utils/opt-weight-gen.py:61-62; seeded here) through the layer operator against the oracle on hidden states -- U[0,1)
    weights saturate softmax and argmax, so token ids say nothing -- and through generate() with the streamed layers in the
    NUMA / CXL tier;
  * configs[3] Llama-3-8B (H 4096, 32 q / 8 kv heads, F 14336; B 128, T 1024, decode past S = 1024): size-independent
    properties of lia_llama_layer_forward at full size, the oracle needing minutes per case there;
  * the harness itself: lia_amd.run_generation.main([...]) with the README command line (README.md:78) on a checkpoint
    directory the test writes (HF layout: config.json + model.safetensors), ids == the HF golden ids.
"""
import ctypes
import json
import os

import numpy as np
import pytest

import synth
from test_gpu_ops import assert_close, to_bits, to_dev

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(autouse=True)
def cxl_nodes_of_this_box():
    from lia_amd import hostinfo
    from lia_amd.cxl.numa_alloc import set_cxl_nodes
    set_cxl_nodes(hostinfo.numa_nodes()[:2] or [0])


# ---------------------------------------------------------------------------------------------------------------------
# configs[2]: dummy U[0,1) weights
# ---------------------------------------------------------------------------------------------------------------------
def _layer_dict(model, li):
    """the 16 tensors of layer li back on the host as bf16 bit arrays, whatever tier holds them"""
    H, F = model.shape.hidden, model.shape.ffn
    flat = model.layers[li]._raw_on_device().cpu().numpy().view(np.uint16)
    shapes = synth.layer_shapes(H, F)
    return {n: flat[model.offsets[i] // 2: model.offsets[i] // 2 + int(np.prod(shapes[n]))].reshape(shapes[n]).copy()
            for i, n in enumerate(synth.LAYER_TENSORS)}


def test_uniform01_dummy_weights_layer_matches_oracle(oracle):
    """random_init(init="uniform01") = the reference's opt-weight-gen recipe: EVERY parameter ~ U[0,1), LayerNorm included.
    One layer, prefill + one decode step, policies 3 and 0/2, against the oracle on the hidden states.  Activations reach
    the thousands (fc2 sums 4 F positive products), so the bound is relative: 2 bf16 ulps of the value, 99.5 % of the elements
    (a saturated softmax that picks another key on a near-tie moves a few elements further)."""
    import torch
    from lia_amd import _native as N, ops
    from lia_amd.model import LiaOPTModel, OPTShape
    shape = OPTShape("dummy", 512, 4, 2048, 2, vocab=512, max_pos=64)
    model = LiaOPTModel.random_init(shape, seed=3, init="uniform01", n_gpu_layers=2)
    W = _layer_dict(model, 1)
    assert 0.45 < synth.bf16_bits_to_f32(W["fc1_w"]).mean() < 0.55 and synth.bf16_bits_to_f32(W["q_b"]).min() >= 0.0
    for n in ("ln1_w", "ln2_b", "out_b"):          # rand_like on every parameter: LayerNorm weights / biases are U[0,1) too
        v = synth.bf16_bits_to_f32(W[n])
        assert 0.0 <= v.min() and v.max() <= 1.0 and 0.4 < v.mean() < 0.6 and v.std() > 0.2, n
    B, T, heads, d = 4, 16, 4, 128
    ctx = ops.Context(0, ops.workspace_bytes(model.desc, B * (T + 2)))
    wptrs = ops.weight_ptr_array(model.layers[1].device_ptr(), model.offsets)
    x = synth.make_hidden(11, B, T, 512)
    xs = synth.make_hidden(12, B, 1, 512)

    def close(got, ref, what):
        a, b = synth.bf16_bits_to_f32(got), synth.bf16_bits_to_f32(ref)
        ratio = np.abs(a - b) / (0.02 + 0.016 * np.abs(b))
        assert np.isfinite(a).all() and np.quantile(ratio, 0.995) <= 1.0 and ratio.max() <= 16.0, \
            f"{what}: q99.5 {np.quantile(ratio, 0.995):.2f} max {ratio.max():.2f} (|ref| up to {np.abs(b).max():.0f})"

    kc = torch.zeros((T + 2, B, heads, d), dtype=torch.bfloat16, device="cuda")
    vc = torch.zeros_like(kc)
    kv = N.KV(kc.data_ptr(), vc.data_ptr(), T + 2, B, 1)
    xd, xsd = to_dev(torch, x), to_dev(torch, xs)
    y, ys = torch.empty_like(xd), torch.empty_like(xsd)
    ctx.layer_forward(model.desc, 3, wptrs, xd, y, kv, B, T, 0)
    ctx.layer_forward(model.desc, 3, wptrs, xsd, ys, kv, B, 1, T)
    ctx.synchronize()
    okc, ovc = np.zeros((T + 2, B, heads, d), np.uint16), np.zeros((T + 2, B, heads, d), np.uint16)
    ref = oracle.layer_forward(3, W, x, okc, ovc, 0, heads)
    ref_s = oracle.layer_forward(3, W, xs, okc, ovc, T, heads)
    close(to_bits(y), ref, "uniform01 policy 3 prefill")
    close(to_bits(ys), ref_s, "uniform01 policy 3 decode")
    assert_close(to_bits(kc)[:T + 1], okc[:T + 1], 0.05, 0.016, 0.9, "uniform01 K cache")
    # the cooperative pair: policy 0 prefill into a host cache, policy 2 decode (host attention) on it
    hk = torch.zeros((T + 2, B, heads, d), dtype=torch.bfloat16).pin_memory()
    hv = torch.zeros_like(hk).pin_memory()
    kvh = N.KV(hk.data_ptr(), hv.data_ptr(), T + 2, B, 0)
    y0, y2 = torch.empty_like(xd), torch.empty_like(xsd)
    ctx.layer_forward(model.desc, 0, wptrs, xd, y0, kvh, B, T, 0)
    ctx.synchronize()
    ctx.kv_store_wait()
    assert torch.equal(y0, y)
    ctx.layer_forward(model.desc, 2, wptrs, xsd, y2, kvh, B, 1, T)
    ctx.synchronize()
    okc2, ovc2 = okc.copy(), ovc.copy()
    okc2[T:] = 0
    ovc2[T:] = 0
    close(to_bits(y2), oracle.layer_forward(2, W, xs, okc2, ovc2, T, heads), "uniform01 policy 2 decode")
    ctx.close()
    model.close()


@pytest.mark.parametrize("fmt", ["raw", "pack10"])
def test_uniform01_generate_streams_from_the_cxl_tier(fmt, monkeypatch):
    """configs[2] in miniature (OPT-175B's head_dim 128, dummy weights, gpu% 25, --enable-cxl --pin-weight, 0/2 and the
    KV-in-HBM pair 3/3): the streamed path must reproduce the all-resident run of the same kernels bit for bit (3/3: same
    arithmetic, only the weights travel), and the NUMA tier must really hold the layers in the requested wire format."""
    import torch
    from lia_amd.generation import generate
    from lia_amd.model import LiaOPTModel, OPTShape
    monkeypatch.setenv("LIA_STREAM_FORMAT", fmt)
    shape = OPTShape("dummy", 512, 4, 2048, 4, vocab=1024, max_pos=64)
    ids = torch.from_numpy(synth.make_prompt_ids(5, 4, 12, 1024))
    model = LiaOPTModel.random_init(shape, seed=9, init="uniform01", n_gpu_layers=4)
    ref, _, ref_logits = generate(model, ids, max_new_tokens=4, min_new_tokens=4, return_logits=True, prefill_policy=0,
                                  decoding_policy=2, gpu_percentage=100, pin_weight=True)
    # same model object, re-tiered: 1 resident layer, 3 in the CXL pool
    out, _, logits = generate(model, ids, max_new_tokens=4, min_new_tokens=4, return_logits=True, prefill_policy=3,
                              decoding_policy=3, gpu_percentage=25, pin_weight=True, enable_cxl=True)
    assert [st.tier for st in model.layers] == ["device", "cxl", "cxl", "cxl"]
    assert all(st.packed == {"raw": 0, "pack10": 10}[fmt] for st in model.layers[1:])
    if fmt == "pack10":       # U[0,1): half the values share one exponent, 7/8 fall on three -> far below the Gaussian's 10.8 bits
        assert all(st.stream_bytes < 0.66 * st.nbytes for st in model.layers[1:])
    assert torch.equal(out, ref) and all(torch.equal(a, b) for a, b in zip(logits, ref_logits))
    out2 = generate(model, ids, max_new_tokens=4, min_new_tokens=4, prefill_policy=0, decoding_policy=2, gpu_percentage=25,
                    pin_weight=True, enable_cxl=True)
    assert out2.shape == ref.shape and (out2[:, :13] == ref[:, :13]).all()      # first token: the prefill is the same arithmetic
    model._lia_scheduler.close()
    model.close()


# ---------------------------------------------------------------------------------------------------------------------
# configs[3]: Llama-3-8B at size
# ---------------------------------------------------------------------------------------------------------------------
LH, LHEADS, LKV, LF = 4096, 32, 8, 14336


@pytest.fixture(scope="module")
def llama_big():
    import torch
    from lia_amd import _native as N, ops
    from lia_amd.llama import rope_tables
    lib = N.lib()
    g = torch.Generator(device="cuda").manual_seed(21)
    d = LH // LHEADS

    def make(kv_heads, kw=None, vw=None):
        desc = N.LlamaDesc(LH, LHEADS, kv_heads, LF, 1e-5, 32)     # interleaved gate|up rows (random weights: any order is a model)
        offs = (ctypes.c_size_t * 9)()
        total = ctypes.c_size_t()
        N.check(lib.lia_llama_pack_offsets(ctypes.byref(desc), ctypes.byref(offs), ctypes.byref(total)))
        flat = torch.zeros(total.value // 2, dtype=torch.bfloat16, device="cuda")
        return desc, list(offs), flat

    desc, offs, flat = make(LKV)
    KD = LKV * d
    dims = {1: LH * LH, 2: KD * LH, 3: KD * LH, 4: LH * LH, 6: LF * LH, 7: LF * LH, 8: LH * LF}
    for i, n in dims.items():
        flat[offs[i] // 2: offs[i] // 2 + n] = (0.02 * torch.randn(n, generator=g, device="cuda")).to(torch.bfloat16)
    for i in (0, 5):
        flat[offs[i] // 2: offs[i] // 2 + LH] = (1.0 + 0.1 * torch.randn(LH, generator=g, device="cuda")).to(torch.bfloat16)
    # the multi-head twin: every KV head's rows repeated for the 4 query heads of its group
    desc_m, offs_m, flat_m = make(LHEADS)
    for i in (0, 1, 4, 5, 6, 7, 8):
        n = dims.get(i, LH)
        flat_m[offs_m[i] // 2: offs_m[i] // 2 + n] = flat[offs[i] // 2: offs[i] // 2 + n]
    for i in (2, 3):
        w = flat[offs[i] // 2: offs[i] // 2 + KD * LH].view(LKV, d, LH)
        flat_m[offs_m[i] // 2: offs_m[i] // 2 + LH * LH] = w[:, None].expand(LKV, LHEADS // LKV, d, LH).reshape(-1)
    rows = 128 * 1024
    ctx = ops.Context(0, lib.lia_llama_workspace_bytes(ctypes.byref(desc_m), rows))
    cos, sin = rope_tables(1280, d, 500000.0)
    torch.cuda.synchronize()
    yield dict(torch=torch, N=N, lib=lib, ctx=ctx, cos=cos, sin=sin, d=d,
               gqa=(desc, (ctypes.c_void_p * 9)(*[flat.data_ptr() + o for o in offs]), flat, LKV),
               mha=(desc_m, (ctypes.c_void_p * 9)(*[flat_m.data_ptr() + o for o in offs_m]), flat_m, LHEADS))
    ctx.close()


def _lkv(lb, which, smax, B):
    torch, N = lb["torch"], lb["N"]
    kvh = lb[which][3]
    k = torch.zeros((smax, B, kvh, lb["d"]), dtype=torch.bfloat16, device="cuda")
    v = torch.zeros_like(k)
    return k, v, N.KV(k.data_ptr(), v.data_ptr(), smax, B, 1)


def _lrun(lb, which, x, kv, T, pos0, b0=0, y=None):
    torch, N, lib, ctx = lb["torch"], lb["N"], lb["lib"], lb["ctx"]
    desc, w = lb[which][0], lb[which][1]
    y = torch.empty_like(x) if y is None else y
    torch.cuda.synchronize()
    N.check(lib.lia_llama_layer_forward(ctx.handle, ctypes.byref(desc), ctypes.byref(w), ctypes.c_void_p(x.data_ptr()),
                                        ctypes.c_void_p(y.data_ptr()), ctypes.byref(kv), ctypes.c_void_p(lb["cos"].data_ptr()),
                                        ctypes.c_void_p(lb["sin"].data_ptr()), x.shape[0], T, pos0, b0, ctypes.c_void_p(ctx.stream)),
            "lia_llama_layer_forward")
    ctx.synchronize()
    return y


def _lx(lb, B, T, seed, identical=False):
    torch = lb["torch"]
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn((1 if identical else B, T, LH), generator=g, device="cuda").to(torch.bfloat16)
    x = x.repeat(B, 1, 1).contiguous() if identical else x.contiguous()
    torch.cuda.synchronize()
    return x


def test_llama3_8b_full_batch_identical_rows_and_minibatches(llama_big):
    """configs[3]'s own shape, B = 128 x T = 1024 = 131072 rows in one call: identical rows in -> identical rows out, and the
    prefill cut into 4 minibatches of 32 rows writes the same hidden states and the same cache, bit for bit."""
    lb, torch = llama_big, llama_big["torch"]
    B, T = 128, 1024
    x = _lx(lb, B, T, 1, identical=True)
    k, v, kv = _lkv(lb, "gqa", T + 8, B)
    y = _lrun(lb, "gqa", x, kv, T, 0)
    assert torch.isfinite(y.float()).all()
    assert (y == y[0:1]).all() and (k[:T] == k[:T, 0:1]).all()
    k2, v2, kv2 = _lkv(lb, "gqa", T + 8, B)
    y2 = torch.empty_like(x)
    for i in range(4):
        _lrun(lb, "gqa", x[i * 32:(i + 1) * 32], kv2, T, 0, b0=i * 32, y=y2[i * 32:(i + 1) * 32])
    assert torch.equal(y2, y) and torch.equal(k2[:T], k[:T]) and torch.equal(v2[:T], v[:T])
    # one decode step of the whole batch on that cache: rows stay identical
    xs = _lx(lb, B, 1, 2, identical=True)
    ys = _lrun(lb, "gqa", xs, kv, 1, T)
    assert torch.isfinite(ys.float()).all() and (ys == ys[0:1]).all()


def test_llama3_8b_decode_past_1024_equals_prefill_last_position(llama_big):
    """KV-cache equivalence where the grouped decode kernel has never been checked (S > 1024): prefill 1152 tokens, decode token
    1153 over the cache == last position of a prefill of all 1153 tokens, to rounding; plus batch independence at this size."""
    lb, torch = llama_big, llama_big["torch"]
    B, T = 8, 1152
    x = _lx(lb, B, T + 1, 3)
    k, v, kv = _lkv(lb, "gqa", T + 4, B)
    y_full = _lrun(lb, "gqa", x, kv, T + 1, 0)
    k2, v2, kv2 = _lkv(lb, "gqa", T + 4, B)
    _lrun(lb, "gqa", x[:, :T].contiguous(), kv2, T, 0)
    y_dec = _lrun(lb, "gqa", x[:, T:T + 1].contiguous(), kv2, 1, T)
    a, b = y_dec[:, 0].float(), y_full[:, T].float()
    assert (a - b).abs().max() <= 0.07 + 0.016 * b.abs().max(), float((a - b).abs().max())
    assert (k2[:T + 1].float() - k[:T + 1].float()).abs().max() <= 0.03 + 0.008 * k.float().abs().max()
    # rows 2..4 alone == rows 2..4 inside the batch (what makes batch sharding valid), to rounding: different GEMM regimes
    ks, vs, kvs = _lkv(lb, "gqa", T + 4, 3)
    y_sub = _lrun(lb, "gqa", x[2:5].contiguous(), kvs, T + 1, 0)
    dsub = (y_sub.float() - y_full[2:5].float()).abs()
    assert dsub.max() <= 0.016 * y_full.float().abs().max() and (y_sub == y_full[2:5]).float().mean() > 0.5


def test_llama3_8b_gqa_heads_only_see_their_group(llama_big):
    """Grouped-query attention == multi-head attention whose K / V projections repeat each KV head for the 4 query heads of
    its group (HF repeat_kv): the grouped prefill kernel at T = 1024 and the grouped decode kernel at S = 1152 must give what the
    per-head kernels give on the repeated heads, to rounding -- a query head that read another group's keys would not."""
    lb, torch = llama_big, llama_big["torch"]
    B, T = 4, 1024
    x = _lx(lb, B, T, 5)
    kg, vg, kvg = _lkv(lb, "gqa", 1160, B)
    km, vm, kvm = _lkv(lb, "mha", 1160, B)
    yg = _lrun(lb, "gqa", x, kvg, T, 0)
    ym = _lrun(lb, "mha", x, kvm, T, 0)
    tol = 0.016 * float(ym.float().abs().max())
    assert (yg.float() - ym.float()).abs().max() <= tol and (yg == ym).float().mean() > 0.9
    # the caches agree head for head: MHA head h holds what GQA head h // 4 holds
    assert torch.equal(km[:T].view(T, B, LKV, LHEADS // LKV, -1)[:, :, :, 0], kg[:T])
    for s in range(T, 1152):          # grow both caches past 1024 with decode steps (cheap: one row each)
        xs = _lx(lb, B, 1, 100 + s)
        ygs = _lrun(lb, "gqa", xs, kvg, 1, s)
        yms = _lrun(lb, "mha", xs, kvm, 1, s)
        if s in (T, 1100, 1151):
            assert (ygs.float() - yms.float()).abs().max() <= 0.016 * float(yms.float().abs().max()) + 0.03, s


# ---------------------------------------------------------------------------------------------------------------------
# the harness entry point on an on-disk HF checkpoint
# ---------------------------------------------------------------------------------------------------------------------
def _write_hf_checkpoint(path, m, c, blocked=False):
    """config.json + model.safetensors in the HF OPT layout; blocked=True stores the Linear weights in the reference's
    TPP-blocked [N/16, K/64, 32, 16, 2] wire format (_weight_prepack.py:19-63) as an IPEX-prepacked checkpoint would."""
    import torch
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    cfg = dict(model_type="opt", hidden_size=c["H"], ffn_dim=c["F"], num_hidden_layers=c["L"], num_attention_heads=c["heads"],
               vocab_size=c["vocab"], max_position_embeddings=c["max_pos"], do_layer_norm_before=True, word_embed_proj_dim=c["H"],
               torch_dtype="bfloat16")
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))

    def t(bits):
        return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.bfloat16).clone()

    hf = {"ln1": "self_attn_layer_norm", "q": "self_attn.q_proj", "k": "self_attn.k_proj", "v": "self_attn.v_proj",
          "out": "self_attn.out_proj", "ln2": "final_layer_norm", "fc1": "fc1", "fc2": "fc2"}
    sd = {"model.decoder.embed_tokens.weight": t(m["embed_tokens"]), "model.decoder.embed_positions.weight": t(m["embed_positions"]),
          "model.decoder.final_layer_norm.weight": t(m["final_ln_w"]), "model.decoder.final_layer_norm.bias": t(m["final_ln_b"])}
    for i, lw in enumerate(m["layers"]):
        for short, name in hf.items():
            w = lw[short + "_w"]
            if blocked and w.ndim == 2:
                n, k = w.shape
                w = w.reshape(n // 16, 16, k // 64, 32, 2).transpose(0, 2, 3, 1, 4)
            sd[f"model.decoder.layers.{i}.{name}.weight"] = t(w)
            sd[f"model.decoder.layers.{i}.{name}.bias"] = t(lw[short + "_b"])
    save_file(sd, os.path.join(path, "model.safetensors"))


@pytest.mark.parametrize("blocked", [False, True], ids=["plain", "tpp-blocked"])
def test_run_generation_main_on_a_checkpoint_directory(tmp_path, capsys, blocked):
    """`python run.py --benchmark -m <dir> --dtype bfloat16 --ipex --input-tokens T --max-new-tokens N --batch-size B
    --token-latency --num-iter 3 --num-warmup 1 --greedy --prefill-policy 0 --decoding-policy 2 --gpu-percentage 50
    --num-minibatch 2 --pin-weight` (README.md:78) through lia_amd.run_generation.main: load_hf_opt (safetensors, optional
    un-blocking), the identical-row batch, generate with the LIA kwargs, the four summary lines."""
    from lia_amd import run_generation
    z = np.load(os.path.join(GOLD, "generate_h256.npz"))
    vocab, max_pos, H, heads, F, L, B, T, new, seed = [int(v) for v in z["cfg"]]
    c = dict(vocab=vocab, max_pos=max_pos, H=H, heads=heads, F=F, L=L)
    m = synth.make_model(seed, vocab, max_pos, H, F, L, float(z["w_std"][0]))
    ckpt = str(tmp_path / "opt-test")
    _write_hf_checkpoint(ckpt, m, c, blocked=blocked)
    prompt = synth.make_prompt_ids(seed + 1, B, T, vocab)[0]
    seen = {}
    orig = run_generation.synthetic_prompt
    run_generation.synthetic_prompt = lambda vocab_, n, batch, seed=0: seen.setdefault(
        "ids", __import__("torch").from_numpy(np.tile(prompt[None, :], (batch, 1))))
    try:
        res = run_generation.main(["--benchmark", "-m", ckpt, "--dtype", "bfloat16", "--ipex", "--input-tokens", str(T),
                                   "--max-new-tokens", str(new), "--batch-size", str(B), "--token-latency", "--num-iter", "3",
                                   "--num-warmup", "1", "--greedy", "--prefill-policy", "0", "--decoding-policy", "2",
                                   "--gpu-percentage", "50", "--num-minibatch", "2", "--pin-weight", "--stream-format", "pack10"])
    finally:
        run_generation.synthetic_prompt = orig
    text = capsys.readouterr().out
    for line in ("Inference latency:", "First token average latency:", "Average 2... latency:", "P90 2... latency:", "P99 2... latency:"):
        assert line in text, text[-2000:]
    assert res["prefill_ms"] > 0 and res["decode_tokens_per_s"] > 0 and res["inference_latency_s"] > 0
    # the ids the harness printed for every iteration are the HF golden continuation
    want = str(z["ids_bf16"][0, T:].tolist())
    assert text.count(want) == 3, text[-2000:]
    assert seen["ids"].shape == (B, T)


def test_run_generation_prompt_through_the_checkpoints_tokenizer(tmp_path, capsys):
    """`run.py -m <dir> --prompt "..."` (run_generation.py:87,264-285,321): the directory's tokenizer turns the text into the ids,
    every row is that prompt, the continuation is the HF golden one and the decoded text is printed"""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    from lia_amd import run_generation
    z = np.load(os.path.join(GOLD, "generate_h256.npz"))
    vocab, max_pos, H, heads, F, L, B, T, new, seed = [int(v) for v in z["cfg"]]
    c = dict(vocab=vocab, max_pos=max_pos, H=H, heads=heads, F=F, L=L)
    m = synth.make_model(seed, vocab, max_pos, H, F, L, float(z["w_std"][0]))
    ckpt = str(tmp_path / "opt-test")
    _write_hf_checkpoint(ckpt, m, c, blocked=False)
    word = lambda i: {1: "<pad>", 2: "</s>", 3: "<unk>"}.get(int(i), f"w{int(i)}")      # noqa: E731  (special tokens are matched as substrings: keep them unlike the words)
    tok = Tokenizer(models.WordLevel({word(i): i for i in range(vocab)}, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", pad_token="<pad>", eos_token="</s>").save_pretrained(ckpt)
    prompt = synth.make_prompt_ids(seed + 1, B, T, vocab)[0]
    text_in = " ".join(word(i) for i in prompt)
    res = run_generation.main(["--benchmark", "-m", ckpt, "--dtype", "bfloat16", "--prompt", text_in, "--max-new-tokens", str(new),
                               "--batch-size", str(B), "--token-latency", "--num-iter", "2", "--num-warmup", "1", "--greedy",
                               "--prefill-policy", "0", "--decoding-policy", "2", "--gpu-percentage", "50", "--pin-weight"])
    text = capsys.readouterr().out
    assert f"---- Prompt size: {T}" in text and res["decode_tokens_per_s"] > 0
    assert text.count(str(z["ids_bf16"][0, T:].tolist())) == 2, text[-2000:]
    cont = [int(i) for i in z["ids_bf16"][0, T:]]
    assert all(i > 3 for i in cont) and " ".join(word(i) for i in cont) in text      # batch_decode of prompt + continuation


# ---------------------------------------------------------------------------------------------------------------------
# f-2: the on-disk streaming format + the dummy-weight generator twin; f-3: --auto-plan
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("wire", ["pack10", "raw"])
def test_dummy_checkpoint_directory_streams_like_the_generator(tmp_path, wire):
    """`python -m lia_amd.packed_checkpoint --model M --save_dir D` (twin of llm/utils/opt-weight-gen.py): the directory holds
    one wire buffer per layer; load_packed maps the files (mmap + hipHostRegister, no second host copy) and the run from the
    directory reproduces the run from LiaOPTModel.random_init with the same seed bit for bit (ids AND logits)."""
    import torch
    from lia_amd import packed_checkpoint as pc
    from lia_amd.generation import generate
    from lia_amd.model import SHAPES, LiaOPTModel, OPTShape
    shape = OPTShape("opt-dummy", 512, 4, 2048, 4, vocab=1024, max_pos=64)
    SHAPES["opt-dummy"] = shape
    try:
        man = pc.main(["--model", "opt-dummy", "--save_dir", str(tmp_path / "d"), "--seed", "9", "--wire", wire])
    finally:
        del SHAPES["opt-dummy"]
    assert man["format"] == pc.FORMAT and len(man["layers"]) == 4 and all(e["wire"] == {"pack10": 10, "raw": 0}[wire] for e in man["layers"])
    if wire == "pack10":
        assert sum(e["bytes"] for e in man["layers"]) < 0.66 * sum(e["raw_bytes"] for e in man["layers"])
    ids = torch.from_numpy(synth.make_prompt_ids(5, 4, 12, 1024))
    ref_model = LiaOPTModel.random_init(shape, seed=9, init="uniform01", n_gpu_layers=4)
    ref, _, ref_logits = generate(ref_model, ids, max_new_tokens=4, min_new_tokens=4, return_logits=True, prefill_policy=0,
                                  decoding_policy=2, gpu_percentage=100, pin_weight=True)
    model = pc.load_packed(str(tmp_path / "d"), n_gpu_layers=1)
    assert [st.tier for st in model.layers] == ["device", "mapped", "mapped", "mapped"]
    assert all(st.packed == {"pack10": 10, "raw": 0}[wire] and st.is_dma_able() for st in model.layers[1:])
    from lia_amd.scheduler import OffloadScheduler
    model._lia_scheduler = OffloadScheduler(model, wire=wire)
    out, _, logits = generate(model, ids, max_new_tokens=4, min_new_tokens=4, return_logits=True, prefill_policy=3, decoding_policy=3,
                              gpu_percentage=25, pin_weight=True)
    assert [st.tier for st in model.layers] == ["device", "mapped", "mapped", "mapped"]      # streamed straight from the files
    assert torch.equal(out, ref) and all(torch.equal(a, b) for a, b in zip(logits, ref_logits))
    # the mapped layers re-tier like any others: policy 1 wants raw host copies (r06: beside the packed one, which keeps serving the
    # link-bound policy-0 prefill -- scheduler.placement_formats)
    out1 = generate(model, ids, max_new_tokens=2, min_new_tokens=2, prefill_policy=0, decoding_policy=1, gpu_percentage=25, pin_weight=True)
    # (a raw layer file that is mapped and registered already IS a raw pinned copy and stays where it is)
    assert out1.shape == (4, 14) and all(st.raw_host_ptr() is not None and st.tier == ("pinned" if wire == "pack10" else "mapped") for st in model.layers[1:])
    assert all(st.packed == {"pack10": 10, "raw": 0}[wire] for st in model.layers[1:])
    model._lia_scheduler.close()
    model.close()
    ref_model.close()


def test_save_packed_round_trip_and_auto_plan_through_the_harness(tmp_path, capsys, monkeypatch):
    """save_packed(model) -> `run.py -m <dir> --auto-plan ...`: the planner calibrates the box through the C ABI (a few
    seconds), chooses gpu% / policies (--plan-max-gpu-percentage caps the resident share, so layers MUST stream), the harness loads the
    packed directory, and the ids are the HF golden ids."""
    from lia_amd import packed_checkpoint as pc, planner, run_generation
    from lia_amd.model import LiaOPTModel, OPTShape
    z = np.load(os.path.join(GOLD, "generate_h256.npz"))
    vocab, max_pos, H, heads, F, L, B, T, new, seed = [int(v) for v in z["cfg"]]
    m = synth.make_model(seed, vocab, max_pos, H, F, L, float(z["w_std"][0]))
    model = LiaOPTModel.from_numpy(OPTShape("opt-golden", H, heads, F, L, vocab=vocab, max_pos=max_pos), m)
    d = str(tmp_path / "packed")
    man = pc.save_packed(model, d, wire=10)
    model.close()
    assert all(e["wire"] == 10 and e["bytes"] < e["raw_bytes"] for e in man["layers"])
    box = planner.calibrate()
    c = box.calibrated
    assert c["seconds"] < 10 and 15 < c["link_gbs"] < 70 and 1500 < c["hbm_gbs"] < 8000 and 400 < c["mfma_tflops"] < 2500 and c["host_attention_gbs"] > 1
    prompt = synth.make_prompt_ids(seed + 1, B, T, vocab)[0]
    monkeypatch.setattr(run_generation, "synthetic_prompt",
                        lambda vocab_, n, batch, seed=0: __import__("torch").from_numpy(np.tile(prompt[None, :], (batch, 1))))
    res = run_generation.main(["--benchmark", "-m", d, "--dtype", "bfloat16", "--input-tokens", str(T), "--max-new-tokens", str(new),
                               "--batch-size", str(B), "--token-latency", "--num-iter", "2", "--num-warmup", "1", "--greedy", "--auto-plan",
                               "--plan-max-gpu-percentage", "50"])         # at most half of the layers may be resident: the others MUST stream
    text = capsys.readouterr().out
    assert "auto-plan: gpu%=" in text and "auto-plan: gpu%=100" not in text and res["decode_tokens_per_s"] > 0
    assert text.count(str(z["ids_bf16"][0, T:].tolist())) == 2, text[-3000:]
