"""-m gpu: WHOLE-MODEL parity at production width -- several full-width layers plus the real vocabulary through generate(), against
oracle.generate (oracle/lia_oracle.py: greedy_search.py:144-424 over the layer loop of lia/modeling_opt.py:1222-1558 and the
tied lm_head of models.py:424-431).

The per-layer tests of test_gpu_fullsize_oracle.py show that one OPT-30B-wide layer is the oracle's arithmetic up to the fp32
summation order, with each op's rare one-ulp flips amplified by the next GEMM.  What they cannot show is whether that matters for
the product's OUTPUT: the ids.  Here 3 OPT-30B-shaped layers (7168 / 56 / 28672) + the 50272-entry head, and 2 Llama-3-8B-shaped
layers (4096 / 32 / 8 / 14336) + the 128256-entry head, generate 8 tokens from B = 4 DIFFERENT prompt rows of 32 tokens under
every policy pair of the path (0/2 with one resident + two streamed layers, raw and pack10 on the wire; 3/3 all-resident; 1/1 all
on the host cores), and the ids must be the oracle's -- except where the ORACLE's own top-2 logits are within two bf16 quanta
(parity_util.ids_equal_or_near_tie), after which that row is not compared further.  Every test prints, per step, the smallest
top-2 gap over the rows, the share of bit-identical logits and the largest logit error in quanta; LIA_PARITY_STATS=<file>
appends them as JSON lines (DESIGN.md section 7 quotes a run).

Also here: one OPT-175B-shaped layer (12288 / 96 / 49152) with the reference's dummy weights -- EVERY parameter ~ U[0,1),
opt-weight-gen.py:61-62 -- decode B = 32 under policies 3 and 2 against the oracle: sums of tens of thousands of positive
products, where the split-K order matters most.
"""
import json
import os

import numpy as np
import pytest

import synth
from parity_util import ids_equal_or_near_tie
from test_gpu_ops import to_bits

pytestmark = pytest.mark.gpu


def _bits(t):
    import torch
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def _draw(torch, g, shape, std=0.02, mean=0.0):
    return _bits((mean + std * torch.randn(shape, generator=g, device="cuda", dtype=torch.float32)).to(torch.bfloat16))


def _stats(what, T, out_ids, ref_ids, logits, ref_logits, first, gaps, skip_col=None):
    """per-step comparison of the logits with the oracle's; printed, and appended to $LIA_PARITY_STATS"""
    steps = []
    for s, (g, r) in enumerate(zip(logits, ref_logits)):
        gb = _bits(g)
        keep = np.ones(gb.shape[1], bool)
        if skip_col is not None:
            keep[skip_col] = False                   # generate() suppresses EOS while fewer than min_new_tokens exist; the oracle has no such hook
        gf, rf = synth.bf16_bits_to_f32(gb[:, keep]), synth.bf16_bits_to_f32(r[:, keep])
        q = 2.0 ** (np.floor(np.log2(max(float(np.abs(rf).max()), 1e-30))) - 7)
        rows_same = (np.asarray(out_ids)[:, :T + s] == np.asarray(ref_ids)[:, :T + s]).all(axis=1)       # rows still on the oracle's sequence
        if not rows_same.any():
            break
        err = np.abs(gf - rf)[rows_same]
        steps.append({"step": s, "rows_compared": int(rows_same.sum()), "frac_bit_identical": float((gb[:, keep] == r[:, keep])[rows_same].mean()),
                      "max_err_quanta": float(err.max() / q), "frac_within_one_quantum": float((err <= q).mean()),
                      "min_top2_gap": gaps[s], "quantum": q})
    rec = {"what": what, "first_divergent_step": first, "steps": steps}
    print("\n" + what + f": first divergent step {first}")
    for st in steps:
        print("  step %d: %d rows, %.2f %% logits bit-identical, max err %.2f quanta, %.3f %% within one, smallest top-2 gap %.4f (quantum %.4f)"
              % (st["step"], st["rows_compared"], 100 * st["frac_bit_identical"], st["max_err_quanta"], 100 * st["frac_within_one_quantum"],
                 st["min_top2_gap"], st["quantum"]))
    path = os.environ.get("LIA_PARITY_STATS")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")
    return steps


def _check_logits(steps, what, max_quanta=4.0, within_one=0.95):
    """on the rows that still follow the oracle's sequence the logits stay within a few quanta of the largest logit (the same
    bound as a whole layer in test_gpu_fullsize_oracle.py, one more GEMM deep: the 7168 / 4096-term head)"""
    for st in steps:
        assert st["max_err_quanta"] <= max_quanta and st["frac_within_one_quantum"] >= within_one, (what, st)


# ---------------------------------------------------------------------------------------------------------------------
# OPT-30B width: 3 layers + vocab 50272
# ---------------------------------------------------------------------------------------------------------------------
OPT30 = dict(H=7168, heads=56, F=28672, L=3, vocab=50272, max_pos=2048, B=4, T=32, new=8)


@pytest.fixture(scope="module")
def opt30_stack():
    import torch
    from lia_amd.model import LiaOPTModel, OPTShape
    c = OPT30
    g = torch.Generator(device="cuda").manual_seed(4242)
    H, F = c["H"], c["F"]
    m = {"embed_tokens": _draw(torch, g, (c["vocab"], H)), "embed_positions": _draw(torch, g, (c["max_pos"] + 2, H)),
         "final_ln_w": _draw(torch, g, (H,), 0.1, 1.0), "final_ln_b": _draw(torch, g, (H,), 0.05), "layers": []}
    for _ in range(c["L"]):
        lw = {}
        for n, shp in synth.layer_shapes(H, F).items():
            if n in ("ln1_w", "ln2_w"):
                lw[n] = _draw(torch, g, shp, 0.1, 1.0)
            elif n.endswith("_b"):
                lw[n] = _draw(torch, g, shp, 0.05)
            else:
                lw[n] = _draw(torch, g, shp, 0.02)
        m["layers"].append(lw)
    torch.cuda.synchronize()
    shape = OPTShape("opt30b-3layers", H, c["heads"], F, c["L"], vocab=c["vocab"], max_pos=c["max_pos"])
    model = LiaOPTModel.from_numpy(shape, m)
    rs = np.random.RandomState(77)
    ids = rs.randint(4, c["vocab"], size=(c["B"], c["T"])).astype(np.int64)          # four DIFFERENT rows
    ids[:, 0] = 2
    cache = {}
    yield dict(m=m, model=model, ids=ids, c=c, oracle_runs=cache)
    model.close()


def _oracle_run(stack, oracle, key):
    """oracle.generate per (prefill policy, decode policy, gpu%) -- its arithmetic depends on them -- once per module"""
    from lia_amd import hostinfo
    if key not in stack["oracle_runs"]:
        oracle.lib().lia_oracle_set_threads(hostinfo.usable_cpus())
        oracle.lib().lia_oracle_set_fast(0)
        c = stack["c"]
        stack["oracle_runs"][key] = oracle.generate(stack["m"], stack["ids"], c["new"], c["heads"], key[0], key[1], key[2], return_logits=True)
    return stack["oracle_runs"][key]


@pytest.mark.parametrize("name,pp,dp,gpu,fmt", [("0/2, 1 resident + 2 streamed, raw wire", 0, 2, 34, "raw"),
                                               ("0/2, 1 resident + 2 streamed, pack10 wire", 0, 2, 34, "pack10"),
                                               ("3/3 all-resident", 3, 3, 100, "raw"),
                                               ("1/1 all layers on the host cores", 1, 1, 0, "raw")])
def test_opt30b_width_generate_vs_oracle(opt30_stack, oracle, name, pp, dp, gpu, fmt):
    import torch
    from lia_amd.generation import generate
    from lia_amd.scheduler import OffloadScheduler
    st, c = opt30_stack, opt30_stack["c"]
    model = st["model"]
    old = getattr(model, "_lia_scheduler", None)
    if old is not None:
        old.close()
    model._lia_scheduler = OffloadScheduler(model, wire=fmt)
    out, _, logits = generate(model, torch.from_numpy(st["ids"]), max_new_tokens=c["new"], min_new_tokens=c["new"], return_logits=True,
                              prefill_policy=pp, decoding_policy=dp, gpu_percentage=gpu, pin_weight=True)
    n_gpu = int(c["L"] * gpu / 100)
    if fmt == "pack10":
        assert all(s.packed == 10 for s in model.layers[n_gpu:]), [s.packed for s in model.layers]
    ref_ids, _, ref_logits = _oracle_run(st, oracle, (pp, dp, gpu))
    what = f"OPT-30B width x {c['L']} layers, vocab {c['vocab']}, B {c['B']} x T {c['T']}, {name}"
    first, gaps = ids_equal_or_near_tie(out.numpy(), ref_ids, ref_logits, c["T"], what)
    steps = _stats(what, c["T"], out.numpy(), ref_ids, logits, ref_logits, first, gaps, skip_col=2)
    _check_logits(steps, what)
    model._lia_scheduler.close()
    model._lia_scheduler = None


def test_opt30b_width_wire_formats_and_policies_agree_bit_for_bit(opt30_stack):
    """the streamed run is the all-resident run's arithmetic (3/3 vs 3/3 at gpu% 34: only the weights travel), and pack10 on the
    wire is lossless: same ids AND same logits bits as raw"""
    import torch
    from lia_amd.generation import generate
    from lia_amd.scheduler import OffloadScheduler
    st, c = opt30_stack, opt30_stack["c"]
    model = st["model"]
    runs = {}
    for key, gpu, fmt in (("resident", 100, "raw"), ("streamed raw", 34, "raw"), ("streamed pack10", 34, "pack10")):
        model._lia_scheduler = OffloadScheduler(model, wire=fmt)
        out, _, logits = generate(model, torch.from_numpy(st["ids"]), max_new_tokens=4, min_new_tokens=4, return_logits=True,
                                  prefill_policy=3, decoding_policy=3, gpu_percentage=gpu, pin_weight=True)
        runs[key] = (out.numpy().copy(), [_bits(l) for l in logits])
        model._lia_scheduler.close()
        model._lia_scheduler = None
    for key in ("streamed raw", "streamed pack10"):
        assert (runs[key][0] == runs["resident"][0]).all(), key
        for s, (a, b) in enumerate(zip(runs[key][1], runs["resident"][1])):
            assert (a == b).all(), f"{key}: logits of step {s} differ from the all-resident run in {(a != b).sum()} places"


# ---------------------------------------------------------------------------------------------------------------------
# Llama-3-8B width: 2 layers + vocab 128256
# ---------------------------------------------------------------------------------------------------------------------
LLAMA8 = dict(H=4096, heads=32, kvh=8, F=14336, L=2, vocab=128256, B=4, T=32, new=8, theta=500000.0)


@pytest.fixture(scope="module")
def llama8_stack():
    import torch
    from lia_amd.llama import LiaLlamaModel, LlamaShape
    c = LLAMA8
    g = torch.Generator(device="cuda").manual_seed(808)
    H, F, d = c["H"], c["F"], c["H"] // c["heads"]
    KD = c["kvh"] * d
    m = {"embed_tokens": _draw(torch, g, (c["vocab"], H)), "lm_head": _draw(torch, g, (c["vocab"], H)),
         "final_norm_w": _draw(torch, g, (H,), 0.1, 1.0), "layers": []}
    shapes = {"in_norm_w": (H,), "q_w": (H, H), "k_w": (KD, H), "v_w": (KD, H), "o_w": (H, H), "post_norm_w": (H,), "gate_w": (F, H),
              "up_w": (F, H), "down_w": (H, F)}
    for _ in range(c["L"]):
        m["layers"].append({n: (_draw(torch, g, shapes[n], 0.1, 1.0) if n.endswith("norm_w") else _draw(torch, g, shapes[n])) for n in synth.LLAMA_TENSORS})
    torch.cuda.synchronize()
    shape = LlamaShape("llama3-8b-2layers", H, c["heads"], c["kvh"], F, c["L"], c["vocab"], max_pos=256, rope_theta=c["theta"])
    model = LiaLlamaModel.from_numpy(shape, m)
    rs = np.random.RandomState(78)
    ids = rs.randint(4, c["vocab"], size=(c["B"], c["T"])).astype(np.int64)
    yield dict(m=m, model=model, ids=ids, c=c, ref=None)
    model.close()


@pytest.mark.parametrize("name,gpu,fmt", [("all-resident", 100, "raw"), ("1 resident + 1 streamed, raw wire", 50, "raw"),
                                          ("1 resident + 1 streamed, pack10 wire", 50, "pack10")])
def test_llama3_8b_width_generate_vs_oracle(llama8_stack, oracle, name, gpu, fmt, monkeypatch):
    import torch
    from lia_amd import hostinfo
    from lia_amd.generation import generate
    monkeypatch.setenv("LIA_STREAM_FORMAT", fmt)
    st, c = llama8_stack, llama8_stack["c"]
    model = st["model"]
    old = getattr(model, "_lia_scheduler", None)
    if old is not None:
        old.close()
        model._lia_scheduler = None
    out, _, logits = generate(model, torch.from_numpy(st["ids"]), max_new_tokens=c["new"], min_new_tokens=c["new"], return_logits=True,
                              gpu_percentage=gpu, pin_weight=True)
    if st["ref"] is None:
        oracle.lib().lia_oracle_set_threads(hostinfo.usable_cpus())
        oracle.lib().lia_oracle_set_fast(0)
        st["ref"] = oracle.llama_generate(st["m"], st["ids"], c["new"], c["heads"], c["kvh"], c["theta"], return_logits=True)
    ref_ids, _, ref_logits = st["ref"]
    what = f"Llama-3-8B width x {c['L']} layers, vocab {c['vocab']}, B {c['B']} x T {c['T']}, {name}"
    first, gaps = ids_equal_or_near_tie(out.numpy(), ref_ids, ref_logits, c["T"], what)
    steps = _stats(what, c["T"], out.numpy(), ref_ids, logits, ref_logits, first, gaps, skip_col=None)
    _check_logits(steps, what)
    if fmt == "pack10":
        assert model.layers[1].packed == 10
    model._lia_scheduler.close()
    model._lia_scheduler = None


# ---------------------------------------------------------------------------------------------------------------------
# configs[2]'s own weights at its own width: one OPT-175B-shaped layer, every parameter ~ U[0,1)
# ---------------------------------------------------------------------------------------------------------------------
def test_opt175b_shape_uniform01_decode_layer_vs_oracle(oracle):
    """The reference's dummy recipe (opt-weight-gen.py:61-62: rand_like on every parameter, LayerNorm included) at 12288 / 96 /
    49152: fc2 adds 49152 positive products (sums of ~1e7), q.k scores reach the tens of thousands and the softmax is one-hot.
    The prefill (B 32 x T 16, policy 3) fills the cache on the GPU; the decode step at S = 17 is then compared with the oracle
    under policy 3 (device cache, GPU attention) and policy 2 (host cache, host attention) ON THE SAME CACHE CONTENTS.  Bound as in
    the miniature (test_gpu_configs.py): relative -- 2 bf16 ulps of the value -- for 99.5 % of the elements; a saturated softmax
    that picks the other key of a near-tie moves a whole head's output, hence the looser maximum."""
    import torch
    from lia_amd import _native as N, hostinfo, ops
    from lia_amd.model import OPTShape, draw_layer
    H, heads, F, B, T = 12288, 96, 49152, 32, 16
    d = H // heads
    desc = ops.make_desc(H, heads, F)
    offs, total = ops.pack_offsets(desc)
    flat = draw_layer(OPTShape("opt-175b-dummy", H, heads, F, 1), offs, total, li=0, seed=5, init="uniform01")
    torch.cuda.synchronize()
    host = _bits(flat)
    dims = {"q_w": (H, H), "k_w": (H, H), "v_w": (H, H), "out_w": (H, H), "fc1_w": (F, H), "fc2_w": (H, F)}
    W = {}
    for i, n in enumerate(ops.LAYER_TENSORS):
        shp = dims.get(n, (F,) if n == "fc1_b" else (H,))
        W[n] = host[offs[i] // 2: offs[i] // 2 + int(np.prod(shp))].reshape(shp)
    assert 0.49 < synth.bf16_bits_to_f32(W["fc2_w"][:64]).mean() < 0.51 and synth.bf16_bits_to_f32(W["ln1_b"]).min() >= 0.0
    g = torch.Generator(device="cuda").manual_seed(6)
    # the hidden state a dummy model's embeddings produce: token + position rows, both U[0,1)
    x = (torch.rand((B, T, H), generator=g, device="cuda") + torch.rand((B, T, H), generator=g, device="cuda")).to(torch.bfloat16)
    xs = (torch.rand((B, 1, H), generator=g, device="cuda") + torch.rand((B, 1, H), generator=g, device="cuda")).to(torch.bfloat16)
    torch.cuda.synchronize()
    ctx = ops.Context(0, ops.workspace_bytes(desc, B * T))
    ctx.set_host_threads(hostinfo.usable_cpus())
    wptrs = ops.weight_ptr_array(flat.data_ptr(), offs)
    kc = torch.zeros((T + 2, B, heads, d), dtype=torch.bfloat16, device="cuda")
    vc = torch.zeros_like(kc)
    torch.cuda.synchronize()
    kv = N.KV(kc.data_ptr(), vc.data_ptr(), T + 2, B, 1)
    y = torch.empty_like(x)
    ctx.layer_forward(desc, 3, wptrs, x, y, kv, B, T, 0)                # prefill: rows [0, T) of the cache
    ctx.synchronize()
    k0, v0 = _bits(kc).copy(), _bits(vc).copy()
    assert np.isfinite(synth.bf16_bits_to_f32(_bits(y))).all() and float(np.abs(synth.bf16_bits_to_f32(_bits(y))).max()) > 1e5

    oracle.lib().lia_oracle_set_threads(hostinfo.usable_cpus())
    oracle.lib().lia_oracle_set_fast(0)

    def close(got, ref, what):
        a, b = synth.bf16_bits_to_f32(got), synth.bf16_bits_to_f32(ref)
        ratio = np.abs(a - b) / (0.02 + 0.016 * np.abs(b))
        print(f"\n{what}: {100 * (got == ref).mean():.2f} % bit-identical, |ref| up to {np.abs(b).max():.3g}, error / (2 ulp): q99.5 {np.quantile(ratio, 0.995):.2f}, "
              f"max {ratio.max():.2f}")
        assert np.isfinite(a).all() and np.quantile(ratio, 0.995) <= 1.0 and ratio.max() <= 16.0, what
        return float((got == ref).mean())

    for policy in (3, 2):
        okc, ovc = k0.copy(), v0.copy()
        if policy == 2:
            hk = torch.from_numpy(k0.view(np.int16).copy()).view(torch.bfloat16).pin_memory()
            hv = torch.from_numpy(v0.view(np.int16).copy()).view(torch.bfloat16).pin_memory()
            kvp = N.KV(hk.data_ptr(), hv.data_ptr(), T + 2, B, 0)
        else:
            hk, hv = kc, vc
            kvp = kv
        ys = torch.empty_like(xs)
        ctx.layer_forward(desc, policy, wptrs, xs, ys, kvp, B, 1, T)
        ctx.synchronize()
        ref = oracle.layer_forward(policy, W, _bits(xs), okc, ovc, T, heads)
        close(to_bits(ys), ref, f"OPT-175B shape, U[0,1) weights, decode B {B} S {T + 1}, policy {policy}")
        close(to_bits(hk)[T], okc[T], f"policy {policy}: new K row")
        close(to_bits(hv)[T], ovc[T], f"policy {policy}: new V row")
        if policy == 3:
            kc[T].zero_()
            vc[T].zero_()
            torch.cuda.synchronize()
    ctx.close()
    del flat
    torch.cuda.empty_cache()
