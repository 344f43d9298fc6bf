"""-m gpu: the persistent decode chain (csrc/lia_chain.hip) against the per-op route, bit for bit.

With lia_set_fused_decode(1), lia_decode_layers / lia_llama_decode_layers run the resident layers of a decode step as, per layer,
one attention launch and ONE persistent launch (out-proj, norm, MLP, the next layer's norm and q|k|v projection, with the split-K
combines inside).  The
arithmetic is the per-op route's: same K chunks into the same accumulators, slabs added slice 0, 1, ..., values finished by the
same device functions.  With lia_gemm_set_split_policy(1) the per-op GEMMs cut K into the chain's slices, so the two routes
must agree on EVERY bit of the logits, the ids and the K/V rows they append -- any race in the in-launch hand-offs (grid
barriers, write-through stores, LDS-DMA rings) shows up as a difference.  The per-op route itself -- the default: the chain
measured 4-9 % slower per step and is opt-in (LIA_FUSED_DECODE=1, lia_set_fused_decode(1)) -- is pinned to the oracle and to the
reference goldens by the other -m gpu tests.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(t):
    import torch
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def _llama_run(shape, B, T, steps, fused, seed=3):
    import torch
    from lia_amd import _native as N
    from lia_amd.llama import LiaLlamaModel, LlamaKVState, LlamaScheduler
    lib = N.lib()
    lib.lia_set_fused_decode(int(fused))
    lib.lia_gemm_set_split_policy(1)
    lib.lia_gemm_set_engine(0)                  # the per-op side on lia_gemm_skinny2_kernel: a kernel of its own to compare with
    n0 = lib.lia_chain_launch_count()
    try:
        model = LiaLlamaModel.random_init(shape, seed=seed)
        sched = LlamaScheduler(model)
        kv = LlamaKVState(model, B, T + steps + 1)
        g = torch.Generator().manual_seed(seed)
        ids = torch.randint(4, shape.vocab, (B, T), generator=g)
        logits, nxt = sched.forward(ids, kv)
        outs = [(_bits(logits), nxt.cpu().numpy().copy())]
        for _ in range(steps):
            logits, nxt = sched.forward(nxt.view(B, 1).cpu(), kv)
            outs.append((_bits(logits), nxt.cpu().numpy().copy()))
        caches = [(_bits(k[:kv.len]), _bits(v[:kv.len])) for k, v in kv.tensors]
        sched.close()
        model.close()
        return outs, caches, lib.lia_chain_launch_count() - n0
    finally:
        lib.lia_set_fused_decode(0)
        lib.lia_gemm_set_split_policy(0)
        lib.lia_gemm_set_engine(0)


def _opt_run(shape, B, T, steps, fused, seed=5):
    import torch
    from lia_amd import _native as N
    from lia_amd.model import LiaOPTModel
    from lia_amd.scheduler import KVState, OffloadScheduler
    lib = N.lib()
    lib.lia_set_fused_decode(int(fused))
    lib.lia_gemm_set_split_policy(1)
    lib.lia_gemm_set_engine(0)
    n0 = lib.lia_chain_launch_count()
    try:
        model = LiaOPTModel.random_init(shape, seed=seed, n_gpu_layers=shape.layers)
        # non-trivial biases and LayerNorm parameters (random_init leaves them at 0 / 1): the combines' bias / affine paths count
        from lia_amd import ops
        g = torch.Generator(device="cuda").manual_seed(seed + 1)
        for st in model.layers:
            base = st.device_ptr()
            for i, n in enumerate(ops.LAYER_TENSORS):
                if n.endswith("_b") or n in ("ln1_w", "ln2_w"):
                    k = shape.ffn if n == "fc1_b" else shape.hidden
                    vals = ((1.0 if n.endswith("_w") else 0.0) + 0.1 * torch.randn(k, generator=g, device="cuda")).to(torch.bfloat16)
                    N.check(lib.lia_blit(base + model.offsets[i], vals.data_ptr(), k * 2, None), "lia_blit")
                    torch.cuda.synchronize()
        torch.cuda.synchronize()
        sched = OffloadScheduler(model)
        kv = KVState(model, shape.layers, B, T + steps + 1)
        g2 = torch.Generator().manual_seed(seed)
        ids = torch.randint(4, shape.vocab, (B, T), generator=g2)
        flags = dict(prefill_policy=0, decoding_policy=2, gpu_percentage=100, pin_weight=True)
        logits, nxt = sched.forward(ids, kv, **flags)
        outs = [(_bits(logits), nxt.cpu().numpy().copy())]
        for _ in range(steps):
            logits, nxt = sched.forward(nxt.view(B, 1).cpu(), kv, **flags)
            outs.append((_bits(logits), nxt.cpu().numpy().copy()))
        caches = [(_bits(k[:kv.len]), _bits(v[:kv.len])) for k, v in kv.tensors]
        sched.close()
        model.close()
        return outs, caches, lib.lia_chain_launch_count() - n0
    finally:
        lib.lia_set_fused_decode(0)
        lib.lia_gemm_set_split_policy(0)
        lib.lia_gemm_set_engine(0)


def _same(a, b, what):
    (oa, ca, na), (ob, cb, nb) = a, b
    assert na > 0 and nb == 0, f"{what}: chain launches fused {na} / per-op {nb} -- the routes were not the ones under test"
    for s, ((la, ia), (lb, ib)) in enumerate(zip(oa, ob)):
        bad = int((la != lb).sum())
        assert bad == 0, f"{what}: step {s}: {bad} / {la.size} logits differ between the chain and the per-op route"
        assert (ia == ib).all(), f"{what}: step {s}: ids differ"
    for li, ((ka, va), (kb, vb)) in enumerate(zip(ca, cb)):
        assert (ka == kb).all() and (va == vb).all(), f"{what}: K/V rows of layer {li} differ"


LLAMA_CASES = {
    # name: (hidden, heads, kv_heads, ffn, layers, vocab, B, T, steps)
    "small_b128": (512, 4, 2, 1024, 3, 1024, 128, 8, 3),
    "small_b64": (512, 4, 4, 1536, 2, 1024, 64, 8, 3),
    "small_b20": (256, 2, 1, 512, 3, 512, 20, 6, 2),             # rows not a multiple of 16: clamped x rows, masked stores
    "llama3_8b_2layers_b128": (4096, 32, 8, 14336, 2, 4096, 128, 16, 2),
}


@pytest.mark.parametrize("case", sorted(LLAMA_CASES))
def test_llama_chain_bit_identical_to_per_op(case):
    from lia_amd.llama import LlamaShape
    H, heads, kvh, F, L, vocab, B, T, steps = LLAMA_CASES[case]
    shape = LlamaShape(case, H, heads, kvh, F, L, vocab, max_pos=64)
    a = _llama_run(shape, B, T, steps, fused=True)
    b = _llama_run(shape, B, T, steps, fused=False)
    _same(a, b, f"llama {case}")


OPT_CASES = {
    # name: (hidden, heads, ffn, layers, vocab, B, T, steps)
    "small_b64": (512, 4, 2048, 3, 1024, 64, 8, 3),
    "small_b128": (256, 4, 1024, 2, 512, 128, 8, 2),
    "small_b8": (256, 2, 1024, 3, 512, 8, 6, 2),
    "opt30b_2layers_b64": (7168, 56, 28672, 2, 4096, 64, 16, 2),
}


@pytest.mark.parametrize("case", sorted(OPT_CASES))
def test_opt_chain_bit_identical_to_per_op(case):
    from lia_amd.model import OPTShape
    H, heads, F, L, vocab, B, T, steps = OPT_CASES[case]
    shape = OPTShape(case, H, heads, F, L, vocab=vocab, max_pos=64)
    a = _opt_run(shape, B, T, steps, fused=True)
    b = _opt_run(shape, B, T, steps, fused=False)
    _same(a, b, f"opt {case}")


@pytest.mark.parametrize("M,N,K", [(4, 128256, 4096), (64, 50272, 7168), (128, 32768, 1024), (20, 4096, 512)])
def test_chain_engine_many_items_per_workgroup(M, N, K):
    """one GEMM with more work items than CUs (lm_head-sized N: a workgroup walks several items, its rings running on across
    them) through the chain kernel as GEMM engine, against lia_gemm_skinny2_kernel with the same K slices: same bits"""
    import torch
    from lia_amd import _native as N_, ops
    lib = N_.lib()
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = (torch.randn((M, K), generator=g, device="cuda")).to(torch.bfloat16)
    w = (0.02 * torch.randn((N, K), generator=g, device="cuda")).to(torch.bfloat16)
    bias = (0.1 * torch.randn((N,), generator=g, device="cuda")).to(torch.bfloat16)
    torch.cuda.synchronize()          # torch fills the inputs on ITS stream; the context's stream is not ordered behind it
    ctx = ops.Context(0, 8 * M * N * 4 + (1 << 20))
    try:
        lib.lia_gemm_set_split_policy(1)
        outs = []
        for engine in (1, 0):
            lib.lia_gemm_set_engine(engine)
            n0 = lib.lia_gemm_chain_engine_count()
            y = ctx.linear(x, w, bias=bias, relu=True)
            ctx.synchronize()
            assert (lib.lia_gemm_chain_engine_count() - n0 > 0) == (engine == 1)
            outs.append(_bits(y))
        bad = int((outs[0] != outs[1]).sum())
        assert bad == 0, f"{bad} / {outs[0].size} values differ between the chain engine and skinny2"
    finally:
        lib.lia_gemm_set_split_policy(0)
        lib.lia_gemm_set_engine(0)
        ctx.close()


def test_chain_repeatable_under_many_launches():
    """the same decode step 300 times over (> the 256 barrier-counter blocks of a context: the ring wraps and is re-zeroed in
    stream order): every repetition gives the bits of the first"""
    import torch
    from lia_amd import _native as N
    from lia_amd.llama import LiaLlamaModel, LlamaKVState, LlamaScheduler, LlamaShape
    shape = LlamaShape("rep", 512, 4, 2, 1024, 2, 1024, max_pos=64)
    N.lib().lia_set_fused_decode(1)
    model = LiaLlamaModel.random_init(shape, seed=9)
    sched = LlamaScheduler(model)
    B, T = 128, 8
    kv = LlamaKVState(model, B, T + 2)
    ids = torch.randint(4, shape.vocab, (B, T), generator=torch.Generator().manual_seed(1))
    _, nxt = sched.forward(ids, kv)
    first = None
    n0 = N.lib().lia_chain_launch_count()
    for _ in range(300):
        kv.len = T                                   # the same step again: same position, same cache prefix
        logits, _ = sched.forward(nxt.view(B, 1).cpu(), kv)
        bits = _bits(logits)
        if first is None:
            first = bits
        assert (bits == first).all()
    N.lib().lia_set_fused_decode(0)
    assert N.lib().lia_chain_launch_count() - n0 >= 300 * 3
    sched.close()
    model.close()


def test_fused_decode_environment_switch(tmp_path):
    """LIA_FUSED_DECODE=1 (read once when the library loads, hence a child process): the resident layers of a decode step take the
    persistent-chain route -- chain launches are counted -- and the ids are those of the default per-op route"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import json, sys
sys.path[:0] = [%r, %r]
import torch
from lia_amd import _native as N
from lia_amd.generation import generate
from lia_amd.llama import LiaLlamaModel, LlamaShape
shape = LlamaShape("env", 512, 4, 2, 1024, 2, 1024, max_pos=64)
model = LiaLlamaModel.random_init(shape, seed=4)
ids = torch.randint(4, 1024, (16, 8), generator=torch.Generator().manual_seed(2))
out = generate(model, ids, max_new_tokens=4, min_new_tokens=4, gpu_percentage=100, pin_weight=True)
print(json.dumps({"ids": out.tolist(), "chain_launches": N.lib().lia_chain_launch_count()}))
''' % (os.path.join(root, "isca-2025-lia_amd"), root)
    res = {}
    for flag in ("1", "0"):
        env = dict(os.environ, LIA_FUSED_DECODE=flag)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[flag] = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["1"]["chain_launches"] > 0 and res["0"]["chain_launches"] == 0
    assert res["1"]["ids"] == res["0"]["ids"]
