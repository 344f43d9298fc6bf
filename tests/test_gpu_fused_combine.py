"""-m gpu: the split-K combines that also run the next op of the decode layer (LayerNorm / RMSNorm of the finished row, SiLU*up,
RoPE; lia_gemm.hip `LiaPost`, lia_ctx_chain_next_norm) give the SAME BITS as the route with one kernel per op.  The reference
launches every one of these as its own op (decoder.py:199-206, 268-276); the oracle parity of the ops themselves is in
test_gpu_ops.py / test_gpu_llama.py -- this file pins the fusion."""
import ctypes

import numpy as np
import pytest

import synth
from test_gpu_ops import _layer_setup, to_bits, to_dev

pytestmark = pytest.mark.gpu
LN, RMS, SILU, ROPE = 1, 2, 3, 4


@pytest.mark.parametrize("B,H,heads,F", [(8, 1024, 8, 4096), (64, 2048, 16, 8192), (33, 1536, 12, 6144), (1, 768, 12, 3072), (256, 1024, 8, 4096)])
def test_opt_decode_layers_fused_equals_unfused(B, H, heads, F):
    """two resident layers, three decode steps: LN2 in the out-proj combine, the next layer's LN1 chained into fc2's."""
    import torch
    from lia_amd import _native as N, ops
    ctx = ops.Context(0, 1 << 30)
    other = ops.Context(0, 0)                    # a second context of the process: its switch stays where it is (per-context options, r05)
    d, T, new = H // heads, 5, 3
    layers = [_layer_setup(torch, ops, synth.make_layer(11 + i, H, F, 0.05), H, heads, F) for i in range(2)]

    def run(fused):
        ctx.set_option(N.LIA_OPT_FUSE_COMBINE, 1 if fused else 0)
        kvs, keep = [], []
        for _ in layers:
            kc = torch.zeros((T + new, B, heads, d), dtype=torch.bfloat16, device="cuda")
            vc = torch.zeros_like(kc)
            keep.append((kc, vc))
            kvs.append(N.KV(kc.data_ptr(), vc.data_ptr(), T + new, B, 1))
        outs = []
        a = to_dev(torch, synth.make_hidden(3, B, T, H))
        b = torch.empty_like(a)
        for (desc, _, wp), kv in zip(layers, kvs):       # prefill fills the caches (tiled GEMMs: nothing to fuse)
            ctx.layer_forward(desc, 3, wp, a, b, kv, B, T, 0)
            a, b = b, a
        for s in range(new):
            a = to_dev(torch, synth.make_hidden(100 + s, B, 1, H))
            b = torch.empty_like(a)
            for i, ((desc, _, wp), kv) in enumerate(zip(layers, kvs)):
                if fused and i + 1 < len(layers):
                    nw = layers[i + 1][2]
                    ctx.chain_next_norm(nw[0], nw[1])
                ctx.layer_forward(desc, 3, wp, a, b, kv, B, 1, T + s)
                a, b = b, a
            ctx.synchronize()
            outs.append(to_bits(a).copy())
        return outs, [to_bits(k).copy() for k, _ in keep]

    n0 = ctx.fused_combines(LN)
    fused, fk = run(True)
    n1 = ctx.fused_combines(LN)
    plain, pk = run(False)
    assert ctx.fused_combines(LN) == n1, "the switch did not turn the fused combines off"
    assert other.fused_combines(LN) == 0 and other.fused_combines(99) == -1      # counters are the context's own
    with pytest.raises(ValueError):
        ctx.set_option(12345, 1)
    other.close()
    # (H = 768: the out-proj GEMM has 12 K-chunks and is not split -- its LN2 stays a kernel of its own; fc2's chained LN1 rides)
    want = 3 * new if H >= 1024 else new
    assert n1 - n0 >= want, f"only {n1 - n0} fused LayerNorm combines ran (expected LN2 x2 + one chained LN1 per step)"
    for s, (f, p) in enumerate(zip(fused, plain)):
        assert (f == p).all(), f"decode step {s}: {(f != p).sum()} of {f.size} values differ between the fused and the per-op route"
    for f, p in zip(fk, pk):
        assert (f == p).all()
    ctx.close()


def test_chain_hint_is_ignored_when_the_next_call_takes_another_input():
    """the promise of lia_ctx_chain_next_norm is checked, not trusted: a following call on a different x computes its own LN1"""
    import torch
    from lia_amd import _native as N, ops
    B, H, heads, F, T = 16, 1024, 8, 4096, 4
    ctx = ops.Context(0, 1 << 30)
    la = _layer_setup(torch, ops, synth.make_layer(21, H, F, 0.05), H, heads, F)
    lb = _layer_setup(torch, ops, synth.make_layer(22, H, F, 0.05), H, heads, F)
    d = H // heads

    def cache():
        kc = torch.zeros((T + 2, B, heads, d), dtype=torch.bfloat16, device="cuda")
        return kc, torch.zeros_like(kc)

    def step(chain, other_x):
        (ka, va), (kb, vb) = cache(), cache()
        kva, kvb = N.KV(ka.data_ptr(), va.data_ptr(), T + 2, B, 1), N.KV(kb.data_ptr(), vb.data_ptr(), T + 2, B, 1)
        x = to_dev(torch, synth.make_hidden(5, B, 1, H))
        y, z = torch.empty_like(x), torch.empty_like(x)
        if chain:
            ctx.chain_next_norm(lb[2][0], lb[2][1])
        ctx.layer_forward(la[0], 3, la[2], x, y, kva, B, 1, 0)
        src = to_dev(torch, synth.make_hidden(6, B, 1, H)) if other_x else y
        ctx.layer_forward(lb[0], 3, lb[2], src, z, kvb, B, 1, 0)
        ctx.synchronize()
        return to_bits(z).copy()

    assert (step(True, True) == step(False, True)).all()
    assert (step(True, False) == step(False, False)).all()
    ctx.close()


@pytest.mark.parametrize("B,H,heads,kvh,F,T", [(16, 1024, 8, 2, 2816, 6), (128, 2048, 16, 4, 5632, 6), (40, 1024, 8, 8, 3072, 6),
                                                (128, 1024, 8, 2, 2816, 9), (1, 1024, 8, 2, 2816, 6)])
def test_llama_layers_fused_equals_unfused(B, H, heads, kvh, F, T):
    """q|k|v in one GEMM with RoPE in its combine, RMSNorm in the o-proj combine, SiLU*up in the gate|up combine, the next
    layer's input RMSNorm chained into down-proj's -- and in the prefill SiLU*up in the tiled GEMM's epilogue (B x T = 96 rows:
    skinny; 240 / 768 rows: the 128 x 128 tiles; 1152 rows: the 256 x 256 tiles): the hidden states of the prefill and of every
    decode step and the post-RoPE K cache, bit for bit against the route with one kernel per op."""
    import torch
    from lia_amd import _native as N, ops
    from lia_amd.llama import LiaLlamaModel, LlamaShape, rope_tables
    lib = N.lib()
    new, L = 3, 2
    m = synth.make_llama_model(7, 64, H, heads, kvh, F, L, 0.05)
    shape = LlamaShape("t", H, heads, kvh, F, L, 64, max_pos=T + new + 4, rope_theta=10000.0)
    model = LiaLlamaModel.from_numpy(shape, m)
    model.place(L, True, False)
    d = H // heads
    ctx = ops.Context(0, max(lib.lia_llama_workspace_bytes(ctypes.byref(model.desc), B * T), 1 << 26))
    cos, sin = rope_tables(T + new + 4, d, shape.rope_theta)
    ws = [(ctypes.c_void_p * 9)(*[model.layers[i].device_ptr() + o for o in model.offsets]) for i in range(L)]

    def call(w, x, y, kv, Tn, pos0):
        N.check(lib.lia_llama_layer_forward(ctx.handle, ctypes.byref(model.desc), ctypes.byref(w), ctypes.c_void_p(x.data_ptr()),
                                            ctypes.c_void_p(y.data_ptr()), ctypes.byref(kv), ctypes.c_void_p(cos.data_ptr()),
                                            ctypes.c_void_p(sin.data_ptr()), B, Tn, pos0, 0, ctypes.c_void_p(ctx.stream)))

    def run(fused):
        ctx.set_option(N.LIA_OPT_FUSE_COMBINE, 1 if fused else 0)
        keep, kvs = [], []
        for _ in range(L):
            kc = torch.zeros((T + new, B, kvh, d), dtype=torch.bfloat16, device="cuda")
            vc = torch.zeros_like(kc)
            keep.append((kc, vc))
            kvs.append(N.KV(kc.data_ptr(), vc.data_ptr(), T + new, B, 1))
        a = to_dev(torch, synth.make_hidden(3, B, T, H))
        b = torch.empty_like(a)
        for w, kv in zip(ws, kvs):
            call(w, a, b, kv, T, 0)
            a, b = b, a
        ctx.synchronize()
        outs = [to_bits(a).copy()]                       # the prefill's hidden states
        for s in range(new):
            a = to_dev(torch, synth.make_hidden(100 + s, B, 1, H))
            b = torch.empty_like(a)
            for i, (w, kv) in enumerate(zip(ws, kvs)):
                if fused and i + 1 < L:
                    ctx.chain_next_norm(ws[i + 1][0])
                call(w, a, b, kv, 1, T + s)
                a, b = b, a
            ctx.synchronize()
            outs.append(to_bits(a).copy())
        return outs, [(to_bits(k).copy(), to_bits(v).copy()) for k, v in keep]

    before = [ctx.fused_combines(k) for k in (RMS, SILU, ROPE)]
    fused, fkv = run(True)
    after = [ctx.fused_combines(k) for k in (RMS, SILU, ROPE)]
    plain, pkv = run(False)
    ran = [a - b for a, b in zip(after, before)]
    assert ran[0] >= 3 * new and ran[1] >= 2 * new and ran[2] >= 2 * new, f"fused combines that ran (rmsnorm, silu, rope): {ran}"
    if B * T > 256:
        assert ran[1] >= 2 * new + L, f"the tiled gate|up epilogue did not write silu(gate) * up itself: {ran}"
    for s, (f, p) in enumerate(zip(fused, plain)):
        assert (f == p).all(), f"{'prefill' if s == 0 else 'decode step %d' % (s - 1)}: {(f != p).sum()} of {f.size} values differ"
    for (fk, fv), (pk, pv) in zip(fkv, pkv):
        assert (fk == pk).all() and (fv == pv).all()
    ctx.close()
    model.close()


def test_two_contexts_of_one_process_keep_their_own_switches():
    """include/lia_hip.h promises that contexts are independent: with the fused combines off in ONE of two live contexts the
    other still fuses (r04 had process-wide lia_gemm_set_* setters), and both give the same bits"""
    import torch
    from lia_amd import _native as N_, ops
    M, H, heads, F = 64, 1024, 8, 4096
    la = _layer_setup(torch, ops, synth.make_layer(41, H, F, 0.05), H, heads, F)
    a, b = ops.Context(0, 1 << 28), ops.Context(0, 1 << 28)
    b.set_option(N_.LIA_OPT_FUSE_COMBINE, 0)
    outs = []
    for ctx in (a, b, a):
        kc = torch.zeros((4, M, heads, H // heads), dtype=torch.bfloat16, device="cuda")
        vc = torch.zeros_like(kc)
        kv = N_.KV(kc.data_ptr(), vc.data_ptr(), 4, M, 1)
        xin = to_dev(torch, synth.make_hidden(8, M, 1, H))
        y = torch.empty_like(xin)
        ctx.layer_forward(la[0], 3, la[2], xin, y, kv, M, 1, 0)
        ctx.synchronize()
        outs.append(to_bits(y).copy())
    assert a.fused_combines(LN) >= 2 and b.fused_combines(LN) == 0
    assert (outs[0] == outs[1]).all() and (outs[0] == outs[2]).all()
    a.close()
    b.close()


def test_phased_256_tiles_equal_the_128_tiles_bit_for_bit():
    """the prefill GEMM's two tile shapes add the same products in the same order: M = 1024 rows x N = 1536 in one call run the phased
    256 x 256 kernel (lia_gemm_tiled256p_kernel; split_k = 1: one K slice), the first 256 output columns computed on their own
    (N = 256 < 512) run the 128 x 128 kernel -- same bits; both within bf16 rounding of an fp32 reference.  r06: with 24 tiles on 256
    CUs the launcher's own choice splits K over fp32 slabs (lia_gemm_tiled256p_kernel<0, true> + the combine kernel): another
    summation order, so >= 99.9 % of the outputs identical to the one-slice result and none further than one quantum."""
    import torch
    from lia_amd import ops
    M, N, K = 1024, 1536, 1024
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = (0.03 * torch.randn((N, K), generator=g, device="cuda")).to(torch.bfloat16)
    bias = (0.1 * torch.randn((N,), generator=g, device="cuda")).to(torch.bfloat16)
    torch.cuda.synchronize()
    ctx = ops.Context(0, 1 << 26)
    try:
        y = ctx.linear(x, w, bias=bias, relu=True, split_k=1)
        small = ctx.linear(x, w[:256].contiguous(), bias=bias[:256].contiguous(), relu=True, split_k=1)
        halves = [ctx.linear(x[i * 512:(i + 1) * 512].contiguous(), w, bias=bias, relu=True, split_k=1) for i in range(2)]
        ysplit = ctx.linear(x, w, bias=bias, relu=True)                   # the launcher's own slice count
        ctx.synchronize()
        want = torch.relu((x.float() @ w.float().T + bias.float()).to(torch.bfloat16).float())
        assert float((y.float() - want).abs().max()) <= 0.02 * float(want.abs().max())
        got = to_bits(y)
        assert (got[:, :256] == to_bits(small)).all(), "the 256^2 and the 128^2 tiles differ"
        assert (got == np.concatenate([to_bits(h) for h in halves], 0)).all(), "one call of 1024 rows and two of 512 differ"
        gs = to_bits(ysplit)
        err = (ysplit.float() - y.float()).abs()
        q = 2.0 ** (np.floor(np.log2(float(want.abs().max()))) - 7)
        assert float((gs == got).mean()) >= 0.999 and float(err.max()) <= q, (float((gs == got).mean()), float(err.max()), q)
    finally:
        ctx.close()
