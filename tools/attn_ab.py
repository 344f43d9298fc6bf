"""A/B of the two d = 128 prefill attention kernels on the GPU box: bit-for-bit agreement and time.
usage: python tools/attn_ab.py [B T heads]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))
import torch  # noqa: E402
from lia_amd import _native as N, ops  # noqa: E402

L = N.lib()
L.lia_attn_set_prefill_variant.argtypes = [__import__("ctypes").c_int]
cases = [(2, 8, 4), (1, 17, 4), (3, 40, 2), (2, 256, 2), (1, 300, 3), (64, 256, 56), (4, 1000, 8), (128, 1024, 32)]
if len(sys.argv) == 4:
    cases = [tuple(int(v) for v in sys.argv[1:4])]
ctx = ops.Context(0, 1 << 20)
for B, T, heads in cases:
    g = torch.Generator(device="cuda").manual_seed(B * 131 + T)
    H = heads * 128
    q = torch.randn((B, T, H), generator=g, device="cuda").to(torch.bfloat16)
    k = torch.randn((T, B, heads, 128), generator=g, device="cuda").to(torch.bfloat16)
    v = torch.randn((T, B, heads, 128), generator=g, device="cuda").to(torch.bfloat16)
    torch.cuda.synchronize()
    outs, times = [], []
    for variant in (1, 2):
        L.lia_attn_set_prefill_variant(variant)
        o = ctx.attention(q, k, v, T, heads)
        ctx.synchronize()
        t0 = time.time()
        for _ in range(3):
            o = ctx.attention(q, k, v, T, heads)
        ctx.synchronize()
        times.append((time.time() - t0) / 3)
        outs.append(o.view(torch.int16).clone())
    bad = int((outs[0] != outs[1]).sum())
    flops = 4 * B * heads * T * T * 128 / 2
    print(f"B={B} T={T} heads={heads}: mismatches {bad} / {outs[0].numel()}   v1 {times[0] * 1e3:.3f} ms  v2 {times[1] * 1e3:.3f} ms "
          f"({flops / times[1] / 1e12:.0f} TF/s causal)")
