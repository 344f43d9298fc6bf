"""Per-op breakdown of a full-size decode layer against the oracle (GPU box): which sub-op carries a deviation.
    python tools/fullsize_diag.py [opt-175b|opt-30b]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "isca-2025-lia_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import lia_oracle as orc  # noqa: E402
import synth  # noqa: E402
from lia_amd import hostinfo, ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "opt-175b"
H, heads, F, B = {"opt-30b": (7168, 56, 28672, 64), "opt-175b": (12288, 96, 49152, 32)}[name]
orc.lib().lia_oracle_set_threads(hostinfo.usable_cpus())
orc.lib().lia_oracle_set_fast(0)
f32 = synth.bf16_bits_to_f32


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def rnd(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    t = (scale * torch.randn(shape, generator=g, device="cuda")).to(torch.bfloat16).contiguous()
    torch.cuda.synchronize()
    return t


def report(what, got, ref):
    a, b = f32(got), f32(ref)
    err = np.abs(a - b)
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(b), 1e-30))) - 7)
    r = err / ulp
    print(f"{what:38s} identical {100 * (got == ref).mean():6.2f} %  max|err| {err.max():.4g}  max err/ulp {r.max():.1f}  "
          f">1ulp {100 * (r > 1.01).mean():.3f} %  >2ulp {100 * (r > 2.01).mean():.4f} %  max|ref| {np.abs(b).max():.3g}", flush=True)


ctx = ops.Context(0, 8 * 64 * max(3 * H, F) * 4 + (1 << 20))
for (N, K, relu, res, xs) in ((3 * H, H, False, False, 1.0), (H, H, False, True, 0.3), (F, H, True, False, 1.0), (H, F, False, True, 1.5)):
    w, b = rnd((N, K), N + K, 0.02), rnd((N,), 5, 0.1)
    x = rnd((B, K), 7, xs)
    if relu is False and K == F:
        x = torch.relu(x)
    r = rnd((B, N), 9) if res else None
    for split in (0, 1):
        y = ctx.linear(x, w, b, r, relu=relu, split_k=split)
        ctx.synchronize()
        ref = orc.linear(bits(x), bits(w), bits(b), None if r is None else bits(r), relu=relu)
        report(f"linear M={B} N={N} K={K} split={split}", bits(y), ref)
    # the same GEMM without bias / residual: the raw accumulation
    y = ctx.linear(x, w)
    ctx.synchronize()
    report(f"  (no bias/res) N={N} K={K}", bits(y), orc.linear(bits(x), bits(w)))
    del w
d = H // heads
T = 256
kc, vc = rnd((T + 2, B, heads, d), 21), rnd((T + 2, B, heads, d), 22)
for qs in (1.0, 2.2):
    q = rnd((B, 1, H), 23, qs)
    out = ctx.attention(q, kc, vc, T + 1, heads)
    ctx.synchronize()
    report(f"attention decode S={T + 1} q scale {qs}", bits(out), orc.attention(bits(q), bits(kc), bits(vc), T + 1, heads, True))
x = rnd((B, H), 31, 3.0)
g, bb = rnd((H,), 32), rnd((H,), 33, 0.1)
report("layernorm", bits(ctx.layernorm(x, g, bb)), orc.layernorm(bits(x), bits(g), bits(bb)))
ctx.close()
