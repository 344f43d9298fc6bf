"""Randomised parity sweep of lia_attention (prefill S == T with the causal mask, decode T == 1 over a cache) against a plain
PyTorch restatement with the reference's rounding points (attentions.py:443-536): q*d^-0.5 -> bf16, q.k -> bf16, softmax in
fp32 -> bf16, p.v -> bf16.  usage: python tools/attn_fuzz.py [n_cases] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))
import torch  # noqa: E402
from lia_amd import ops  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = ops.Context(0, 1 << 20)
bf = torch.bfloat16


def ref(q, k, v, causal):
    # q [B,T,h,d], k/v [S,B,h,d] -> [B,T,h,d]
    d = q.shape[-1]
    qs = (q.float() * d ** -0.5).to(bf).float().permute(0, 2, 1, 3)          # [B,h,T,d]
    kk = k.float().permute(1, 2, 0, 3)                                      # [B,h,S,d]
    vv = v.float().permute(1, 2, 0, 3)
    s = (qs @ kk.transpose(-1, -2)).to(bf).float()
    if causal:
        T, S = s.shape[-2:]
        s = s.masked_fill(torch.triu(torch.ones(T, S, dtype=torch.bool, device=s.device), 1), float("-inf"))
    p = torch.softmax(s, dim=-1).to(bf).float()
    return (p @ vv).to(bf).permute(0, 2, 1, 3)


for case in range(n_cases):
    d = rng.choice([64, 128, 128])
    heads = rng.choice([1, 2, 3, 4, 7, 8])
    B = rng.choice([1, 2, 3, 5, 8])
    prefill = rng.random() < 0.5
    S = rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, 256, 257, 300, 511, 700])
    T = S if prefill else 1
    g = torch.Generator(device="cuda").manual_seed(1000 + case)
    q = torch.randn((B, T, heads * d), generator=g, device="cuda").to(bf)
    smax = S + rng.choice([0, 1, 5])
    # the cache may be wider than the batch (a minibatch of a larger cache: rows b0 .. b0 + B - 1 of Bc)
    b0 = rng.choice([0, 0, 1, 3])
    Bc = B + b0 + rng.choice([0, 2])
    k = torch.randn((smax, Bc, heads, d), generator=g, device="cuda").to(bf)
    v = torch.randn((smax, Bc, heads, d), generator=g, device="cuda").to(bf)
    torch.cuda.synchronize()
    y = ctx.attention(q, k, v, S, heads, b0=b0)
    ctx.synchronize()
    t = ref(q.view(B, T, heads, d), k[:S, b0:b0 + B], v[:S, b0:b0 + B], prefill).reshape(B, T, heads * d)
    err = (y.float() - t.float()).abs()
    tol = 0.03 + 0.02 * t.float().abs()          # a few bf16 ulps of the output: p carries 8 bits, the sums differ in order
    bad = int((err > tol).sum())
    close = float((err <= 0.008 + 0.008 * t.float().abs()).float().mean())
    flag = "" if (bad == 0 and close > 0.97) else "   <-- FAIL"
    if flag or case % 10 == 0:
        print(f"case {case}: {'prefill' if prefill else 'decode'} B={B} (rows {b0}.. of {Bc}) T={T} S={S} h={heads} d={d}: max err {float(err.max()):.4f}, "
              f"within 1 ulp {close:.4f}, outside tol {bad}{flag}", flush=True)
    if flag:
        sys.exit(1)
print(f"{n_cases} cases ok")
