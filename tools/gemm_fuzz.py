"""Randomised parity sweep of lia_linear (every GEMM regime: skinny MT 1..16, 128- and 256-row workgroups, the two-block x
cut, split-K 1..4, 128^2 and 256^2 tiles) against torch's fp32 matmul of the same bf16 operands, with the reference's
rounding points (bf16 after the matmul, after + bias, after + residual).  usage: python tools/gemm_fuzz.py [n_cases] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))
import torch  # noqa: E402
from lia_amd import ops  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = ops.Context(0, 1 << 20)
worst = 0.0
for case in range(n_cases):
    regime = rng.choice(["skinny", "skinny", "skinny", "mid", "big"])
    if regime == "skinny":
        M = rng.choice([1, 2, 7, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 96, 100, 127, 128, 129, 200, 255, 256])
        N = 16 * rng.randint(1, 2100)
        K = 128 * rng.randint(1, 64)
    elif regime == "mid":
        M = rng.randint(257, 1023)
        N = 16 * rng.randint(1, 200)
        K = 64 * rng.randint(1, 40)
    else:
        M = rng.choice([1024, 1100, 1280, 2048, 3000])
        N = 16 * rng.randint(32, 300)
        K = 64 * rng.randint(1, 60)
    relu, has_bias, has_res = rng.random() < 0.3, rng.random() < 0.7, rng.random() < 0.5
    g = torch.Generator(device="cuda").manual_seed(case)
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = (0.5 * torch.randn((N,), generator=g, device="cuda")).to(torch.bfloat16) if has_bias else None
    r = torch.randn((M, N), generator=g, device="cuda").to(torch.bfloat16) if has_res else None
    torch.cuda.synchronize()
    y = ctx.linear(x, w, b, r, relu=relu)
    ctx.synchronize()
    t = (x.float() @ w.float().t()).to(torch.bfloat16).float()
    if has_bias:
        t = (t + b.float()).to(torch.bfloat16).float()
    if relu:
        t = torch.relu(t)
    if has_res:
        t = (r.float() + t).to(torch.bfloat16).float()
    err = (y.float() - t).abs()
    # one bf16 ulp of the largest intermediate (|t| < 8 -> 0.031; residual cancellation can expose the pre-residual ulp)
    tol = 0.04 + 0.008 * t.abs()
    bad = int((err > tol).sum())
    exact = float((y.float() == t).float().mean())
    worst = max(worst, float(err.max()))
    flag = "" if (bad == 0 and exact > 0.95) else "   <-- FAIL"
    if flag or case % 20 == 0:
        print(f"case {case}: {regime} M={M} N={N} K={K} relu={int(relu)} bias={int(has_bias)} res={int(has_res)}: "
              f"max err {float(err.max()):.4f}, exact {exact:.4f}, outside tol {bad}{flag}", flush=True)
    if flag:
        sys.exit(1)
print(f"{n_cases} cases ok, worst abs err {worst:.4f}")
