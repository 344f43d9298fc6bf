"""Time lia_attention's prefill at a few (B, T, heads, d) on the GPU box: python tools/attn_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))
import torch
from lia_amd import ops
ctx = ops.Context(0, 1 << 20)
bf = torch.bfloat16
for (B, T, heads, d) in [(64, 256, 32, 64), (32, 1024, 32, 64), (16, 2048, 32, 64), (64, 256, 56, 128), (16, 2048, 32, 128)]:
    q = torch.randn((B, T, heads * d), device="cuda").to(bf)
    k = torch.randn((T, B, heads, d), device="cuda").to(bf)
    v = torch.randn((T, B, heads, d), device="cuda").to(bf)
    for _ in range(3):
        y = ctx.attention(q, k, v, T, heads)
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        y = ctx.attention(q, k, v, T, heads)
    ctx.synchronize()
    print(f"B {B} T {T} heads {heads} d {d}: {(time.time() - t0) / 20 * 1e6:.1f} us   checksum {float(y.float().abs().sum()):.6e}", flush=True)
