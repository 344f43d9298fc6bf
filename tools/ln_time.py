"""Time the prefill-sized LayerNorm (16384 x 7168) and RMSNorm-sized rows on the GPU box: python tools/ln_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))
import torch
from lia_amd import ops
ctx = ops.Context(0, 1 << 20)
for rows, H in [(16384, 7168), (16384, 12288), (4096, 2048)]:
    x = torch.randn((rows, H), device="cuda").to(torch.bfloat16)
    g = torch.randn((H,), device="cuda").to(torch.bfloat16); b = torch.randn((H,), device="cuda").to(torch.bfloat16)
    for _ in range(3): y = ctx.layernorm(x, g, b)
    ctx.synchronize(); t0 = time.time()
    for _ in range(50): y = ctx.layernorm(x, g, b)
    ctx.synchronize(); dt = (time.time() - t0) / 50
    print(f"layernorm {rows} x {H}: {dt * 1e6:.1f} us = {2 * rows * H * 2 / dt / 1e12:.2f} TB/s   checksum {float(y.float().sum()):.4f}", flush=True)
