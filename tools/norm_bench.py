"""LayerNorm at prefill size (rows x H bf16) stand-alone: GB/s of the kernel lia_layernorm_launch picks.
LIA_ROW_NORM_MAX_ROWS=1000000 python tools/norm_bench.py   -> the workgroup-per-row kernel instead of the wave-per-row one"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "isca-2025-lia_amd"))
import torch
from lia_amd import ops
ctx = ops.Context(0, 1 << 24)
for rows, H in ((16384, 7168), (8192, 7168), (131072, 4096), (64, 7168)):
    x = torch.randn(rows, H, device="cuda").to(torch.bfloat16)
    g = torch.ones(H, device="cuda", dtype=torch.bfloat16); b = torch.zeros_like(g)
    for _ in range(3):
        ctx.layernorm(x, g, b)
    ctx.synchronize()
    import time
    t0 = time.time()
    n = 20
    for _ in range(n):
        ctx.layernorm(x, g, b)
    ctx.synchronize()
    dt = (time.time() - t0) / n
    print(f"rows {rows} H {H}: {dt * 1e6:.1f} us per call, {2 * rows * H * 2 / dt / 1e9:.0f} GB/s")
