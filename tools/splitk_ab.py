"""In-launch split-K combine of the skinny GEMM (lia_gemm.hip) against the two-kernel form of r01, bit for bit, and against
itself over many back-to-back launches with other work in between (a stale slab read would show as a changing result).
    python tools/splitk_ab.py save ref.pt                                   (the default two-kernel form)
    LIA_GEMM_SPLITK=inlaunch python tools/splitk_ab.py check ref.pt"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "isca-2025-lia_amd"))
from lia_amd import ops  # noqa: E402

mode, path = sys.argv[1], sys.argv[2]
ctx = ops.Context(0, 1 << 30)
g = torch.Generator(device="cuda").manual_seed(3)
shapes = [(7168, 7168), (21504, 7168), (7168, 28672), (28672, 7168), (4096, 4096), (6144, 4096), (4096, 14336), (50272, 7168)]
outs = {}
junk = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
for (n, k) in shapes:
    w = (0.02 * torch.randn(n, k, generator=g, device="cuda")).to(torch.bfloat16)
    bias = (0.1 * torch.randn(n, generator=g, device="cuda")).to(torch.bfloat16)
    for m in (1, 16, 33, 64, 128, 200, 256):
        x = torch.randn(m, k, generator=g, device="cuda").to(torch.bfloat16)
        res = torch.randn(m, n, generator=g, device="cuda").to(torch.bfloat16)
        torch.cuda.synchronize()
        first = None
        for it in range(12):
            y = ctx.linear(x, w, bias, res, relu=(it % 2 == 0 and False))
            if it % 3 == 1:
                ctx.synchronize()
                junk.random_(0, 255)              # dirty the caches between launches
                torch.cuda.synchronize()
            ctx.synchronize()
            if first is None:
                first = y.clone()
            elif not torch.equal(first, y):
                bad = (first != y).sum().item()
                raise SystemExit(f"N={n} K={k} M={m}: launch {it} differs from launch 0 in {bad} elements (stale slab?)")
        outs[(n, k, m)] = first.cpu()
    del w
print("self-consistent over 12 launches per case:", len(outs), "cases")
if mode == "save":
    torch.save(outs, path)
else:
    ref = torch.load(path)
    for key, v in outs.items():
        if not torch.equal(ref[key], v):
            raise SystemExit(f"{key}: in-launch combine differs from the two-kernel form in {(ref[key] != v).sum().item()} elements")
    print("bit-identical to the reference run:", len(outs), "cases")
