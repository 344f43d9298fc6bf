import sys, os, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))
import torch
from lia_amd import _native as N_, ops
lib = N_.lib()
def bits(t): return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
class Plan(ctypes.Structure): _fields_=[("bn",ctypes.c_int),("split",ctypes.c_int),("cps",ctypes.c_int)]
lib.lia_chain_plan_gemm.argtypes=[ctypes.c_int]*5+[ctypes.POINTER(Plan)]
for (M, N, K) in [(4, 128256, 4096), (64, 50272, 7168), (128, 32768, 1024), (20, 4096, 512), (4, 65536, 4096), (4, 131072, 4096), (16, 128256, 4096), (64, 128256, 4096)]:
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = (0.02 * torch.randn((N, K), generator=g, device="cuda")).to(torch.bfloat16)
    bias = (0.1 * torch.randn((N,), generator=g, device="cuda")).to(torch.bfloat16)
    torch.cuda.synchronize()          # torch fills the inputs on ITS stream; the context's stream is not ordered behind it
    ctx = ops.Context(0, 8 * M * N * 4 + (1 << 20))
    p = Plan(); lib.lia_chain_plan_gemm(M, N, K, 0, 256, ctypes.byref(p))
    lib.lia_gemm_set_split_policy(1)
    outs = []
    for engine in (1, 0):
        lib.lia_gemm_set_engine(engine)
        y = ctx.linear(x, w, bias=bias, relu=True); ctx.synchronize(); outs.append(bits(y))
    lib.lia_gemm_set_engine(1)
    y2 = ctx.linear(x, w, bias=bias, relu=True); ctx.synchronize(); again = bits(y2)
    ref = torch.relu((x.float() @ w.float().t()).to(torch.bfloat16).float() + bias.float()).to(torch.bfloat16)
    rb = bits(ref)
    def f32(b): return (b.astype(np.uint32) << 16).view(np.float32)
    e1 = np.abs(f32(outs[0]) - f32(rb)).max(); e0 = np.abs(f32(outs[1]) - f32(rb)).max()
    print(f"   chain twice equal: {(again == outs[0]).all()}  max|chain - torch| {e1:.4g}  max|skinny2 - torch| {e0:.4g}  exact-vs-torch chain {(outs[0]==rb).mean():.4f} skinny2 {(outs[1]==rb).mean():.4f}")
    d = outs[0] != outs[1]
    cols = np.nonzero(d.any(axis=0))[0]
    print(f"M={M} N={N} K={K} plan bn={p.bn} split={p.split} cps={p.cps}: {int(d.sum())} differ; cols {cols[:6]}..{cols[-6:] if len(cols) else ''} ({len(cols)} cols), tiles {sorted(set((cols // p.bn).tolist()))[:12]}")
    lib.lia_gemm_set_split_policy(0); lib.lia_gemm_set_engine(0); ctx.close()
