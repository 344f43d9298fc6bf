import sys, os, time, ctypes, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/isca-2025-lia_amd"]
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a", "affinity:", len(os.sched_getaffinity(0)))
import torch
from lia_amd import _native as N
L = N.lib()
for (B, heads, d, S) in [(64, 56, 128, 272)]:
    H = heads * d
    q = torch.randn(B, 1, H).bfloat16(); k = torch.randn(B, 1, H).bfloat16(); v = torch.randn(B, 1, H).bfloat16()
    kc = torch.randn(S + 8, B, heads, d).bfloat16().pin_memory(); vc = torch.randn(S + 8, B, heads, d).bfloat16().pin_memory()
    out = torch.empty(B, 1, H, dtype=torch.bfloat16)
    for nt in (2, 16):
        ts = []
        for it in range(15):
            t0 = time.time()
            rc = L.lia_host_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), B, 1, S - 1, heads, d, B, 0, nt)
            ts.append(time.time() - t0)
        ts.sort()
        dt = ts[0]
        print(f"B={B} h={heads} d={d} S={S} threads={nt}: min {dt*1e3:.2f} median {ts[len(ts)//2]*1e3:.2f} ms  ({2*S*B*H*2/dt/1e9:.1f} GB/s at min)")
