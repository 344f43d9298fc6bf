"""One opt-125m-shaped decode layer at batch 1 on the host cores (lia_host_layer_forward) over the thread count -- configs[0]'s unit of
work.  Run it under the launchers' wait policy (OMP_WAIT_POLICY=PASSIVE GOMP_SPINCOUNT=0) to see what the product pays per OpenMP
region; LIA_HOST_LAYER_REGIONS=9 restores one region per op."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo/isca-2025-lia_amd')
import torch
from lia_amd import _native as N, ops
L = N.lib()
H, heads, F, B, S = 768, 12, 3072, 1, 40
desc = ops.make_desc(H, heads, F)
offs, total = ops.pack_offsets(desc)
rs = np.random.RandomState(0)
flat = ((0.02 * rs.standard_normal(total // 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
x = ((rs.standard_normal((B, 1, H))).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
y = np.empty_like(x)
k = np.zeros((S + 8, B, heads, H // heads), np.uint16) + 0x3c00
v = k.copy()
w = ops.weight_ptr_array(flat.ctypes.data, offs)
for threads in (1, 2, 4, 8):
    args = (ctypes.byref(desc), ctypes.byref(w), x.ctypes.data, y.ctypes.data, k.ctypes.data, v.ctypes.data, S + 8, B, B, 1, S, 0, threads)
    for _ in range(5): N.check(L.lia_host_layer_forward(*args))
    ts = []
    for _ in range(50):
        t0 = time.time(); N.check(L.lia_host_layer_forward(*args)); ts.append(1e3 * (time.time() - t0))
    ts.sort()
    print(f"opt-125m layer B=1: threads={threads}: min {ts[0]:.3f} median {ts[25]:.3f} ms")
