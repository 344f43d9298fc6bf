"""Time one policy-1 decode step of an OPT-30B-shaped layer on the host cores (lia_host_layer_forward), with the weights
in pageable numpy memory and in pinned (hipHostMalloc) memory."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))
import torch  # noqa: F401,E402
from lia_amd import _native as N, hostinfo, ops  # noqa: E402

L = N.lib()
threads = int(sys.argv[1]) if len(sys.argv) > 1 else hostinfo.default_host_threads(1)
H, heads, F, B, S = 7168, 56, 28672, 64, 272
desc = ops.make_desc(H, heads, F)
offs, total = ops.pack_offsets(desc)
rs = np.random.RandomState(0)
blk = ((0.02 * rs.standard_normal(1 << 22)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
flat = np.resize(blk, total // 2)
x = ((rs.standard_normal((B, 1, H))).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
y = np.empty_like(x)
k = np.resize(blk, (S + 8) * B * H).reshape(S + 8, B, heads, H // heads).copy()
v = k.copy()


def run(wptr, label):
    w = ops.weight_ptr_array(wptr, offs)
    args = (ctypes.byref(desc), ctypes.byref(w), x.ctypes.data, y.ctypes.data, k.ctypes.data, v.ctypes.data, S + 8, B, B, 1, S, 0, threads)
    N.check(L.lia_host_layer_forward(*args))
    ts = []
    for _ in range(20):
        t0 = time.time()
        N.check(L.lia_host_layer_forward(*args))
        ts.append(1e3 * (time.time() - t0))
    ts.sort()
    print(f"{label}: min {ts[0]:.2f}  median {ts[len(ts) // 2]:.2f}  mean {sum(ts) / len(ts):.2f} ms per layer step (threads={threads})")


run(flat.ctypes.data, "pageable weights")
p = L.lia_host_alloc_pinned(total)
ctypes.memmove(p, flat.ctypes.data, total)
run(p, "pinned weights  ")
