"""Time the product's host (policy-1) linear on the box's cores: OPT-30B decode shapes at M = 64."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))
import torch  # noqa: F401,E402  (the library shares torch's HIP runtime)
from lia_amd import _native as N, hostinfo  # noqa: E402

L = N.lib()
threads = int(sys.argv[1]) if len(sys.argv) > 1 else hostinfo.default_host_threads(1)
M = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rs = np.random.RandomState(0)
for name, n, k in (("qkv", 21504, 7168), ("out", 7168, 7168), ("fc1", 28672, 7168), ("fc2", 7168, 28672)):
    x = (rs.standard_normal((M, k)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    w = (np.resize((0.02 * rs.standard_normal(1 << 22)).astype(np.float32).view(np.uint32) >> 16, n * k)).astype(np.uint16)
    y = np.empty((M, n), np.uint16)
    args = (x.ctypes.data, w.ctypes.data, None, None, y.ctypes.data, M, n, k, 0, threads)
    L.lia_host_linear(*args)
    t0 = time.time()
    for _ in range(10):
        L.lia_host_linear(*args)
    dt = (time.time() - t0) / 10
    print(f"{name}: {dt * 1e3:.2f} ms  {2 * n * k / dt / 1e9:.1f} GB/s of weights  {2.0 * M * n * k / dt / 1e12:.2f} TFLOP/s  threads={threads}")
