#!/usr/bin/env python3
"""Where a persistent decode-chain launch (csrc/lia_chain.hip) spends its time: per step of the program, when the workgroups enter
it (behind the seam), when their first chunk has landed, when they are done and when their stores have drained -- 100 MHz clock
stamps of every workgroup (lia_chain_set_stamps), min / median / max over the workgroups, relative to the launch's first stamp.

    python tools/chain_stamps.py [llama|opt] [layers] [B] [T]
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))

MAX_OPS, SLOTS = 8, 8


def main():
    import torch
    from lia_amd import _native as N
    fam = sys.argv[1] if len(sys.argv) > 1 else "llama"
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    B = int(sys.argv[3]) if len(sys.argv) > 3 else (128 if fam == "llama" else 64)
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    lib = N.lib()
    lib.lia_chain_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.lia_chain_set_stamps.restype = None
    if fam == "llama":
        from lia_amd.llama import LiaLlamaModel, LlamaKVState, LlamaScheduler, LlamaShape
        shape = LlamaShape("llama-3-8b-cut", 4096, 32, 8, 14336, L, 4096, max_pos=T + 64)
        model = LiaLlamaModel.random_init(shape, seed=1)
        sched = LlamaScheduler(model)
        kv = LlamaKVState(model, B, T + 40)
        fwd = lambda ids: sched.forward(ids, kv)   # noqa: E731
    else:
        from lia_amd.model import LiaOPTModel, OPTShape
        from lia_amd.scheduler import KVState, OffloadScheduler
        shape = OPTShape("opt-30b-cut", 7168, 56, 28672, L, vocab=4096, max_pos=T + 64)
        model = LiaOPTModel.random_init(shape, seed=1, n_gpu_layers=L)
        sched = OffloadScheduler(model)
        kv = KVState(model, L, B, T + 40)
        fwd = lambda ids: sched.forward(ids, kv, prefill_policy=0, decoding_policy=2, gpu_percentage=100, pin_weight=True)   # noqa: E731
    ids = torch.randint(4, shape.vocab, (B, T), generator=torch.Generator().manual_seed(0))
    _, nxt = fwd(ids)
    for _ in range(6):                                     # warm
        _, nxt = fwd(nxt.view(B, 1).cpu())
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    buf = torch.zeros((SLOTS, n_cu, MAX_OPS, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    lib.lia_chain_set_stamps(buf.data_ptr(), SLOTS)
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    _, nxt = fwd(nxt.view(B, 1).cpu())
    torch.cuda.synchronize()
    lib.lia_chain_set_stamps(None, 0)
    st = buf.cpu().numpy().astype(np.int64)
    names = {"llama": ["o", "R norm", "gate|up", "down", "R norm", "q|k|v", "R rope"], "opt": ["o", "R norm", "fc1", "R relu", "fc2", "R norm", "q|k|v", "R bias"]}[fam]
    for slot in range(min(SLOTS, L + 1)):
        s = st[slot]
        live = s[:, :, 0] > 0
        if not live.any():
            continue
        base = s[:, :, 0][live].min()
        n_ops = int(live.any(axis=0).sum())
        print(f"--- launch {slot}: {n_ops} steps; times in us from the launch's first stamp (min / median / max over the workgroups)")
        lab = names if n_ops >= 5 else names[-2:]
        for op in range(n_ops):
            row = []
            for k, what in enumerate(("enter", "first chunk", "done", "drained")):
                v = s[:, op, k]
                v = v[v > 0]
                if v.size == 0:
                    row.append(f"{what:>11}: -")
                    continue
                r = (v - base) / 100.0
                row.append(f"{what:>11}: {r.min():7.2f} {np.median(r):7.2f} {r.max():7.2f}")
            print(f"  {op} {lab[op] if op < len(lab) else '?':8s} " + " | ".join(row))
        end = s[:, :n_ops, 3]
        print(f"  launch span {((end[end > 0].max() - base) / 100.0):.2f} us")
    # whole steps, un-stamped, for scale
    for _ in range(3):
        _, nxt = fwd(nxt.view(B, 1).cpu())
    torch.cuda.synchronize()
    import time
    t = time.time()
    for _ in range(10):
        _, nxt = fwd(nxt.view(B, 1).cpu())
    torch.cuda.synchronize()
    print(f"decode step (un-stamped): {1e3 * (time.time() - t) / 10:.3f} ms for {L} layers")


if __name__ == "__main__":
    main()
