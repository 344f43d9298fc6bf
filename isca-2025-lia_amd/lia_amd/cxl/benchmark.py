"""Host-tier H2D bandwidth microbenchmark: twin of lia/cxl/benchmark.py (:9-128) + run.sh.

    python -m lia_amd.cxl.benchmark --gpu [--cpu] [--cxl] [--register] [--nodes 1]

--gpu : time `number` x 4 GiB host->HBM copies (2048 x 2048 x 1024 int8, benchmark.py:16-18), DDR-pinned or, with
        --cxl, from the NUMA-interleaved tier (numa_alloc_tensor); --register additionally hipHostRegisters the
        NUMA range (what the scheduler's --enable-cxl path does; the reference cannot).
--cpu : concurrently run 8192^3 fp32 CPU GEMMs (benchmark.py:46-63) to show the interference.
Prints the reference's two lines: "[t s] Average Transfer Bandwidth: x GB/s" / "[t s] Average Compute Time: y seconds".
"""
import argparse
import time
from queue import Queue
from threading import Thread

import numpy as np
import torch

from .numa_alloc import numa_alloc_tensor, numa_free_tensor, set_cxl_nodes


def benchmark(is_compute, is_transfer, from_cxl, register=False, size_scale=1.0, mm=8192, out=print):
    number, repeat, warmup = 3, 3, 2
    res = {}
    if is_transfer:
        b0, s0, h0 = int(2048 * size_scale), 2048, 1024
        dtype = torch.int8
        size = b0 * s0 * h0 * number / (1024 ** 3)
        if from_cxl:
            t_cpu = numa_alloc_tensor((b0, s0, h0), dtype, register=register)
            if t_cpu is None:
                out("Failed to allocate NUMA memory.")
                return res
            t_cpu.fill_(1)
        else:
            t_cpu = torch.ones((b0, s0, h0), dtype=dtype, pin_memory=True, device="cpu")
        t_gpu = torch.ones((b0, s0, h0), dtype=dtype, device="cuda:0")

        def memcpy(queue):
            costs = []
            total = time.time()
            for _ in range(repeat):
                torch.cuda.synchronize()
                st = time.time()
                for _ in range(number):
                    t_gpu.copy_(t_cpu, non_blocking=True)
                torch.cuda.synchronize()
                costs.append(time.time() - st)
            queue.put(costs)
            queue.put(time.time() - total)

    if is_compute:
        mat1 = torch.ones(mm, mm, dtype=torch.float32)
        mat2 = torch.ones(mm, mm, dtype=torch.float32)

        def compute(queue):
            costs = []
            total = time.time()
            for _ in range(repeat):
                st = time.time()
                for _ in range(number):
                    _ = torch.mm(mat1, mat2)
                costs.append(time.time() - st)
            queue.put(costs)
            queue.put(time.time() - total)

    def one_round():
        qs, ths = {}, []
        if is_transfer:
            qs["t"] = Queue()
            ths.append(Thread(target=memcpy, args=(qs["t"],)))
        if is_compute:
            qs["c"] = Queue()
            ths.append(Thread(target=compute, args=(qs["c"],)))
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return qs

    for _ in range(warmup):
        one_round()
    qs = one_round()
    if is_transfer:
        times, total = qs["t"].get(), qs["t"].get()
        res["transfer_gbs"] = size / float(np.mean(times))
        out(f"[{total:.3f} s] Average Transfer Bandwidth: {res['transfer_gbs']:.3f} GB/s")
    if is_compute:
        times, total = qs["c"].get(), qs["c"].get()
        res["compute_s"] = float(np.mean(times))
        out(f"[{total:.3f} s] Average Compute Time: {res['compute_s']:.3f} seconds")
    if is_transfer and from_cxl:
        numa_free_tensor(t_cpu)
    return res


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpu", action="store_true", help="time the host->GPU transfer")
    p.add_argument("--cpu", action="store_true", help="run the CPU GEMM concurrently")
    p.add_argument("--cxl", action="store_true", help="source the transfer from the NUMA/CXL tier")
    p.add_argument("--register", action="store_true", help="hipHostRegister the NUMA range")
    p.add_argument("--nodes", default=None, help="comma-separated NUMA nodes of the CXL tier (default LIA_CXL_NODES or 2,3)")
    p.add_argument("--size-scale", type=float, default=1.0)
    p.add_argument("--mm", type=int, default=8192)
    a = p.parse_args(argv)
    if a.nodes:
        set_cxl_nodes([int(x) for x in a.nodes.split(",")])
    return benchmark(a.cpu, a.gpu, a.cxl, a.register, a.size_scale, a.mm)


if __name__ == "__main__":
    main()
