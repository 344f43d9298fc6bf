"""Host-tier H2D bandwidth microbenchmark: the measurement of lia/cxl/benchmark.py (:9-128) + run.sh, taken on the path the
scheduler itself uses -- the C ABI's weight streamer (lia_stream_prefetch: pinned hipMemcpyAsync on the copy stream into an HBM
slot) -- instead of the reference's `Tensor.copy_(non_blocking=True)`.

    python -m lia_amd.cxl.benchmark --gpu [--cpu] [--cxl] [--register] [--nodes 1]

--gpu : `number` = 3 copies of a 4 GiB host buffer (2048 x 2048 x 1024 int8, benchmark.py:16-18) per timed repeat, 3 repeats
        after 2 warm-up rounds; the source is DDR-pinned (lia_host_alloc_pinned) or, with --cxl, the NUMA-interleaved tier
        (numa_alloc_tensor); --register hipHostRegisters that range (what --enable-cxl does in the scheduler; the reference
        leaves it pageable, numa_alloc.py:49, so its copies are staged -- here an unregistered source goes through the
        streamer's pinned bounce buffer, the same staging).
--cpu : 8192^3 fp32 GEMMs on the host cores at the same time (benchmark.py:46-63), to show the interference.
Prints the reference's two lines: "[t s] Average Transfer Bandwidth: x GB/s" / "[t s] Average Compute Time: y seconds", and
returns {"transfer_gbs", "copy_engine_gbs", "compute_s"} (copy_engine_gbs: bytes / the copy stream's own busy time, lia_stream_stats).
"""
import argparse
import ctypes
import threading
import time

import torch

from .. import _native as N
from ..ops import Context
from .numa_alloc import numa_alloc_tensor, numa_free_tensor, set_cxl_nodes

NUMBER, REPEAT, WARMUP = 3, 3, 2          # benchmark.py:10-12
CHUNK = 4 << 30                           # slot size: the reference's 4 GiB buffer is ONE copy per transfer (larger buffers go in 4 GiB pieces)


class _Transfer:
    """the host buffer of one tier + a two-slot streamer; run() = REPEAT timed groups of NUMBER whole-buffer transfers.  A transfer
    is enqueued asynchronously; the only host-side wait is lia_stream_begin's collection of the SAME slot's previous copy, i.e. the
    copy engine always has the next transfer queued while the host thread sleeps -- which matters beside the CPU GEMM, when that
    thread may not get a CPU for tens of milliseconds (r05's first version moved 1 GiB pieces and read 22 GB/s there instead of 47)"""

    def __init__(self, from_cxl, register, nbytes):
        self.lib, self.nbytes, self.numa = N.lib(), nbytes, None
        if from_cxl:
            self.numa = numa_alloc_tensor((nbytes,), torch.int8, register=register)
            if self.numa is None:
                raise MemoryError("Failed to allocate NUMA memory.")
            self.numa.fill_(1)
            self.ptr, self.pinned = self.numa.data_ptr(), bool(register)
        else:
            self.ptr, self.pinned = self.lib.lia_host_alloc_pinned(nbytes), True
            if not self.ptr:
                raise MemoryError(N.lib().lia_last_error().decode(errors="replace"))
            ctypes.memset(self.ptr, 1, nbytes)
        self.ctx = Context(0, 0)
        from .. import hostinfo
        self.ctx.set_host_threads(hostinfo.default_host_threads(1))      # the team that stages an unregistered source (lia_api.hip::staged_copy)
        self.chunk = min(CHUNK, nbytes)
        self.h = ctypes.c_void_p()
        N.check(self.lib.lia_stream_create(self.ctx.handle, 2, self.chunk, ctypes.byref(self.h)), "lia_stream_create")
        self.costs, self.total, self.busy = [], 0.0, (0.0, 0.0)

    def _whole_buffer(self, slot):
        for off in range(0, self.nbytes, self.chunk):
            n = min(self.chunk, self.nbytes - off)
            N.check(self.lib.lia_stream_prefetch(self.h, slot, ctypes.c_void_p(self.ptr + off), n, int(self.pinned)), "lia_stream_prefetch")
            N.check(self.lib.lia_stream_wait(self.h, slot, ctypes.c_void_p(self.ctx.stream)), "lia_stream_wait")
            N.check(self.lib.lia_stream_release(self.h, slot, ctypes.c_void_p(self.ctx.stream)), "lia_stream_release")
            slot ^= 1
        return slot

    def run(self):
        b, ms = ctypes.c_double(), ctypes.c_double()
        self.lib.lia_stream_stats(self.h, ctypes.byref(b), ctypes.byref(ms), 1)
        self.costs, t_all, slot = [], time.time(), 0
        for _ in range(REPEAT):
            self.ctx.synchronize()
            t0 = time.time()
            for _ in range(NUMBER):
                slot = self._whole_buffer(slot)
            self.ctx.synchronize()
            self.costs.append(time.time() - t0)
        self.total = time.time() - t_all
        self.lib.lia_stream_stats(self.h, ctypes.byref(b), ctypes.byref(ms), 1)
        self.busy = (b.value, ms.value)

    def close(self):
        self.lib.lia_stream_destroy(self.h)
        self.ctx.close()
        if self.numa is not None:
            numa_free_tensor(self.numa)
        else:
            self.lib.lia_host_free_pinned(ctypes.c_void_p(self.ptr))


class _Compute:
    def __init__(self, mm):
        from .. import hostinfo
        # torch sizes its pool by the CPUs it SEES (256 on the GPU box), not by the container's quota (16): an unbounded 8192^3 GEMM
        # is throttled by CFS and takes every other thread of the process down with it.  The reference runs on a box it owns
        # (run.sh:1-14); the bound from this box's own numbers is the equivalent.
        hostinfo.cap_torch_threads()
        self.a, self.b = torch.ones(mm, mm), torch.ones(mm, mm)
        self.costs, self.total = [], 0.0

    def run(self):
        self.costs, t_all = [], time.time()
        for _ in range(REPEAT):
            t0 = time.time()
            for _ in range(NUMBER):
                torch.mm(self.a, self.b)
            self.costs.append(time.time() - t0)
        self.total = time.time() - t_all

    def close(self):
        pass


def benchmark(is_compute, is_transfer, from_cxl, register=False, size_scale=1.0, mm=8192, out=print):
    res, workers = {}, {}
    nbytes = int(2048 * size_scale) * 2048 * 1024
    try:
        if is_transfer:
            workers["transfer"] = _Transfer(from_cxl, register, nbytes)
    except MemoryError as e:
        out(str(e))
        return res
    if is_compute:
        workers["compute"] = _Compute(mm)
    for rnd in range(WARMUP + 1):                                     # the last round is the one reported
        threads = [threading.Thread(target=w.run) for w in workers.values()]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if is_transfer:
        w = workers["transfer"]
        res["transfer_gbs"] = nbytes * NUMBER / 2 ** 30 / (sum(w.costs) / len(w.costs))
        res["copy_engine_gbs"] = w.busy[0] / max(w.busy[1], 1e-9) / 1e6
        out(f"[{w.total:.3f} s] Average Transfer Bandwidth: {res['transfer_gbs']:.3f} GB/s")
    if is_compute:
        w = workers["compute"]
        res["compute_s"] = sum(w.costs) / len(w.costs)
        out(f"[{w.total:.3f} s] Average Compute Time: {res['compute_s']:.3f} seconds")
    for w in workers.values():
        w.close()
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
