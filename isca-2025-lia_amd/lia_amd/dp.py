"""Batch-sharded data parallelism for the offloaded decoder: one process per GPU,
rows [r*B/G, (r+1)*B/G) on rank r, no activation exchange.  The ONE collective is the per-layer weight
broadcast: rank 0 owns the host copy of every streamed layer, moves it once over its PCIe link in chunks,
and each chunk is RCCL-broadcast over xGMI to the other ranks' slot while the next chunk is still in
flight from the host.  (torch.distributed backend "nccl" IS RCCL on ROCm.)

The reference has no data parallelism on this path (its only multi-rank mode is a 2-socket CPU tensor
parallel branch, decoder.py:60-77); this module is the build's own addition named by BASELINE.json.
"""
import os

import torch

DEFAULT_CHUNK = 256 << 20  # bytes per host->device->broadcast pipeline stage (64 MB chunks cost 2 % in copy gaps at world size 1)


def shard_rows(n_rows, rank, world):
    """Contiguous row range of `rank`; the first n_rows % world ranks take one extra row."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def chunk_ranges(nbytes, chunk):
    return [(o, min(chunk, nbytes - o)) for o in range(0, nbytes, chunk)]


def broadcast_chunked(dist, buf, src, chunk_bytes=DEFAULT_CHUNK, group=None, before_chunk=None):
    """Broadcast a flat uint8 tensor in chunks, each issued asynchronously right after `before_chunk(off, n)`
    (on the root: the host->device copy of that chunk) so transfer k+1 overlaps broadcast k.  Returns the
    list of work handles (wait on them before declaring the buffer ready)."""
    works = []
    for off, n in chunk_ranges(buf.numel(), chunk_bytes):
        if before_chunk is not None:
            before_chunk(off, n)
        works.append(dist.broadcast(buf[off:off + n], src=src, group=group, async_op=True))
    return works


class DataParallelGroup:
    def __init__(self, dist, rank, world, local_rank=0, chunk_bytes=None):
        self.dist, self.rank, self.world, self.local_rank = dist, rank, world, local_rank
        self.chunk_bytes = chunk_bytes or int(os.environ.get("LIA_DP_CHUNK_BYTES", DEFAULT_CHUNK))
        self.root = 0
        # how a streamed layer reaches the G ranks: "broadcast" (BASELINE.json: rank 0 reads it over its link, one RCCL
        # broadcast) or "allgather" (SURVEY.md section 8e alternative: every rank keeps and reads 1/G of the wire bytes over
        # ITS OWN host link, one all-gather over xGMI puts the layer together -- G times the host-link rate)
        self.mode = os.environ.get("LIA_DP_STREAM", "broadcast").lower()
        if self.mode not in ("broadcast", "allgather"):
            raise ValueError(f"LIA_DP_STREAM={self.mode!r}: expected broadcast or allgather")

    @property
    def is_root(self):
        return self.rank == self.root

    def shard(self, input_ids):
        lo, hi = shard_rows(input_ids.shape[0], self.rank, self.world)
        return input_ids[lo:hi]

    def gather_ids(self, ids_local, n_rows_total):
        """All ranks' generated ids, concatenated in row order (token ids are tiny; gathered once per generate)."""
        device = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        sizes = [shard_rows(n_rows_total, r, self.world) for r in range(self.world)]
        width = ids_local.shape[1]
        pad = max(hi - lo for lo, hi in sizes)
        mine = torch.zeros((pad, width), dtype=torch.int64, device=device)
        mine[:ids_local.shape[0]] = ids_local.to(device)
        outs = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(outs, mine)
        return torch.cat([o[:hi - lo].cpu() for o, (lo, hi) in zip(outs, sizes)], dim=0)

    def pin_host_threads(self, ncpu=None):
        """Give each rank its own slice of the physical cores it may use (the current affinity mask -- bench.py has already
        narrowed it to the NUMA node of this rank's GPU) for the policy-2 host attention, so G ranks do not oversubscribe one
        another (OpenMP threads inherit the process affinity mask).  Slices are indexed by local rank over `world` equal
        parts, so ranks that share a node never overlap (host_core_slice)."""
        ncpu = ncpu or (os.cpu_count() or 1)
        try:
            avail = sorted(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            return 0
        cores = host_core_slice(avail, ncpu, self.local_rank, self.world)
        try:
            os.sched_setaffinity(0, cores)
        except (AttributeError, OSError):
            pass
        return len(cores)


def host_core_slice(avail, ncpu, local_rank, world):
    """The cores of `avail` (sorted ids this process may run on) that local rank `local_rank` of `world` keeps: the physical cores
    (ids below ncpu / 2 -- the SMT siblings follow them in Linux's numbering) cut into `world` equal runs; the remainder goes
    unused rather than making slices uneven (a rank with one more thread finishes its host attention no earlier: the step waits
    for the slowest).  Fewer physical cores than ranks: the ranks share them round-robin, one core each."""
    phys = [c for c in avail if c < max(1, ncpu // 2)] or list(avail)
    if not phys:
        return set()
    if len(phys) < world:
        return {phys[local_rank % len(phys)]}
    per = len(phys) // world
    lo = (local_rank % world) * per
    return set(phys[lo:lo + per])


class RawDeviceBuffer:
    """Zero-copy torch view of device memory owned by liblia_hip.so (a streamer slot)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    def tensor(self):
        return torch.as_tensor(self, device="cuda")
