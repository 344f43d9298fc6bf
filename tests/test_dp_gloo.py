"""world_size-2 CPU tests (gloo) of the data-parallel plumbing: batch sharding, id gather, and the chunked
layer-weight broadcast pipeline (the one collective of the path).  The GPU kernels are not involved; the
per-rank layer executor is the CPU oracle (tests may use it) so the sharded result can be compared with the
unsharded one."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir):
    for p in (ROOT, os.path.join(ROOT, "isca-2025-lia_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lia_amd import dp
        import synth
        import lia_oracle as orc
        g = dp.DataParallelGroup(dist, rank, world, rank, chunk_bytes=1000)

        # 1. chunked broadcast of a "layer": root copies host -> slot chunk by chunk, peers receive
        nbytes = 10_007
        host = torch.arange(nbytes, dtype=torch.int64).to(torch.uint8)       # root's host copy
        slot = torch.zeros(nbytes, dtype=torch.uint8)
        copied = []

        def before(off, n):                                                 # the H2D of that chunk
            slot[off:off + n] = host[off:off + n]
            copied.append((off, n))

        works = dp.broadcast_chunked(dist, slot, 0, g.chunk_bytes, before_chunk=before if g.is_root else None)
        for w in works:
            w.wait()
        assert len(works) == 11 and torch.equal(slot, host)
        if g.is_root:
            assert copied == dp.chunk_ranges(nbytes, 1000)

        # 2. batch-sharded generation == unsharded generation (rows are independent)
        z = np.load(os.path.join(ROOT, "tests", "golden", "generate_h256.npz"))
        vocab, max_pos, H, heads, F, L, B, T, new, seed = [int(v) for v in z["cfg"]]
        m = synth.make_model(seed, vocab, max_pos, H, F, L, float(z["w_std"][0]))
        rs = np.random.RandomState(3)
        ids = rs.randint(4, vocab, size=(5, T)).astype(np.int64)            # 5 DIFFERENT rows over 2 ranks: 3 + 2
        mine = g.shard(torch.from_numpy(ids))
        lo, hi = dp.shard_rows(5, rank, world)
        assert mine.shape[0] == hi - lo == (3 if rank == 0 else 2)
        out, _ = orc.generate(m, mine.numpy(), 3, heads, 0, 2, 50)
        full = g.gather_ids(torch.from_numpy(out), 5)
        ref, _ = orc.generate(m, ids, 3, heads, 0, 2, 50)
        assert full.shape == (5, T + 3) and (full.numpy() == ref).all()
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_dp_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_shard_rows_partition():
    from lia_amd import dp
    for n in (1, 7, 64, 256):
        for w in (1, 2, 3, 8):
            spans = [dp.shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert dp.chunk_ranges(10, 4) == [(0, 4), (4, 4), (8, 2)]


class _FakeLib:
    """Records the streamer calls the WeightPipeline makes (no HIP)."""

    def __init__(self):
        self.log = []

    def lia_stream_create(self, ctx, n, bytes_, out):
        return 0

    def lia_stream_slot_ptr(self, h, s):
        return 0x1000 * (s + 1)

    def lia_stream_prefetch(self, h, slot, ptr, n, pinned):
        self.log.append(("prefetch", slot))
        return 0

    def lia_stream_wait(self, h, slot, st):
        self.log.append(("wait", slot))
        return 0

    def lia_stream_release(self, h, slot, st):
        self.log.append(("release", slot))
        return 0

    def lia_stream_destroy(self, h):
        pass


@pytest.mark.parametrize("n_slots", [2, 3])
def test_weight_pipeline_slot_order(monkeypatch, n_slots):
    """prefetch depth = n_slots - 1, slots rotate, and after the last layer the pipeline wraps to the first
    streamed layer of the next forward (so the copy engine never idles between token steps)."""
    from lia_amd import scheduler, _native

    fake = _FakeLib()
    monkeypatch.setattr(_native, "lib", lambda: fake)

    class Store:
        nbytes = 64
        packed = 0
        stream_bytes = 64

        def host_ptr(self):
            return 0x9000

        def is_dma_able(self):
            return True

    class Model:
        layer_bytes = 64
        offsets = list(range(16))
        layers = [Store() for _ in range(6)]

    class Ctx:
        handle, stream = 1, 2

    pipe = scheduler.WeightPipeline(Ctx(), Model(), n_slots)
    first, L = 2, 6
    for step in range(2):
        pipe.prefetch(first)
        for idx in range(first, L):
            pipe.acquire(idx)
            pipe.prefetch(idx + 1 if idx + 1 < L else first)
            pipe.release(idx)
    pf = [s for op, s in fake.log if op == "prefetch"]
    assert len(pf) == 2 * (L - first) + 1                       # one copy per layer per step (+ the wrapped one)
    assert pf == [i % n_slots for i in range(len(pf))]          # slots rotate
    waits = [s for op, s in fake.log if op == "wait"]
    assert waits == pf[:len(waits)]                             # layers are consumed in the order they were copied
    # a slot is never re-filled before the layer that used it was released
    held = set()
    for op, s in fake.log:
        if op == "prefetch":
            assert s not in held
        elif op == "wait":
            held.add(s)
        elif op == "release":
            held.discard(s)
