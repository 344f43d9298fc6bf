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


# ---------------------------------------------------------------------------------------------------------------------
# world 4 and 8 (gloo, CPU): row sharding with remainders, the id gather, per-rank host-core slices, and the two ways a
# streamed layer reaches every rank -- WeightPipeline._prefetch_broadcast / _prefetch_allgather -- over a streamer whose slots
# are host memory (_HostSlotLib below stands in for lia_stream_*: the copies are memmoves, the collectives are the real ones).
# ---------------------------------------------------------------------------------------------------------------------
class _HostSlotLib:
    """lia_stream_* over host memory: what the WeightPipeline drives, with the H2D copies as memmoves.  The `packed` wire
    format of this stand-in is the raw bytes' first `stream_bytes` XOR 0x5A (decode = XOR back, zero-fill the rest)."""

    def __init__(self, n_slots, layer_bytes, cap):
        import ctypes
        self.ct = ctypes
        self.slots = [(ctypes.c_uint8 * layer_bytes)() for _ in range(n_slots)]
        self.staging = [(ctypes.c_uint8 * cap)() for _ in range(n_slots)]
        self.cap, self.layer_bytes, self.ready, self.copies = cap, layer_bytes, [], []

    def lia_stream_create(self, ctx, n, bytes_, out):
        return 0

    def lia_stream_slot_ptr(self, h, s):
        return self.ct.addressof(self.slots[s])

    def lia_stream_staging_ptr(self, h, s):
        return self.ct.addressof(self.staging[s])

    def lia_stream_copy_stream(self, h):
        return 0

    def lia_stream_begin(self, h, slot):
        return 0

    def _copy(self, dst, slot, off, src, n):
        src = src.value if hasattr(src, "value") else src
        self.ct.memmove(self.ct.addressof(dst[slot]) + off, src, n)
        self.copies.append((slot, off, n))
        return 0

    def lia_stream_copy_chunk(self, h, slot, off, src, n, pinned):
        return self._copy(self.slots, slot, off, src, n)

    def lia_stream_copy_chunk_packed(self, h, slot, off, src, n, pinned):
        return self._copy(self.staging, slot, off, src, n)

    def lia_pack10_bound(self, n):
        return self.cap

    def lia_stream_decode_packed(self, h, slot, n_values, fmt):
        self.decoded = getattr(self, "decoded", []) + [(slot, fmt)]
        return 0

    def lia_stream_mark_ready(self, h, slot):
        self.ready.append(slot)
        return 0

    def lia_stream_wait(self, h, slot, st):
        return 0

    def lia_stream_release(self, h, slot, st):
        return 0

    def lia_stream_destroy(self, h):
        pass


class _HostBuffer:
    """dp.RawDeviceBuffer over host memory"""

    def __init__(self, ptr, nbytes):
        import ctypes
        self.arr = np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(ptr))

    def tensor(self):
        return torch.from_numpy(self.arr)


def _wide_worker(rank, world, port, tmpdir):
    for p in (ROOT, os.path.join(ROOT, "isca-2025-lia_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes
        from lia_amd import _native, dp, scheduler
        from lia_amd.model import LayerStore
        g = dp.DataParallelGroup(dist, rank, world, rank, chunk_bytes=1000)

        # 1. rows with a remainder: every rank's shard, and the gather puts them back in row order
        n_rows = world * 2 + 3                                            # 11 rows over 4 ranks, 19 over 8
        ids = torch.arange(n_rows * 5, dtype=torch.int64).view(n_rows, 5)
        mine = g.shard(ids)
        lo, hi = dp.shard_rows(n_rows, rank, world)
        assert mine.shape[0] == hi - lo == 2 + (1 if rank < 3 else 0) and torch.equal(mine, ids[lo:hi])
        full = g.gather_ids(mine + 7, n_rows)
        assert torch.equal(full, ids + 7)
        # fewer rows than ranks: the last ranks hold none, the gather still returns every row once
        few = torch.arange(3 * 2, dtype=torch.int64).view(3, 2)
        assert g.shard(few).shape[0] == (1 if rank < 3 else 0)
        assert torch.equal(g.gather_ids(g.shard(few), 3), few)

        # 2. host-core slices: disjoint, equal, inside the mask (computed for a 64-thread node; the affinity call itself runs on
        #    whatever this container allows)
        slices = [dp.host_core_slice(list(range(64)), 64, r, world) for r in range(world)]
        assert all(len(s_) == 32 // world for s_ in slices) and len(set().union(*slices)) == 32 and max(set().union(*slices)) < 32
        assert [dp.host_core_slice([0, 1, 2], 6, r, world) for r in range(4)] == [{0}, {1}, {2}, {0}][:4] if world >= 4 else True
        assert g.pin_host_threads() >= 1

        # 3. a streamed layer reaches every rank: broadcast from the root, then all-gather of per-rank slices; raw and "packed"
        layer_bytes, cap, n_layers, n_slots = 10_240, 12_288, 3, 2
        rs = np.random.RandomState(5)
        layers_raw = [rs.randint(0, 256, layer_bytes).astype(np.uint8) for _ in range(n_layers)]      # the same draw on every rank
        packed_len = 7_001                                                # odd: the slices need padding
        for mode in ("broadcast", "allgather"):
            for packed in (0, 10):
                wire = [(l[:packed_len] ^ 0x5A) if packed else l for l in layers_raw]
                nbytes = len(wire[0])
                sh = LayerStore.shard_bytes(nbytes, world)
                fake = _HostSlotLib(n_slots, layer_bytes, cap)
                _native.lib = lambda fake=fake: fake
                dp.RawDeviceBuffer = _HostBuffer
                torch.cuda.ExternalStream = lambda ptr: None
                g.mode = mode

                class Store:
                    tier = "pinned"

                    def __init__(self, i):
                        self.nbytes, self.packed, self.stream_bytes, self.i = layer_bytes, packed, nbytes, i
                        if mode == "allgather":
                            buf = np.zeros(sh, np.uint8)
                            part = wire[i][rank * sh:(rank + 1) * sh]
                            buf[:len(part)] = part
                            self.shard = (rank, world, sh)
                        else:
                            buf = wire[i].copy() if rank == 0 else np.zeros(1, np.uint8)      # only the root holds the layer
                            self.shard = None
                        self.buf = buf

                    def host_ptr(self):
                        return self.buf.ctypes.data

                    def is_dma_able(self):
                        return True

                class Model:
                    pass

                model = Model()
                model.layer_bytes, model.offsets, model.layers = layer_bytes, list(range(16)), [Store(i) for i in range(n_layers)]

                class Ctx:
                    handle, stream = 1, 2

                pipe = scheduler.WeightPipeline(Ctx(), model, n_slots, dp_group=g)
                assert pipe.layer_meta == [[packed, nbytes]] * n_layers                 # the root's table on every rank
                for i in range(n_layers):
                    pipe.prefetch(i)
                    pipe.acquire(i)
                    slot = pipe.held[i]
                    got = np.ctypeslib.as_array(fake.staging[slot] if packed else fake.slots[slot])[:nbytes]
                    assert (got == wire[i]).all(), (mode, packed, i, rank)
                    if packed:
                        assert fake.decoded[-1] == (slot, packed)
                    pipe.release(i)
                assert fake.ready == [i % n_slots for i in range(n_layers)]
                if mode == "broadcast":
                    # only the root touches its host link, chunk by chunk; in all-gather mode every rank copies exactly its slice
                    assert (len(fake.copies) == n_layers * len(dp.chunk_ranges(nbytes, 1000))) == (rank == 0) or (rank != 0 and not fake.copies)
                else:
                    assert [(o, n) for _, o, n in fake.copies] == [(rank * sh, sh)] * n_layers
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_dp_world_4_and_8_gloo(tmp_path, world):
    port = _free_port()
    mp.spawn(_wide_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
