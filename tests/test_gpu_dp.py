"""-m gpu: the batch-shard data-parallel path end to end with TWO ranks sharing the one GPU of the test box.
Each rank runs the real OffloadScheduler (HIP kernels, streamer, staging + decode of the packed wire formats); the
root owns the host copy of the streamed layers, the other rank's layers are "remote" and arrive only through the
chunked per-layer broadcast.  The collective backend here is gloo (RCCL refuses two ranks on one device); the
scheduler code is the same one `bench.py --gpus N` runs over RCCL.  The gathered ids must equal the HF golden ids."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir, fmt, name, gpu_percentage, mode, backend="gloo", rows=None):
    for p in (ROOT, os.path.join(ROOT, "isca-2025-lia_amd"), GOLD):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LIA_DP_CHUNK_BYTES=str(96 * 1024), LIA_DP_STREAM=mode)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    if backend == "nccl":          # RCCL: one rank per device, so world size 1 on the test box
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import synth
        from lia_amd import dp
        from lia_amd.generation import generate
        from lia_amd.model import LiaOPTModel, OPTShape
        from lia_amd.scheduler import OffloadScheduler
        z = np.load(os.path.join(GOLD, name + ".npz"))
        vocab, max_pos, H, heads, F, L, B, T, new, seed = [int(v) for v in z["cfg"]]
        m = synth.make_model(seed, vocab, max_pos, H, F, L, float(z["w_std"][0]))
        ids = synth.make_prompt_ids(seed + 1, B, T, vocab)
        want = z["ids_bf16"]
        if rows is not None:                      # a global batch of len(rows) rows drawn from the golden ones (rows are independent)
            ids, want, B = ids[list(rows)], want[list(rows)], len(rows)
        shape = OPTShape("test", H, heads, F, L, vocab=vocab, max_pos=max_pos)
        model = LiaOPTModel.from_numpy(shape, m)
        n_gpu = int(L * gpu_percentage / 100)
        g = dp.DataParallelGroup(dist, rank, world, rank)
        if not g.is_root and mode == "broadcast":
            for li, st in enumerate(model.layers):
                if li >= n_gpu:
                    st._free()
                    st._np = None
                    st.tier = "remote"          # this rank never sees the host copy
        model._lia_scheduler = OffloadScheduler(model, device=0, dp_group=g, wire=fmt)
        mine = g.shard(torch.from_numpy(ids))
        lo, hi = dp.shard_rows(B, rank, world)
        out = generate(model, mine, max_new_tokens=new, min_new_tokens=new, prefill_policy=0, decoding_policy=2,
                       gpu_percentage=gpu_percentage, pin_weight=True)
        assert out.shape[0] == hi - lo and (out.numpy() == want[lo:hi]).all(), (rank, out[:, T:].tolist())
        full = g.gather_ids(out, B)
        assert full.shape[0] == B and (full.numpy() == want).all()
        if g.is_root or mode == "allgather":       # the host copies really are in the wire format that was asked for
            assert all(st.packed == {"raw": 0, "pack10": 10}[fmt] for st in model.layers[n_gpu:]), [st.packed for st in model.layers]
        if backend == "nccl":
            assert dist.get_backend() == "nccl" and model._lia_scheduler.pipe.copy_stream is not None
            b, ms = model._lia_scheduler.stream_stats()
            assert b > 0 and ms > 0          # the chunked copies really went through the streamer's copy stream
        if mode == "allgather" and world > 1:       # every rank pinned exactly its slice of every streamed layer's wire bytes
            assert all(st.shard is not None and st.shard[:2] == (rank, world) for st in model.layers[n_gpu:])
        elif not g.is_root:
            assert all(st.tier == "remote" for st in model.layers[n_gpu:])
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
        model._lia_scheduler.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["broadcast", "allgather"])
@pytest.mark.parametrize("fmt", ["raw", "pack10"])
@pytest.mark.parametrize("gpu_percentage", [0, 50])
def test_two_ranks_one_gpu_match_golden(tmp_path, fmt, gpu_percentage, mode):
    """mode = how a streamed layer reaches both ranks: the root's copy + one broadcast, or each rank's half + one all-gather."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), fmt, "generate_h256", gpu_percentage, mode), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


@pytest.mark.parametrize("mode", ["broadcast", "allgather"])
@pytest.mark.parametrize("fmt", ["raw", "pack10"])
def test_rccl_backend_world1_streamer_matches_golden(tmp_path, fmt, mode):
    """The code path the first multi-GPU run executes, with the REAL backend (torch.distributed "nccl" = RCCL) at the only world
    size one GPU allows: meta broadcast, per-chunk lia_stream_copy_chunk + dist.broadcast(async_op=True) issued under
    torch.cuda.stream(ExternalStream(copy stream)), work.wait() ordering, packed staging -> lia_stream_decode_packed ->
    lia_stream_mark_ready, both LIA_DP_STREAM modes.  (r02 only ever ran it by hand with LIA_FORCE_DP=1.)"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(1, port, str(tmp_path), fmt, "generate_h256", 25, mode, "nccl"), nprocs=1, join=True)
    assert (tmp_path / "ok0").exists()


@pytest.mark.parametrize("mode,fmt", [("broadcast", "pack10"), ("allgather", "raw")])
def test_four_ranks_one_gpu_ragged_global_batch(tmp_path, mode, fmt):
    """FOUR ranks on the one GPU, a global batch of 6 rows (not divisible by 4: ranks hold 2, 2, 1, 1): dp.shard_rows' remainder
    ranks run a smaller batch through the same broadcast / all-gather schedule, and the gathered ids are the golden rows in order.
    (What `bench.py --gpus 4 --global-batch 6` shards.)"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(4, port, str(tmp_path), fmt, "generate_h256", 50, mode, "gloo", (0, 1, 2, 3, 1, 0)), nprocs=4, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(4))
