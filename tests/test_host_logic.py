"""CPU tests of the host-side logic added around the hot path: NUMA pinning / cgroup guards (hostinfo), the slices of the
all-gather weight stream, the host-computed layer set of the cooperative split, the data-parallel stream mode switch."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))


def test_pin_to_node_narrows_affinity_and_is_harmless_when_unknown():
    from lia_amd import hostinfo
    before = os.sched_getaffinity(0)
    try:
        cpus = hostinfo.node_cpus(0)
        if cpus:
            n = hostinfo.pin_to_node(0)
            assert n == len(before & cpus) and os.sched_getaffinity(0) == (before & cpus)
        assert hostinfo.pin_to_node(987) == 0                    # no such node: nothing changes, no exception
        assert hostinfo.node_cpus(987) is None
    finally:
        os.sched_setaffinity(0, before)
    assert hostinfo.gpu_numa_node(0) in (-1, 0, 1, 2, 3)        # -1 without a GPU


def test_cgroup_guards():
    from lia_amd import hostinfo
    m = hostinfo.cgroup_memory()
    assert set(m) == {"current", "peak", "max"}
    hostinfo.guard_host_allocation(1 << 20, "one MiB")           # always fits
    if m["max"] is not None:
        with pytest.raises(MemoryError):
            hostinfo.guard_host_allocation(m["max"], "the whole limit again")
    thr = hostinfo.cgroup_cpu_throttle()
    assert len(thr) == 2 and thr[0] >= 0 and thr[1] >= 0
    assert hostinfo.default_host_threads(8) >= 1 and hostinfo.default_host_threads(1) >= hostinfo.default_host_threads(8)


def test_allgather_slices_cover_the_wire_bytes():
    from lia_amd.model import LayerStore
    for total in (1, 255, 256, 832_000_123, 1_233_315_840):
        for world in (1, 2, 4, 8):
            sh = LayerStore.shard_bytes(total, world)
            assert sh % 256 == 0 and sh * world >= total and sh * (world - 1) < total + 256 * world
    # a raw OPT-30B packed layer (2048-aligned) splits exactly for 2 / 4 / 8 ranks: the slot has no room for padding
    assert all(LayerStore.shard_bytes(1_233_315_840, g) * g == 1_233_315_840 for g in (2, 4, 8))


def test_cpu_layer_set_properties():
    from lia_amd.scheduler import OffloadScheduler as S
    for n_gpu, L in ((4, 48), (0, 12), (3, 64), (9, 96)):
        for c in (0, 1, 5, 11, 16, 200):
            s = S.cpu_layer_set(n_gpu, L, c)
            assert len(s) == min(max(c, 0), L - n_gpu - 1)
            assert all(n_gpu < i < L for i in s)                 # never a resident layer, never the first streamed one
    s = sorted(S.cpu_layer_set(4, 48, 11))
    gaps = [b - a for a, b in zip(s, s[1:])]
    assert max(gaps) - min(gaps) <= 1                            # evenly spread


def test_dp_stream_mode_switch(monkeypatch):
    from lia_amd import dp
    monkeypatch.delenv("LIA_DP_STREAM", raising=False)
    assert dp.DataParallelGroup(None, 0, 2).mode == "broadcast"
    monkeypatch.setenv("LIA_DP_STREAM", "AllGather")
    assert dp.DataParallelGroup(None, 1, 2).mode == "allgather"
    monkeypatch.setenv("LIA_DP_STREAM", "ring")
    with pytest.raises(ValueError):
        dp.DataParallelGroup(None, 0, 2)


def test_rank_core_slices_do_not_overlap():
    from lia_amd import dp
    before = os.sched_getaffinity(0)
    try:
        seen = []
        for r in range(2):
            os.sched_setaffinity(0, before)
            g = dp.DataParallelGroup(None, r, 2, local_rank=r)
            n = g.pin_host_threads()
            cores = os.sched_getaffinity(0)
            assert n == len(cores) >= 1 and cores <= before
            seen.append(cores)
        if len(before) >= 4:
            assert not (seen[0] & seen[1])
    finally:
        os.sched_setaffinity(0, before)
