"""The analytical planner reproduces the r01 measurements it is calibrated on and behaves
monotonically."""
from lia_amd import planner
from lia_amd.model import resolve_shape

BOX = planner.Box(host_threads=16, host_mem_gb=280.0)


def test_headline_prediction_close_to_measurement():
    sh = resolve_shape("opt-30b")
    raw = planner.Box(host_threads=16, host_mem_gb=280.0, wire_ratio=1.0)
    pre, dec, hbm, host, n_gpu = planner.estimate(sh, 64, 256, 32, 10, 2, raw)
    assert n_gpu == 4
    assert abs(dec - 955.0) / 955.0 < 0.05          # raw bf16 on the wire: measured 955 ms/step
    assert abs(pre - 1050.0) / 1050.0 < 0.10        # measured 1013-1050 ms
    assert 50 < host < 90 and hbm < 30
    pre10, dec10, *_ = planner.estimate(sh, 64, 256, 32, 10, 2, BOX)           # pack10 (the default wire ratio)
    assert abs(dec10 - 649.0) / 649.0 < 0.05        # measured 649 ms/step
    assert abs(pre10 - 840.0) / 840.0 < 0.12        # measured 831-861 ms: compute-bound once the wire is packed
    pre_r, dec_r, *_ = planner.estimate(sh, 64, 256, 32, 100, 3, BOX)
    assert abs(dec_r - 18.7) / 18.7 < 0.35          # measured 18.7 ms/step fully resident
    assert abs(pre_r - 858.0) / 858.0 < 0.15


def test_monotonic_in_gpu_percentage_and_plan_picks_resident_when_it_fits():
    sh = resolve_shape("opt-30b")
    decs = [planner.estimate(sh, 64, 256, 32, p, 2, BOX)[1] for p in (0, 10, 50, 90)]
    assert decs == sorted(decs, reverse=True)
    p = planner.plan(sh, 64, 256, 32, BOX)
    assert p.n_gpu_layers == sh.layers and p.gpu_percentage == 100       # 59 GB fits 288 GB of HBM
    big = resolve_shape("opt-175b")
    p2 = planner.plan(big, 32, 256, 32, BOX)
    assert 0 < p2.n_gpu_layers < big.layers and p2.host_gb <= 0.85 * BOX.host_mem_gb and p2.hbm_gb <= 0.92 * 288


def test_plan_cpu_layers_matches_the_measured_optimum():
    """OPT-30B, B = 64, gpu% = 10, 16 pinned host threads, pack10 on the wire: the r03 scans found the link-bound/host-bound
    crossover at 18-19 host-computed layers (384-390 ms per step) with the KV cache on the host and at 21-23 layers (318-335 ms)
    with the GPU layers' cache in HBM (16 / 412 and 19 / 367 with r02's host linears)."""
    from lia_amd import planner
    from lia_amd.model import resolve_shape
    sh = resolve_shape("opt-30b")
    box = planner.Box(host_threads=16, host_mem_gb=300.0)
    c, ms = planner.plan_cpu_layers(sh, 64, 256, 32, 10, box)
    assert 17 <= c <= 20 and 350 <= ms <= 400, (c, ms)
    c3, ms3 = planner.plan_cpu_layers(sh, 64, 256, 32, 10, box, kv_in_hbm=True)
    assert 20 <= c3 <= 24 and 310 <= ms3 <= 345 and c3 >= c and ms3 < ms, (c3, ms3)
    # a raw wire format makes every shipped layer dearer, so more layers move to the host; few host threads -> fewer
    c_raw, _ = planner.plan_cpu_layers(sh, 64, 256, 32, 10, planner.Box(host_threads=16, host_mem_gb=300.0, wire_ratio=1.0))
    c_weak, _ = planner.plan_cpu_layers(sh, 64, 256, 32, 10, planner.Box(host_threads=4, host_mem_gb=300.0))
    assert c_raw > c and c_weak < c
