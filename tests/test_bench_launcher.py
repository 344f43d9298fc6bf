"""`python bench.py --gpus N` must start its own ranks when no launcher did (the first multi-GPU run of the driver
would otherwise exit non-zero).  CPU check of exactly that code path: the parent (which never imports torch) spawns
`python -m torch.distributed.run --nproc-per-node 2 bench.py ...`, the two ranks rendezvous over gloo on 127.0.0.1, one
all-reduce, rank 0 prints one JSON line, the parent relays it as ITS last line and returns the children's exit code."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_its_ranks():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launcher"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    last = [ln for ln in p.stdout.splitlines() if ln.strip()][-1]
    d = json.loads(last)
    assert d == {"launcher_selftest": True, "world": 2, "sum": 3.0}


def test_bench_flag_surface():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.build_parser().parse_args([])
    assert (a.gpus, a.steps, a.warmup, a.batch, a.prompt, a.gpu_percentage, a.prefill_policy, a.decoding_policy) == (1, 31, 0, 64, 256, 10, None, None)
    from lia_amd.model import resolve_shape
    shape = resolve_shape("opt-30b")
    lb = 2 * (4 * shape.hidden ** 2 + 2 * shape.hidden * shape.ffn)
    # one GPU: the reference's 0 / 2 (BASELINE configs[1]); N > 1: KV in HBM when it fits, the named policies always win
    assert bench.plan_policies(a, 1, shape, 64, 256, 32, lb, 288 << 30, 4)[:2] == (0, 2)
    assert bench.plan_policies(a, 8, shape, 32, 256, 32, lb, 288 << 30, 4)[:2] == (3, 3)
    assert bench.plan_policies(a, 8, shape, 32, 256, 32, lb, 16 << 30, 4)[:2] == (0, 2)           # a 16 GiB device: no room
    named = bench.build_parser().parse_args(["--prefill-policy", "0", "--decoding-policy", "2"])
    assert bench.plan_policies(named, 8, shape, 32, 256, 32, lb, 288 << 30, 4)[:2] == (0, 2)
    assert 1 + a.warmup + a.steps == 32                      # the config's "out 32" by default
    a = bench.build_parser().parse_args(["--gpus", "8", "--global-batch", "256", "--steps", "20", "--warmup", "5"])
    assert a.global_batch // a.gpus == 32                    # BASELINE config 5


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("lia_bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_watchdog_names_the_leg_and_exits_non_zero(capsys):
    """a data-parallel extra leg that hangs in a collective: rank 0 re-prints the headline line with the leg's name and what had
    finished, every rank ends with exit code 3 -- never 0, the process was killed inside GPU work"""
    import json
    bench = _load_bench()
    out = {"metric": "m", "value": 1.0, "roofline": {"bound": "pcie", "dominant_kernel": {"frac": 0.6, "avg_launch_us": 70.0}}, "config": {"workload": "w"},
           "dp_extra_legs": "pending"}
    progress = {"current": "allgather", "res": {"value_kv_in_hbm": 123.0, "kv_in_hbm_leg": {"ms_per_step": 5.0}}}
    codes = []
    bench.watchdog_fire(out, progress, 0, 420, _exit=codes.append)
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert codes == [3] and bench.WATCHDOG_EXIT_CODE == 3
    assert line["dp_extra_legs"] == {"timed_out": True, "leg_running": "allgather", "timeout_s": 420, "legs_finished": ["kv_in_hbm"], "exit_code": 3}
    assert line["value_kv_in_hbm"] == 123.0 and line["value"] == 1.0
    codes.clear()
    naps = []
    bench.watchdog_fire(out, progress, 2, 420, _exit=codes.append, _sleep=naps.append)      # another rank: no line, same exit code ...
    assert codes == [3] and capsys.readouterr().out == ""
    assert naps == [bench.WATCHDOG_GRACE_S] and bench.WATCHDOG_GRACE_S >= 1.0                # ... after a grace period: rank 0's line gets out first


def test_dp_line_at_world_8_over_gloo():
    """`python bench.py --gpus 8 --global-batch 256` as far as a box without GPUs can take it (r04 verdict item 6): the self-launch,
    eight ranks over gloo, and everything the line says about the SPLIT of the job through the functions main() itself uses --
    plan_rows (32 rows per rank), plan_policies (KV in HBM by default at N > 1), hostinfo.default_host_threads (this container's CPUs
    over 8 ranks), reduce_over_ranks (MAX of the per-rank times, all-gather of the per-rank figures), dp_config_fields."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--selftest-dp-line", "--global-batch", "256", "--model", "opt-30b",
                        "--steps", "5", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert d["dp_line_selftest"] is True and d["n_gpus"] == 8 and d["collective_ranks"] == 8 and d["rccl_ranks"] == 8 and d["scaling"] == "strong"
    # the prediction the line carries for itself (planner.predict_dp): both cache placements, the binding resource named
    pred = d["config"]["predicted"]
    assert set(pred) >= {"3/3", "0/2", "assumptions"} and pred["3/3"]["bound_by"] and pred["3/3"]["ms_per_step"] > 0
    assert pred["0/2"]["host_attention_threads_per_rank"] == 2 and pred["0/2"]["ms_per_step"] >= pred["3/3"]["ms_per_step"]
    assert d["config"]["global_batch"] == 256 and d["config"]["rows_per_rank"] == [32] * 8 and d["config"]["parallelism"].startswith("dp8 batch-shard")
    assert d["config"]["policies"]["prefill"] == 3 and d["config"]["policies"]["decode"] == 3 and "HBM" in d["config"]["policies"]["why"]
    assert [r["rank"] for r in d["per_rank"]] == list(range(8)) and all(r["rows"] == 32 for r in d["per_rank"])
    from lia_amd import hostinfo
    thr = hostinfo.default_host_threads(8)
    assert all(r["host_attention_threads"] == thr and r["host_threads_starved"] == (thr < 4) for r in d["per_rank"])
    assert [r["host_attention_ms_per_step"] for r in d["per_rank"]] == [1.5 * r for r in range(8)] and d["per_rank"][0]["h2d_gbs"] == 50.0
    # MAX over ranks: the slowest rank (17 ms per step) sets the job's rate
    assert abs(d["ms_per_step"] - 17.0) < 1e-9 and abs(d["value"] - 256 * 5 / (5 * 17e-3)) < 1e-6 and abs(d["prefill_ms"] - 107.0) < 1e-9
    # a ragged split: 19 rows over 8 ranks
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--selftest-dp-line", "--global-batch", "19", "--model", "opt-125m",
                        "--prefill-policy", "0", "--decoding-policy", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert d["config"]["rows_per_rank"] == [3, 3, 3, 2, 2, 2, 2, 2] and [r["rows"] for r in d["per_rank"]] == [3, 3, 3, 2, 2, 2, 2, 2]
    assert d["config"]["policies"]["prefill"] == 0 and d["config"]["policies"]["decode"] == 2


def test_bench_line_carries_its_scalars_inside_roofline_and_config():
    """the driver keeps only the standard keys: value_raw_format, value_cooperative, prefill_ms, parity, first-divergence steps,
    the dominant kernel's fraction and the prefill's MFMA fraction are repeated inside `roofline`, the wire format inside `config`"""
    from types import SimpleNamespace as NS
    bench = _load_bench()
    out = {"roofline": {"bound": "pcie", "frac": 0.89, "dominant_kernel": {"frac": 0.6, "avg_launch_us": 70.0}},
           "config": {"workload": "w"}, "prefill_ms": 690.0, "value_raw_format": 66.9, "value_cooperative": 176.4,
           "cooperative_leg": {"controller": {"converged": True}}, "cooperative_kv_in_hbm_leg": {"controller": {"converged": False}},
           "prefill_detail": {"mfma_frac": 0.58}, "parity": {"max_err_in_quanta": 2.0, "frac_bit_identical": 0.4},
           "prefill_defer_kv_0_leg": {"prefill_ms": 727.0},
           "ids_check": {"pack10_vs_raw_wire": {"first_divergent_step": None}, "cooperative_vs_headline": {"first_divergent_step": 3, "top2_logit_gap_at_divergence": 0.0625}},
           "host_link": {"stream_format": "pack10", "bits_per_value": 10.8, "bits_per_value_by_layer": {"layers": 44, "min": 10.7, "mean": 10.8, "max": 16.0, "layers_shipped_raw": 1}}}
    sc = bench.promote_scalars(out)["roofline"]["scalars"]
    assert sc["value_raw_format"] == 66.9 and sc["value_cooperative"] == 176.4 and sc["prefill_ms"] == 690.0 and sc["prefill_ms_defer_kv_0"] == 727.0
    assert sc["dominant_kernel_frac"] == 0.6 and sc["prefill_mfma_frac"] == 0.58 and sc["parity_max_err_in_quanta"] == 2.0
    assert sc["ids_first_divergent_step"] == {"pack10_vs_raw_wire": None, "cooperative_vs_headline": 3} and sc["cooperative_converged"] == [True, False]
    assert sc["ids_top2_gap_at_divergence"] == {"cooperative_vs_headline": 0.0625}
    assert out["config"]["stream_format"] == "pack10" and out["config"]["bits_per_value"]["layers_shipped_raw"] == 1
    # all-resident line: the decode GEMM is the roofline itself
    res = bench.promote_scalars({"roofline": {"bound": "hbm", "kernel": "k", "frac": 0.7, "avg_launch_us": 50.0}, "config": {}})
    assert res["roofline"]["scalars"] == {"dominant_kernel_frac": 0.7, "dominant_kernel_avg_launch_us": 50.0}
    st = lambda n, sb, packed, tier="pinned": NS(nbytes=n, stream_bytes=sb, packed=packed, tier=tier, shard=None)      # noqa: E731
    model = NS(layers=[st(1000, 1000, 0, "device"), st(1000, 675, 10), st(1000, 1000, 0), st(1000, 700, 10)])
    ws = bench.wire_stats(model, 1)
    assert ws["layers"] == 3 and ws["layers_shipped_raw"] == 1 and ws["max"] == 16.0 and abs(ws["min"] - 10.8) < 1e-9


def test_bench_and_harness_share_one_default_wire_format(monkeypatch):
    """bench.py, run.py and OffloadScheduler resolve an unnamed wire format through the same function"""
    from lia_amd import run_generation, scheduler
    bench = _load_bench()
    monkeypatch.delenv("LIA_STREAM_FORMAT", raising=False)
    assert scheduler.default_stream_format() == scheduler.DEFAULT_STREAM_FORMAT == "pack10"
    assert bench.build_parser().parse_args([]).stream_format is None and run_generation.build_parser().parse_args([]).stream_format is None
    monkeypatch.setenv("LIA_STREAM_FORMAT", "raw")
    assert scheduler.default_stream_format() == "raw"
    monkeypatch.setenv("LIA_STREAM_FORMAT", "zip")
    with pytest.raises(ValueError):
        scheduler.default_stream_format()
