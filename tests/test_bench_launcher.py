"""`python bench.py --gpus N` must start its own ranks when no launcher did (the first multi-GPU run of the driver
would otherwise exit non-zero).  CPU check of exactly that code path: the parent (which never imports torch) spawns
`python -m torch.distributed.run --nproc-per-node 2 bench.py ...`, the two ranks rendezvous over gloo on 127.0.0.1, one
all-reduce, rank 0 prints one JSON line, the parent relays it as ITS last line and returns the children's exit code."""
import json
import os
import subprocess
import sys

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
    assert (a.gpus, a.steps, a.warmup, a.batch, a.prompt, a.gpu_percentage, a.prefill_policy, a.decoding_policy) == (1, 31, 0, 64, 256, 10, 0, 2)
    assert 1 + a.warmup + a.steps == 32                      # the config's "out 32" by default
    a = bench.build_parser().parse_args(["--gpus", "8", "--global-batch", "256", "--steps", "20", "--warmup", "5"])
    assert a.global_batch // a.gpus == 32                    # BASELINE config 5
