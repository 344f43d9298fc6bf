"""A/B on the box: one OPT-30B decode layer on the host cores (tools/host_layer_bench.py) with the process restricted to
(a) every logical CPU of the GPU's NUMA node (what bench.py does), (b) ONE logical CPU per physical core of that node,
(c) 16 fixed physical cores of that node, one thread each (OMP_PROC_BIND)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))
from lia_amd import hostinfo  # noqa: E402

node = hostinfo.gpu_numa_node(0) or 0
cpus = sorted(hostinfo.node_cpus(node) & os.sched_getaffinity(0))
seen, one_per_core = set(), []
for c in cpus:
    sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
    if sib not in seen:
        seen.add(sib)
        one_per_core.append(c)
masks = {"node": cpus, "one_per_core": one_per_core, "16_cores_bound": one_per_core[:16]}
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for name, m in masks.items():
        env = dict(os.environ)
        if name == "16_cores_bound":
            env.update(OMP_PROC_BIND="true", OMP_PLACES="cores")
        code = (f"import os; os.sched_setaffinity(0, {set(m)!r}); import runpy, sys; sys.argv=['host_layer_bench.py','16']; "
                f"runpy.run_path({os.path.join(ROOT, 'tools', 'host_layer_bench.py')!r}, run_name='__main__')")
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        line = [l for l in r.stdout.splitlines() if "pinned" in l]
        print(f"round {rnd} {name:16s} ({len(m)} CPUs): {line[0] if line else r.stderr[-300:]}", flush=True)
