"""Print what the host side of this box looks like to the process: affinity mask, physical cores behind it, CFS quota, NUMA
nodes, the GPU's node -- the facts behind box-to-box differences of the host-computed layers (cpu_baseline 77 vs 107 tokens/s)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "isca-2025-lia_amd"))
from lia_amd import hostinfo  # noqa: E402

aff = sorted(os.sched_getaffinity(0))
cores = {}
for c in aff:
    try:
        sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
    except OSError:
        sib = str(c)
    cores.setdefault(sib, []).append(c)
print("affinity:", len(aff), "logical CPUs,", len(cores), "physical cores;", "first:", aff[:8], "last:", aff[-4:])
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a",
      "| usable_cpus:", hostinfo.usable_cpus(), "| default_host_threads:", hostinfo.default_host_threads(1))
print("cpu model:", hostinfo.cpu_model(), "| numa nodes:", hostinfo.numa_nodes(), "| gpu node:", hostinfo.gpu_numa_node(0))
for n in hostinfo.numa_nodes() or []:
    nc = hostinfo.node_cpus(n)
    print(f"  node {n}: {len(nc)} CPUs, {len(nc & set(aff))} in the affinity mask")
try:
    print("loadavg:", open("/proc/loadavg").read().strip())
    mhz = [float(l.split(":")[1]) for l in open("/proc/cpuinfo") if l.startswith("cpu MHz")]
    print("cpu MHz now: min %.0f max %.0f mean %.0f over %d" % (min(mhz), max(mhz), sum(mhz) / len(mhz), len(mhz)))
except Exception as e:
    print("no /proc facts:", e)
