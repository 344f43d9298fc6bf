"""Host facts the cooperative (CPU+GPU) policies depend on: how many cores this process may really use.

The reference pins its CPU work with `OMP_NUM_THREADS=40 numactl -m 0 -C 0-39` (README.md:78;
llm/scripts/*.sh).  Here the usable core count is the minimum of the affinity mask and the cgroup CPU
quota: a container may expose 256 CPUs yet be throttled to 16 CPUs' worth of time, and an OpenMP team larger
than the quota stalls every layer's host-attention round trip.
"""
import math
import os


def cgroup_cpu_quota():
    """CPUs' worth of time the cgroup grants (cpu.max: "<quota> <period>" or "max"), or None if unlimited."""
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                return max(1, int(math.floor(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            return max(1, q // p)
    except (OSError, ValueError):
        pass
    return None


def usable_cpus():
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    q = cgroup_cpu_quota()
    return max(1, min(n, q) if q else n)


def default_host_threads(world=1):
    """Threads for the policy-2 host attention of one rank: the usable CPUs split across ranks; when the box is
    not quota-limited only the physical cores (half the SMT threads) are counted."""
    n = usable_cpus()
    if cgroup_cpu_quota() is None:
        n = max(1, n // 2)
    return max(1, n // max(1, world))


class HostTeam:
    """Threads of the team that computes WHOLE LAYERS back to back (the cooperative split's host layers, policy 1): the attention
    team's count, fixed.  (r03 tried a governor that gave threads back while cpu.stat showed CFS throttling -- sixteen busy threads
    plus the main thread's helpers overshoot a sixteen-CPU quota now and then: three boxes, 15 threads ahead by 6 % on one, 16
    ahead by 3-5 % on the other two, run-to-run spread as large as the effect, LABNOTES.md.  The switch is gone; bench.py still
    reports the throttled time per leg as `cpu_throttle`.)"""

    def __init__(self, team):
        self.threads = max(1, int(team))


def cap_torch_threads(n=None):
    """torch sizes its intra-op thread pool by the CPUs it SEES (128 on the GPU box) -- not by the cgroup quota (16 there).  The
    few CPU tensor ops of the token loop (torch.cat of the ids, greedy_search.py:408) then start a 128-thread team that the quota
    serialises: measured 0.4 - 39 ms per decode step for a 1 MB cat, against 30 us on one thread.  The reference runs under
    OMP_NUM_THREADS=40 (README.md:78), which bounds that pool too; this is the same bound from the box's own numbers.  Only ever
    lowers the setting.  Returns the thread count in force."""
    import torch
    n = n or default_host_threads(1)
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()


def node_cpus(node):
    """CPU ids of a NUMA node (sysfs cpulist), or None"""
    try:
        txt = open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
    except OSError:
        return None
    cpus = set()
    for part in txt.split(","):
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def numa_nodes():
    """ids of the NUMA nodes that hold memory (sysfs), sorted; [] when unknown"""
    import glob
    import re
    out = []
    for p in glob.glob("/sys/devices/system/node/node[0-9]*"):
        m = re.search(r"node(\d+)$", p)
        if not m:
            continue
        try:
            txt = open(os.path.join(p, "meminfo")).read()
            mt = re.search(r"MemTotal:\s+(\d+)", txt)
            if mt and int(mt.group(1)) > 0:
                out.append(int(m.group(1)))
        except OSError:
            pass
    return sorted(out)


def gpu_numa_node(dev_index=0):
    """NUMA node of a GPU from sysfs (PCI ids via torch's device properties), or -1 when unknown: pinned host memory is
    allocated next to the GPU, so the host-compute threads belong on that node's cores."""
    try:
        import torch
        p = torch.cuda.get_device_properties(dev_index)
        path = f"/sys/bus/pci/devices/{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0/numa_node"
        return int(open(path).read().strip())
    except Exception:
        return -1


def pin_node(dev_index=0):
    """The NUMA node the host threads are confined to: $LIA_PIN_NODE when set (-1 = no pinning), else the GPU's own node."""
    env = os.environ.get("LIA_PIN_NODE")
    if env is not None and env.strip() != "":
        return int(env)
    return gpu_numa_node(dev_index)


def pin_to_node(node):
    """Restrict this process (and the OpenMP teams it will create) to the CPUs of one NUMA node -- the reference's
    `numactl -m 0 -C 0-39` (README.md:78) for a box where numactl is not ours to run.  Returns the CPU count or 0."""
    cpus = node_cpus(node)
    if not cpus:
        return 0
    try:
        allowed = os.sched_getaffinity(0) & cpus
        if allowed:
            os.sched_setaffinity(0, allowed)
            return len(allowed)
    except (AttributeError, OSError):
        pass
    return 0


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def isa_flags():
    want = ("avx512f", "avx512_bf16", "amx_bf16", "amx_tile")
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                have = set(line.split(":", 1)[1].split())
                return {w: (w in have) for w in want}
    except OSError:
        pass
    return {w: False for w in want}


def host_memory_budget():
    """Bytes of host memory this process may still take without risking the container: the smaller of
    MemAvailable and (cgroup memory.max - memory.current).  None when neither can be read."""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                avail = int(line.split()[1]) * 1024
                break
    except OSError:
        pass
    room = None
    try:
        mx = open("/sys/fs/cgroup/memory.max").read().strip()
        if mx != "max":
            room = int(mx) - int(open("/sys/fs/cgroup/memory.current").read().strip())
    except (OSError, ValueError):
        pass
    vals = [v for v in (avail, room) if v is not None]
    return min(vals) if vals else None


def cgroup_cpu_throttle():
    """(nr_throttled, throttled_usec) of the cgroup so far: CFS quota stalls hit every thread of the container at once, so a
    host-compute team that (with the runtime's helper threads) overshoots cpu.max pays for it in whole scheduler periods."""
    out = {"nr_throttled": 0, "throttled_usec": 0}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, _, v = line.partition(" ")
            if k in out:
                out[k] = int(v)
    except (OSError, ValueError):
        pass
    return out["nr_throttled"], out["throttled_usec"]


def cgroup_memory():
    """{"current", "peak", "max"} bytes of the cgroup (None where unreadable): what a run really pinned / may pin."""
    out = {}
    for k in ("current", "peak", "max"):
        try:
            v = open(f"/sys/fs/cgroup/memory.{k}").read().strip()
            out[k] = None if v == "max" else int(v)
        except (OSError, ValueError):
            out[k] = None
    return out


def guard_host_allocation(nbytes, what, ceiling=0.93):
    """Per-allocation guard for pinned / registered host memory: refuse (MemoryError) when the cgroup would end up
    above `ceiling` of its limit.  check_host_allocation() judges a plan up front; this one is called right before every
    large pinned allocation, so an underestimated plan ends in an exception, never in a dead container."""
    m = cgroup_memory()
    if m["max"] is None or m["current"] is None:
        return
    if m["current"] + nbytes > ceiling * m["max"]:
        raise MemoryError(f"{what}: {nbytes / 2**30:.2f} GiB more would put the container at "
                          f"{(m['current'] + nbytes) / 2**30:.1f} GiB of its {m['max'] / 2**30:.1f} GiB limit; refusing")


def check_host_allocation(nbytes, what, safety=0.85):
    """Refuse (MemoryError) a planned host allocation that would not fit: pinned / registered pages cannot be
    reclaimed, so overshooting a cgroup limit takes the whole container down instead of failing one malloc."""
    budget = host_memory_budget()
    if budget is not None and nbytes > safety * budget:
        raise MemoryError(f"{what}: needs {nbytes / 2**30:.1f} GiB of host memory but only {budget / 2**30:.1f} GiB "
                          f"are available to this container (cgroup limit / MemAvailable); refusing to allocate")
