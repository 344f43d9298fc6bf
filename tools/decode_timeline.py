"""Where one decode step goes: kernels and the idle gaps between them, from a rocprofv3 --kernel-trace CSV of bench.py.
usage: python tools/decode_timeline.py <dir with *_kernel_trace.csv> [step-from-the-end=2]"""
import csv, glob, re, sys
from collections import defaultdict
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kt = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
def _grid(r):
    # workgroups of the launch (rocprofv3 reports the grid in work-items): tells the GEMM shapes of one template apart
    try:
        g = [int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0), int(r.get("Grid_Size_Y") or 1), int(r.get("Grid_Size_Z") or 1)]
        w = [int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 1), int(r.get("Workgroup_Size_Y") or 1), int(r.get("Workgroup_Size_Z") or 1)]
        return "x".join(str(max(1, a // max(1, b))) for a, b in zip(g, w))
    except (TypeError, ValueError):
        return ""
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"] + ("  grid " + _grid(r) if "lia_gemm" in r["Kernel_Name"] else "")) for r in csv.DictReader(open(kt))), key=lambda r: r[0])
marks = [i for i, r in enumerate(rows) if "lia_embed" in r[2]]      # one embed per forward
a, b = marks[-back - 1], marks[-back]
step = rows[a:b]
wall = step[-1][1] - step[0][0]
busy = defaultdict(lambda: [0, 0]); gap_after = defaultdict(lambda: [0, 0])
def short(n):
    g = n[n.index("  grid "):] if "  grid " in n else ""
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    return (n + g)[:64]
tot_gap = 0
for i, (s, e, n) in enumerate(step):
    k = short(n); busy[k][0] += 1; busy[k][1] += e - s
    if i + 1 < len(step):
        g = max(0, step[i + 1][0] - e); tot_gap += g; gap_after[k][0] += 1; gap_after[k][1] += g
print("step of %d kernels: wall %.3f ms, kernels %.3f ms, idle between kernels %.3f ms" % (len(step), wall / 1e6, sum(v[1] for v in busy.values()) / 1e6, tot_gap / 1e6))
for k, v in sorted(busy.items(), key=lambda kv: -kv[1][1]):
    ga = gap_after[k]
    print("  %-64s x%5d  avg %8.2f us  total %8.3f ms   idle after: avg %6.2f us" % (k, v[0], v[1] / v[0] / 1e3, v[1] / 1e6, (ga[1] / ga[0] / 1e3) if ga[0] else 0))
