"""tools/prefill_breakdown.py <rocprofv3 --kernel-trace output dir>: where the timed PREFILL of a bench.py run spends its time.
Takes the last run of >= 150 consecutive tiled-GEMM launches (one prefill), and prints for the window from its first kernel to
the argmax behind it: per kernel name the launches / total / average on the compute queue, the idle time of that queue (gaps
between consecutive kernels), what ran on the other queues meanwhile, and one mid layer kernel by kernel."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
kt = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(kt)))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
kern = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r[qkey]) for r in rows), key=lambda k: k[0])
tiled = [i for i, k in enumerate(kern) if "lia_gemm_tiled" in k[2]]
# prefills = maximal runs of tiled launches with < 100 ms between them
runs, cur = [], [tiled[0]]
for i in tiled[1:]:
    if kern[i][0] - kern[cur[-1]][1] < 100e6:
        cur.append(i)
    else:
        runs.append(cur)
        cur = [i]
runs.append(cur)
runs = [r for r in runs if len(r) >= 150]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
run = runs[which]
cq = kern[run[0]][3]
on_q = [k for k in kern if k[3] == cq]
# window: from the last non-tiled kernel before the first GEMM that starts within 5 ms (embed / LN1) to the first argmax behind the last GEMM
t_first, t_last = kern[run[0]][0], kern[run[-1]][1]
start = min((k[0] for k in on_q if t_first - 5e6 < k[0] <= t_first), default=t_first)
end = next((k[1] for k in on_q if k[0] > t_last and "argmax" in k[2]), t_last)
win = [k for k in on_q if start <= k[0] <= end]
print(f"prefill #{which} of {len(runs)}: {len(run)} tiled GEMM launches, compute-queue window {(end - start) / 1e6:.1f} ms")
agg = defaultdict(lambda: [0, 0])
for k in win:
    a = agg[k[2].split("(")[0][:70]]
    a[0] += 1
    a[1] += k[1] - k[0]
busy = sum(a[1] for a in agg.values())
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {t / 1e6:9.2f} ms  {n:5d} x {t / n / 1e3:9.1f} us  {name}")
gaps = [b[0] - a[1] for a, b in zip(win, win[1:]) if b[0] > a[1]]
print(f"  busy {busy / 1e6:.1f} ms, idle {sum(gaps) / 1e6:.1f} ms in {len(gaps)} gaps (largest {max(gaps) / 1e3:.0f} us, {sum(1 for g in gaps if g > 50e3)} over 50 us: "
      f"{sum(g for g in gaps if g > 50e3) / 1e6:.1f} ms)")
other = defaultdict(lambda: [0, 0])
for k in kern:
    if k[3] != cq and start <= k[0] <= end:
        a = other[(k[3], k[2].split("(")[0][:60])]
        a[0] += 1
        a[1] += k[1] - k[0]
for (q, name), (n, t) in sorted(other.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  other queue {q}: {t / 1e6:9.2f} ms  {n:5d} x {t / n / 1e3:9.1f} us  {name}")
# the four GEMMs of a layer (q|k|v, out, fc1, fc2 in launch order), with / without a kernel of another queue running beside them
others_iv = [(k[0], k[1]) for k in kern if k[3] != cq and start <= k[0] <= end]
names = ["q|k|v", "out", "fc1", "fc2"]
per = defaultdict(lambda: [[], []])
for j, i in enumerate(run[:4 * (len(run) // 4)]):
    k = kern[i]
    ov = sum(max(0, min(k[1], b_) - max(k[0], a_)) for a_, b_ in others_iv)
    per[names[j % 4]][1 if ov > 0 else 0].append((k[1] - k[0], ov))
for nm in names:
    alone, beside = per[nm]
    fa = (sum(t for t, _ in alone) / len(alone) / 1e3) if alone else float("nan")
    fb = (sum(t for t, _ in beside) / len(beside) / 1e3) if beside else float("nan")
    fo = (sum(o for _, o in beside) / len(beside) / 1e3) if beside else 0.0
    print(f"  GEMM {nm:6s}: alone {len(alone):3d} x {fa:8.1f} us   beside another queue's kernel {len(beside):3d} x {fb:8.1f} us (overlap {fo:6.1f} us)")
# one mid layer: between the 80th and 84th tiled launch
a, b = kern[run[80]][0], kern[run[84]][1]
print("one layer, kernel by kernel (start ms / dur us / gap-before us / name; * = other queue):")
prev = None
for k in kern:
    if a - 2e6 <= k[0] <= b:
        mine = k[3] == cq
        gap = (k[0] - prev) / 1e3 if (prev and mine) else 0.0
        print(f"   {(k[0] - a) / 1e6:8.3f} {(k[1] - k[0]) / 1e3:9.1f} {gap:8.1f} {'' if mine else '*'} {k[2].split('(')[0][:60]}")
        if mine:
            prev = k[1]
