"""Summarise a rocprofv3 --kernel-trace --memory-copy-trace run of bench.py: weight-copy intervals vs GEMM intervals."""
import csv, glob, sys
d = sys.argv[1]
kt = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
mt = glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True)[0]
copies = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"]) for r in csv.DictReader(open(mt))]
kern = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt))]
tiled = [k for k in kern if "tiled" in k[2]]
t_first, t_last = tiled[-192][0], tiled[-1][1]
print("last prefill: GEMM window %.1f ms, sum of GEMM %.1f ms" % ((t_last - t_first) / 1e6, sum(k[1] - k[0] for k in tiled[-192:]) / 1e6))
lo, hi = t_first - 150e6, t_last + 150e6
big = [c for c in copies if c[1] - c[0] > 8e6 and "HOST_TO_DEVICE" in c[2] and lo < c[0] < hi]
t0 = min(t_first, big[0][0])
prev = None
for i, c in enumerate(big):
    gap = (c[0] - prev) / 1e6 if prev else 0.0
    if i < 7 or i > len(big) - 4 or gap > 1.0:
        print("  H2D %2d start %8.1f dur %6.2f ms gap-before %6.2f" % (i, (c[0] - t0) / 1e6, (c[1] - c[0]) / 1e6, gap))
    prev = c[1]
print("H2D big copies: %d, sum %.1f ms, span %.1f..%.1f" % (len(big), sum(c[1] - c[0] for c in big) / 1e6, (big[0][0] - t0) / 1e6, (big[-1][1] - t0) / 1e6))
d2h = [c for c in copies if "DEVICE_TO_HOST" in c[2] and c[1] - c[0] > 1e6 and lo < c[0] < hi]
if d2h:
    print("D2H copies >1ms: %d, sum %.1f ms, first start %.1f last end %.1f" % (len(d2h), sum(c[1] - c[0] for c in d2h) / 1e6, (d2h[0][0] - t0) / 1e6, (d2h[-1][1] - t0) / 1e6))
print("first GEMM start %.1f, last GEMM end %.1f" % ((t_first - t0) / 1e6, (t_last - t0) / 1e6))
others = [k for k in kern if t_last < k[0] < t_last + 80e6]
for k in others[:12]:
    print("   after: %-40s start %.2f dur %.3f ms" % (k[2][:40], (k[0] - t0) / 1e6, (k[1] - k[0]) / 1e6))
