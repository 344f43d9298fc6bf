"""Reduce the rocprofv3 output of tools/profile_round.sh to the files kept under profiles/:
<tag>_bench_kernel_stats.csv (the --stats kernel table), <tag>_bench_pmc_hbm.json (per-kernel HBM bytes per launch,
FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section prescribes for gfx950) and <tag>_bench_pmc_mfma.json (matrix-core
busy cycles per launch against the launch's own GPU-active cycles)."""
import csv, glob, json, os, re, shutil, sys

out, tag = sys.argv[1], sys.argv[2]
extra = " ".join(sys.argv[3:])
dst = os.path.join("gpurun_out", f"{tag}_bench")
stats = glob.glob(os.path.join(out, "kt", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], dst + "_kernel_stats.csv")


def short(name):
    return re.sub(r"\(.*", "", name).strip()


kern = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, ctr, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != ctr:
                    continue
                k = short(row["Kernel_Name"])
                a = acc.setdefault(k, [0.0, set()])
                a[0] += float(row["Counter_Value"])
                a[1].add(row["Dispatch_Id"])
        for k, (tot, ids) in acc.items():
            d = kern.setdefault(k, {})
            d[ctr + "_KiB_avg"] = tot / max(1, len(ids))
            d["calls"] = len(ids)
for k, d in kern.items():
    if "FETCH_SIZE_KiB_avg" in d and "WRITE_SIZE_KiB_avg" in d:
        d["traffic_bytes_per_launch"] = (2 * d["FETCH_SIZE_KiB_avg"] + d["WRITE_SIZE_KiB_avg"]) * 1024
json.dump({
    "command": f"rocprofv3 --pmc <CTR> --output-format csv -- python3 bench.py --steps 2 --no-cpu-baseline --no-raw-leg {extra} "
               "(one counter per pass: FETCH_SIZE, then WRITE_SIZE)",
    "note": "FETCH_SIZE/WRITE_SIZE are KiB. On gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read "
            "(MI355X_MICROARCH.md, HBM section) -> traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch",
    "kernels": kern}, open(dst + "_pmc_hbm.json", "w"), indent=1)
# matrix-core utilisation: SQ_VALU_MFMA_BUSY_CYCLES (summed over every SIMD of the chip) against GRBM_GUI_ACTIVE (summed over the
# 8 XCDs: / 8 = the launch's cycles, MI355X_MICROARCH.md "DVFS give-back") x 256 CUs x 4 SIMDs
mf = {}
for f in glob.glob(os.path.join(out, "MFMA", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row["Kernel_Name"])
            d = per.setdefault((k, row["Dispatch_Id"]), {})
            d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    for (k, _), d in per.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("GRBM_GUI_ACTIVE", 0) > 0:
            a = mf.setdefault(k, {"calls": 0, "mfma_busy_cycles": 0.0, "gui_active": 0.0})
            a["calls"] += 1
            a["mfma_busy_cycles"] += d["SQ_VALU_MFMA_BUSY_CYCLES"]
            a["gui_active"] += d["GRBM_GUI_ACTIVE"]
for k, a in mf.items():
    a["mfma_busy_cycles_per_launch"] = a["mfma_busy_cycles"] / a["calls"]
    a["launch_cycles"] = a["gui_active"] / a["calls"] / 8.0
    a["mfma_util"] = a["mfma_busy_cycles"] / (a["gui_active"] / 8.0 * 256 * 4)
if mf:
    json.dump({"command": f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -- python3 bench.py --steps 2 --no-cpu-baseline --no-raw-leg {extra}",
               "note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): the share of SIMD cycles in which the matrix "
                       "pipe was busy while the kernel ran (counters are serialised per dispatch by the profiler; launch_cycles / duration = clock)",
               "kernels": {k: v for k, v in mf.items() if v["mfma_busy_cycles"] > 0}}, open(dst + "_pmc_mfma.json", "w"), indent=1)
print("wrote", dst + "_kernel_stats.csv", dst + "_pmc_hbm.json", "kernels:", len(kern), "mfma kernels:", len([1 for v in mf.values() if v["mfma_busy_cycles"] > 0]))
