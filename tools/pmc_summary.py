"""Reduce the rocprofv3 output of tools/profile_round.sh to the two files kept under profiles/:
<tag>_bench_kernel_stats.csv (the --stats kernel table) and <tag>_bench_pmc_hbm.json (per-kernel HBM bytes per launch,
FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section prescribes for gfx950)."""
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
    "command": f"rocprofv3 --pmc <CTR> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline {extra} "
               "(one counter per pass: FETCH_SIZE, then WRITE_SIZE)",
    "note": "FETCH_SIZE/WRITE_SIZE are KiB. On gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read "
            "(MI355X_MICROARCH.md, HBM section) -> traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch",
    "kernels": kern}, open(dst + "_pmc_hbm.json", "w"), indent=1)
print("wrote", dst + "_kernel_stats.csv", dst + "_pmc_hbm.json", "kernels:", len(kern))
