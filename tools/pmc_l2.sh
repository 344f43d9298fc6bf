#!/bin/bash
# tools/pmc_l2.sh <tag> [bench.py args]: L2-side counters of one bench.py command (separate pass, no trace domain): requests from the
# CUs' vector L1s to L2 and L2's hits / misses -- how many bytes the CUs pulled out of L2 per launch beside the bytes L2 pulled
# out of HBM (tools/profile_round.sh's FETCH_SIZE).  Output: gpurun_out/<tag>_pmc_l2.csv (per-kernel averages).
set -u
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/prof_${tag}_l2
mkdir -p "$out"
timeout 900 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d "$out" -o pmc -- python3 bench.py --steps 2 --no-cpu-baseline --no-raw-leg --no-cooperative-leg --no-defer-kv-leg "$@" > "$out/run.log" 2>&1
tail -2 "$out/run.log" | cut -c1-200
python3 - "$out" "$tag" <<'PY'
import csv, glob, os, re, sys
out, tag = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).strip()
        a = acc.setdefault(k, {})
        c = a.setdefault(row["Counter_Name"], [0.0, set()])
        c[0] += float(row["Counter_Value"]); c[1].add(row["Dispatch_Id"])
with open(f"gpurun_out/{tag}_pmc_l2.csv", "w") as fh:
    names = sorted({c for a in acc.values() for c in a})
    fh.write("kernel,calls," + ",".join(n + "_per_launch" for n in names) + "\n")
    for k, a in sorted(acc.items()):
        calls = max(len(v[1]) for v in a.values())
        fh.write(f"\"{k}\",{calls}," + ",".join(f"{a[n][0] / max(1, len(a[n][1])):.0f}" if n in a else "" for n in names) + "\n")
print(open(f"gpurun_out/{tag}_pmc_l2.csv").read()[:6000])
PY
