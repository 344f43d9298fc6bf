#!/bin/bash
# tools/pmc_decode_attn.sh [bench.py args]: SQ counters of lia_attn_decode_kernel inside one bench.py command, one small group per pass
# (no trace domain beside --pmc): how busy the VALU / LDS / memory pipes of a CU are while the kernel streams the K/V rows
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/pmc_decode_attn
mkdir -p "$out"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -o pmc -- python3 bench.py --steps 2 --no-cpu-baseline --no-raw-leg --no-cooperative-leg --no-defer-kv-leg "$@" > "$out/p$i.log" 2>&1
  f=$(find "$out/p$i" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:44]
    if "attn_decode" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in acc:
    for c, v in acc[k].items():
        print(f"{k:46s} {c:28s} per launch {v / n[(k, c)]:.5g}  (launches {n[(k, c)]})")
P
  rm -rf "$out/p$i"
done
