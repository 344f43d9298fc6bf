#!/bin/bash
# counters of the prefill attention kernels (tools/attn_prefill_bench: product kernel and candidate), one small group per pass
# usage (GPU box): tools/pmc_attn.sh          env: ABL / REMAP / LAYOUT as tools/attn_prefill_bench reads them
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp QUICK=1
out=gpurun_out/pmc_attn
mkdir -p "$out"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_TRANS_F32"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -o pmc -- tools/attn_prefill_bench 3 > "$out/p$i.log" 2>&1
  f=$(find "$out/p$i" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:34] + " g" + r.get("Grid_Size", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in sorted(acc):
    if "attn" not in k: continue
    for c, v in acc[k].items():
        print(f"{k:50s} {c:32s} per launch {v / n[(k, c)]:.4g}  (launches {n[(k, c)]})")
P
done
