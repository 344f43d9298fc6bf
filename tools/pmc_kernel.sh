#!/bin/bash
# tools/pmc_kernel.sh <kernel-substring> <bench.py args...>: SQ / TCP / TCC counters of one kernel of a short bench.py run, one small
# group per pass (PMC passes carry no trace domain); per-launch averages are printed and kept in gpurun_out/pmc_kernel/summary.txt
set -u
pat=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/pmc_kernel
mkdir -p "$out"
: > "$out/summary.txt"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-raw-leg --no-cpu-baseline "$@" > "$out/p$i.log" 2>&1
  f=$(find "$out/p$i" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$pat" <<'P' | tee -a "$out/summary.txt"
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c, v in acc.items():
    print(f"{sys.argv[2]:28s} {c:32s} per launch {v / n[c]:.5g}  (launches {n[c]})")
P
done
