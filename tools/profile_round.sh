#!/bin/bash
# Collect the per-round rocprofv3 evidence for one bench.py configuration on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> [bench.py args]
# pass 1: --kernel-trace --stats (kernel durations); passes 2/3: one HBM counter each (FETCH_SIZE, WRITE_SIZE); pass 4: the
# matrix-core counters (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE).  PMC passes never carry a trace domain.  The program
# goes directly after `--` (python3 bench.py ...).  Summaries land in gpurun_out/<tag>_*; copy them into profiles/.
set -u
tag=${1:-r02}; shift || true
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p "$out"
common="--no-cpu-baseline --no-raw-leg --no-cooperative-leg --no-defer-kv-leg --no-auto-plan"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python3 bench.py --steps 8 $common "$@" > "$out/kt.log" 2>&1
tail -1 "$out/kt.log" | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d "$out/$c" -o pmc -- python3 bench.py --steps 2 $common "$@" > "$out/$c.log" 2>&1
done
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/MFMA" -o pmc -- python3 bench.py --steps 2 $common "$@" > "$out/MFMA.log" 2>&1
python3 tools/pmc_summary.py "$out" "$tag" "$@"
