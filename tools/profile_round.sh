#!/bin/bash
# Collect the per-round rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01 [extra bench.py args]
# pass 1: --kernel-trace --stats (kernel durations); pass 2/3: one PMC counter each (FETCH_SIZE, WRITE_SIZE),
# never combined with a trace domain. Summaries land in gpurun_out/<tag>_*; copy them into profiles/.
set -u
tag=${1:-r01}; shift || true
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline "$@" > "$out/kt.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/$c" -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$out/$c.log" 2>&1
done
python3 tools/pmc_summary.py "$out" "$tag" "$@"
