#!/bin/bash
# tools/profile_decode.sh <tag> <bench.py args...>: rocprofv3 --kernel-trace --stats of one bench.py command (8 timed decode steps);
# leaves gpurun_out/<tag>_kernel_stats.csv (the per-kernel summary for profiles/) and gpurun_out/<tag>_decode_timeline.txt
# (one decode step: kernels and the idle gaps between them).  PMC passes are separate (tools/pmc_kernel.sh).
set -u
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p "$out"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python3 bench.py --steps 8 --no-cpu-baseline --no-raw-leg --no-cooperative-leg --no-defer-kv-leg --no-auto-plan "$@" > "$out/kt.log" 2>&1
tail -1 "$out/kt.log" | cut -c1-260
cp "$(find "$out/kt" -name '*kernel_stats.csv' | head -1)" "gpurun_out/${tag}_kernel_stats.csv"
python3 tools/decode_timeline.py "$out/kt" 2 | tee "gpurun_out/${tag}_decode_timeline.txt"
rm -rf "$out/kt"
