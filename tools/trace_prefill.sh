#!/bin/bash
# tools/trace_prefill.sh <tag> <bench.py args...>: kernel trace of one bench.py command, breakdown of its timed PREFILL forward
set -u
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/trace_$tag
mkdir -p "$out"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$out/kt" -o kt -- python3 bench.py --steps 4 --warmup 1 --no-raw-leg --no-cpu-baseline --no-cooperative-leg --no-defer-kv-leg --no-auto-plan "$@" > "$out/kt.log" 2>&1
tail -1 "$out/kt.log" | cut -c1-300
python3 tools/prefill_breakdown.py "$out/kt" | tee "gpurun_out/${tag}_prefill_breakdown.txt"
rm -rf "$out/kt"
