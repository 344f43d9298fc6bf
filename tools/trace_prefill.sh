#!/bin/bash
# tools/trace_prefill.sh <tag> <bench.py args...>: kernel trace of one bench.py command (--steps 8 --warmup 2), timeline of its timed PREFILL forward
set -u
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/trace_$tag
mkdir -p "$out"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$out/kt" -o kt -- python3 bench.py --steps 8 --warmup 2 --no-raw-leg --no-cpu-baseline "$@" > "$out/kt.log" 2>&1
tail -1 "$out/kt.log" | cut -c1-300
python3 tools/decode_timeline.py "$out/kt" 10 | tee "gpurun_out/${tag}_prefill_timeline.txt"
