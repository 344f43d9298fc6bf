#!/bin/bash
# tools/trace_cmd.sh <tag> <bench.py args...>: rocprofv3 --kernel-trace of one bench.py command, then tools/decode_timeline.py on it
set -u
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/trace_$tag
mkdir -p "$out"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$out/kt" -o kt -- python3 bench.py "$@" > "$out/kt.log" 2>&1
tail -1 "$out/kt.log" | cut -c1-300
python3 tools/decode_timeline.py "$out/kt" 2 | tee "gpurun_out/${tag}_decode_timeline.txt"
