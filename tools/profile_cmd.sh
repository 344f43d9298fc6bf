#!/bin/bash
# tools/profile_cmd.sh <tag> <bench.py args...>: rocprofv3 --kernel-trace --stats of one bench.py command on the GPU box;
# the kernel table lands in gpurun_out/<tag>_kernel_stats.csv (copy it into profiles/ to keep it).
set -u
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p "$out"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python3 bench.py "$@" > "$out/kt.log" 2>&1
tail -1 "$out/kt.log" | cut -c1-400
f=$(find "$out/kt" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" "gpurun_out/${tag}_kernel_stats.csv"; head -24 "$f" | cut -c1-200; else echo "no kernel stats produced"; tail -20 "$out/kt.log"; fi
