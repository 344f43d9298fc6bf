#!/bin/bash
# r06 GPU call 11: rocprofv3 kernel stats of the reference's long-prompt line (lia_offline.sh:15: --input-tokens 2016 --num-minibatch 8) through run.py
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/prof_r06_matrix_2016
mkdir -p "$out"
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python3 run.py --benchmark -m facebook/opt-30b --dtype bfloat16 --ipex --input-tokens 2016 --max-new-tokens 8 --batch-size 64 --token-latency --num-iter 2 --num-warmup 1 --greedy --prefill-policy 0 --decoding-policy 1 --num-minibatch 8 --gpu-percentage 0 --pin-weight > "$out/kt.log" 2>&1
f=$(find "$out/kt" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/r06_matrix_opt30b_2016_b64_kernel_stats.csv; head -12 "$f" | cut -c1-160; else tail -20 "$out/kt.log"; fi
tail -8 "$out/kt.log"
rm -rf "$out/kt"
