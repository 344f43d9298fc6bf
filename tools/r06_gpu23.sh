#!/bin/bash
# r06 GPU call 23: the prefill attention with its stages issued but never waited for (wrong results: is the exposed time latency or contention?)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
for v in SK_NOSTORE SK_NOQ SK_NONE NOSTORE NOQ; do echo "== $v"; timeout 300 ./tools/attn_prefill_bench_$v 5 2>&1 | grep -v "first mismatch" | head -5 | cut -c1-110; done > gpurun_out/r06/attn_experiments3.txt 2>/dev/null
cat gpurun_out/r06/attn_experiments3.txt
