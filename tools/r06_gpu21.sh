#!/bin/bash
# r06 GPU call 21: where a step of the prefill attention goes: the kernel without its loads, without its arithmetic, without both
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
for v in NO_LOADS NO_MATH NO_BOTH; do echo "== $v"; timeout 300 ./tools/attn_prefill_bench_$v 5 2>&1 | head -5 | cut -c1-110; done > gpurun_out/r06/attn_experiments.txt
cat gpurun_out/r06/attn_experiments.txt
