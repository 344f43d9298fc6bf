#!/bin/bash
# r06 GPU call 16: the three-loop sweep 2 of the prefill attention (no accumulator copies on the loop latch): bit-identity on every shape + time
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
{ timeout 400 ./tools/attn_prefill_bench 5; timeout 200 python tools/attn_time.py; } > gpurun_out/r06/attn_bench3.txt 2>&1
grep -c MISMATCH gpurun_out/r06/attn_bench3.txt; head -6 gpurun_out/r06/attn_bench3.txt; tail -6 gpurun_out/r06/attn_bench3.txt
