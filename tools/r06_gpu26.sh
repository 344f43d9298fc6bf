#!/bin/bash
# r06 GPU call 26: two query blocks per workgroup (a block and its mirror image): bit-identity + time against one block per workgroup
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
{ timeout 400 ./tools/attn_prefill_bench 5; timeout 200 python tools/attn_time.py; } > gpurun_out/r06/attn_bench8.txt 2>&1
echo "mismatches: $(grep -c MISMATCH gpurun_out/r06/attn_bench8.txt)  identical shapes: $(grep -c 'differing outputs 0 of' gpurun_out/r06/attn_bench8.txt)"; head -8 gpurun_out/r06/attn_bench8.txt | cut -c1-175; tail -6 gpurun_out/r06/attn_bench8.txt
echo "== one block per workgroup (same binary otherwise)"; timeout 300 ./tools/attn_prefill_bench_nopairs 5 2>&1 | head -5 | cut -c1-110
