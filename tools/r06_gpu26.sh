#!/bin/bash
# r06 GPU call 26: two query blocks per workgroup (a block and its mirror image): bit-identity + time against one block per workgroup
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
{ timeout 400 ./tools/attn_prefill_bench 5; timeout 200 python tools/attn_time.py; } > gpurun_out/r06/attn_bench10.txt 2>&1
echo "mismatches: $(grep -c MISMATCH gpurun_out/r06/attn_bench10.txt)  identical shapes: $(grep -c 'differing outputs 0 of' gpurun_out/r06/attn_bench10.txt)"; head -8 gpurun_out/r06/attn_bench10.txt | cut -c1-175; tail -6 gpurun_out/r06/attn_bench10.txt

