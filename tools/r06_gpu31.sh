#!/bin/bash
# r06 GPU call 31: persistent workgroups in the d = 128 prefill attention: bit-identity + time against one workgroup per item
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
{ timeout 400 ./tools/attn_prefill_bench 5; timeout 200 python tools/attn_time.py; } > gpurun_out/r06/attn_bench11.txt 2>&1
echo "mismatches: $(grep -c MISMATCH gpurun_out/r06/attn_bench11.txt)  identical shapes: $(grep -c 'differing outputs 0 of' gpurun_out/r06/attn_bench11.txt)"; head -8 gpurun_out/r06/attn_bench11.txt | cut -c1-175; tail -6 gpurun_out/r06/attn_bench11.txt
echo "== one workgroup per item (same source)"; timeout 300 ./tools/attn_prefill_bench_nopersist 5 2>&1 | head -5 | cut -c1-110
