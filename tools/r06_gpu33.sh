#!/bin/bash
# r06 GPU call 33: SQ counters of the final prefill attention kernel (tools/pmc_attn.sh)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
timeout 1500 bash tools/pmc_attn.sh > gpurun_out/r06/attn_pmc_final.txt 2>&1; echo "rc=$?"; grep -c "per launch" gpurun_out/r06/attn_pmc_final.txt; grep "lia_attn_prefill128_kernel<0> g917504\|lia_attn_prefill128_kernel<0> g114688\|kernel<0> g" gpurun_out/r06/attn_pmc_final.txt | head -40
