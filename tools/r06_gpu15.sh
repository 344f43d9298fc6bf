#!/bin/bash
# r06 GPU call 15: SQ counters of the d = 128 prefill attention (tools/pmc_attn.sh: OPT-30B T 256, Llama quarter batch, T 2016 / 1792 shapes)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
bash tools/pmc_attn.sh > gpurun_out/r06/pmc_attn.txt 2>&1
grep -c . gpurun_out/r06/pmc_attn.txt
rm -rf gpurun_out/pmc_attn
