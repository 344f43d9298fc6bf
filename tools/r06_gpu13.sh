#!/bin/bash
# r06 GPU call 13: the CPU-only large-batch line (ipex_offline.sh: batch 900, 32 / 32) with a time limit that fits it (~20 min of host compute)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
timeout 2300 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --only 'ipexoffline_opt30b_32_32_b900' --timeout 2200 > gpurun_out/r06/matrix13.txt 2>&1
cat gpurun_out/r06/matrix13.txt
