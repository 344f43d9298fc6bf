#!/bin/bash
# r06 GPU call 10: --auto-plan runs of the four OPT-175B lines with 256 new tokens (every refused line then has a measured counterpart)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
timeout 2700 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --auto-plan-only --only 'offline_opt175b_32_256|online_opt175b_32_256|online_opt175b_256_256|online_opt175b_1792_256' --timeout 640 --budget-s 2400 > gpurun_out/r06/matrix10.txt 2>&1
cat gpurun_out/r06/matrix10.txt
