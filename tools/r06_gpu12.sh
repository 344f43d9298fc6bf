#!/bin/bash
# r06 GPU call 12: the CPU-only baseline lines of the reference's matrix (ipex_offline.sh / ipex_online.sh shapes: policies 1 / 1, gpu% 0)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
timeout 2700 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --only '^ipex' --timeout 900 --budget-s 2400 > gpurun_out/r06/matrix12.txt 2>&1
cat gpurun_out/r06/matrix12.txt
