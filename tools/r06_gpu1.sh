#!/bin/bash
# r06 first GPU call: mid-M GEMM microbenchmark (new dispatch vs r05's), the new parity tests, two matrix lines for timing
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
for M in 900 450 300; do
  echo "== new dispatch M=$M"; timeout 120 ./tools/gemm_bench $M
  echo "== r05 dispatch (128x128 kernel) M=$M"; timeout 120 ./tools/gemm_bench_old128 $M
done > gpurun_out/r06/gemm_midm.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_generate.py -q -x -k "ragged" -s > gpurun_out/r06/test_ragged.txt 2>&1; echo "ragged rc=$?" >> gpurun_out/r06/summary.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize_oracle.py -q -k "mid_m or t2016 or prefill_layer or m16384" -s > gpurun_out/r06/test_midm.txt 2>&1; echo "midm rc=$?" >> gpurun_out/r06/summary.txt
timeout 1500 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --only 'offline_opt30b_32_32_b64|offline_opt30b_32_32_b900' --timeout 600 > gpurun_out/r06/matrix1.txt 2>&1
tail -5 gpurun_out/r06/gemm_midm.txt gpurun_out/r06/test_ragged.txt gpurun_out/r06/test_midm.txt gpurun_out/r06/matrix1.txt gpurun_out/r06/summary.txt
