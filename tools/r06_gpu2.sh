#!/bin/bash
# r06 second GPU call: the whole -m gpu suite on the new dispatch, the mid-M microbenchmark again, matrix batch A
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
{ for M in 300 384 385 640 900; do echo "== M=$M"; timeout 120 ./tools/gemm_bench $M; done; } > gpurun_out/r06/gemm_midm2.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r06/test_all_gpu.txt 2>&1; echo "gpu suite rc=$?" > gpurun_out/r06/summary2.txt
timeout 2700 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --only 'readme|offline_opt30b|online_opt30b' --timeout 900 --budget-s 2400 > gpurun_out/r06/matrix2.txt 2>&1
tail -n 8 gpurun_out/r06/test_all_gpu.txt; cat gpurun_out/r06/matrix2.txt gpurun_out/r06/summary2.txt
