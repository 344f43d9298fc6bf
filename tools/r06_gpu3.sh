#!/bin/bash
# r06 third GPU call: attention prologue A/B, the whole -m gpu suite, the matrix lines that were refused for the wrong reason
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
timeout 300 ./tools/attn_prefill_bench > gpurun_out/r06/attn_bench.txt 2>&1
timeout 300 python tools/attn_time.py >> gpurun_out/r06/attn_bench.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06/test_all_gpu3.txt 2>&1; echo "gpu suite rc=$?" > gpurun_out/r06/summary3.txt
timeout 2400 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --only 'readme|offline_opt30b_2016|offline_opt30b_1792' --timeout 1100 --budget-s 2000 > gpurun_out/r06/matrix3.txt 2>&1
tail -n 12 gpurun_out/r06/test_all_gpu3.txt; cat gpurun_out/r06/attn_bench.txt gpurun_out/r06/matrix3.txt gpurun_out/r06/summary3.txt
