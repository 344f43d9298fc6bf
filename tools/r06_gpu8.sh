#!/bin/bash
# r06 GPU call 8: the workspace-sizing fix and the dual-copy test fix through the suite; the planner's large-batch picks again
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06/test_all_gpu8.txt 2>&1; echo "gpu suite rc=$?" > gpurun_out/r06/summary8.txt
timeout 2400 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --auto-plan-only --only 'cxl_opt30b_32_256_b1050|cxl_opt30b_32_128_b1150|offline_opt30b_32_256_b900' --timeout 760 > gpurun_out/r06/matrix8.txt 2>&1
tail -n 4 gpurun_out/r06/test_all_gpu8.txt; cat gpurun_out/r06/summary8.txt gpurun_out/r06/matrix8.txt
