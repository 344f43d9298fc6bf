#!/bin/bash
# r06 fifth GPU call: the 0 / 1 lines again (packed + raw copies), then --auto-plan runs of the lines this box refuses and of the headline lines
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
timeout 1500 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --only 'offline_opt30b_32_32_b64|offline_opt30b_2016_32|offline_opt30b_32_256_b64|online_opt30b_2016_32|online_opt30b_1792_256' --timeout 600 > gpurun_out/r06/matrix5.txt 2>&1
timeout 1200 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --only 'opt175b' --timeout 120 >> gpurun_out/r06/matrix5.txt 2>&1
timeout 3600 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --auto-plan-only --only 'readme|offline_opt30b_32_32_b900|offline_opt30b_32_256_b900|cxl_opt30b_32_128_b1150|cxl_opt30b_32_256_b1050|offline_opt175b_32_32|online_opt175b_32_32|online_opt175b_256_32|online_opt175b_2016_32' --timeout 900 --budget-s 3000 >> gpurun_out/r06/matrix5.txt 2>&1
cat gpurun_out/r06/matrix5.txt
