#!/bin/bash
# r06 GPU call 7: the whole -m gpu suite + smoke on the final tree; the matrix lines whose records predate the final record format
# (host-layer rows, per-run memory) or the attention change; the planner's batch-1050 pick again
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06/test_all_gpu7.txt 2>&1; echo "gpu suite rc=$?" > gpurun_out/r06/summary7.txt
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r06/summary7.txt 2>&1
timeout 1500 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --only 'online_opt30b_.*_p11|offline_opt30b_32_32_b900|offline_opt30b_2016_32' --timeout 600 > gpurun_out/r06/matrix7.txt 2>&1
timeout 900 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --auto-plan-only --only 'cxl_opt30b_32_256_b1050' --timeout 800 >> gpurun_out/r06/matrix7.txt 2>&1
tail -n 5 gpurun_out/r06/test_all_gpu7.txt; cat gpurun_out/r06/summary7.txt gpurun_out/r06/matrix7.txt
