#!/bin/bash
# r06 GPU call 28: the final tree again (paired-blocks attention): whole GPU suite, smoke, the driver's bench line
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06/test_gpu_final3.txt 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r06/test_gpu_final3.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06/smoke_final3.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r06/smoke_final3.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_final3.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r06/bench_final3.log > gpurun_out/r06/bench_final3.json; cut -c1-330 gpurun_out/r06/bench_final3.json
