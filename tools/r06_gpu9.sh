#!/bin/bash
# r06 GPU call 9: the -m gpu suite, smoke and the driver's bench command on the final tree
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06/test_all_gpu9.txt 2>&1; echo "gpu suite rc=$?" > gpurun_out/r06/summary9.txt
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r06/summary9.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_final.log 2>&1; echo "bench rc=$?" >> gpurun_out/r06/summary9.txt
tail -1 gpurun_out/r06/bench_final.log > gpurun_out/r06/bench_final.json
tail -n 4 gpurun_out/r06/test_all_gpu9.txt; cat gpurun_out/r06/summary9.txt; cut -c1-300 gpurun_out/r06/bench_final.json
