#!/bin/bash
# r06 GPU call 17: the -m gpu suite + smoke + the driver's bench command + Llama-3-8B + the 2016-token matrix line on the tree with the three-loop attention
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06 gpurun_out/final_r06
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06/test_all_gpu17.txt 2>&1; echo "gpu suite rc=$?" > gpurun_out/r06/summary17.txt
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r06/summary17.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_final3.log 2>&1; echo "bench rc=$?" >> gpurun_out/r06/summary17.txt
tail -1 gpurun_out/r06/bench_final3.log > gpurun_out/final_r06/final3_bench_driver_flags.json
timeout 900 python bench.py --model llama-3-8b --gpu-percentage 100 --batch 128 --prompt 1024 --steps 127 > gpurun_out/r06/bench_llama3.log 2>&1; tail -1 gpurun_out/r06/bench_llama3.log > gpurun_out/final_r06/final3_llama3_8b_gpu100_b128_t1024_n128.json
timeout 900 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --only 'offline_opt30b_2016_32' --timeout 600 > gpurun_out/r06/matrix17.txt 2>&1
tail -n 3 gpurun_out/r06/test_all_gpu17.txt; cat gpurun_out/r06/summary17.txt gpurun_out/r06/matrix17.txt
python - <<'P'
import json
for n in ("final3_bench_driver_flags", "final3_llama3_8b_gpu100_b128_t1024_n128"):
    d = json.load(open(f"gpurun_out/final_r06/{n}.json")); print(n, round(d["value"], 2), round(d["prefill_ms"], 1), round(d["ms_per_step"], 3))
P
