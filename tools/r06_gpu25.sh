#!/bin/bash
# r06 GPU call 25: the final tree -- whole GPU suite, smoke, the driver's bench line, the rocprofv3 passes of the headline, the
# Llama-3-8B configuration and the 2016-token matrix line with the final prefill attention
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06 gpurun_out/final_r06b
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06/test_gpu_final2.txt 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r06/test_gpu_final2.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06/smoke_final2.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r06/smoke_final2.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_final2.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r06/bench_final2.log > gpurun_out/r06/bench_final2.json; cut -c1-330 gpurun_out/r06/bench_final2.json
timeout 1500 bash tools/profile_round.sh r06final > gpurun_out/r06/profile_final.txt 2>&1; tail -5 gpurun_out/r06/profile_final.txt
timeout 900 python bench.py --model llama-3-8b --gpu-percentage 100 --batch 128 --prompt 1024 --steps 127 > gpurun_out/final_r06b/llama3_8b.log 2>&1; tail -1 gpurun_out/final_r06b/llama3_8b.log > gpurun_out/final_r06b/llama3_8b.json; cut -c1-330 gpurun_out/final_r06b/llama3_8b.json
timeout 900 python tools/run_matrix.py --only 'offline_opt30b_2016_32_b64' --outdir gpurun_out/final_r06b --timeout 600 2>&1 | tail -3
