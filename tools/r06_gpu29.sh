#!/bin/bash
# r06 GPU call 29: the final tree (attention changes 8-9 in): whole GPU suite, smoke, the driver's bench line, kernel stats of the
# headline, Llama-3-8B, the 2016-token matrix line
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r06 gpurun_out/final_r06c
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06/test_gpu_final4.txt 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r06/test_gpu_final4.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06/smoke_final4.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r06/smoke_final4.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_final4.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r06/bench_final4.log > gpurun_out/r06/bench_final4.json; cut -c1-200 gpurun_out/r06/bench_final4.json
common="--no-cpu-baseline --no-raw-leg --no-cooperative-leg --no-defer-kv-leg --no-auto-plan"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_r06c/kt -o kt -- python3 bench.py --steps 8 $common > gpurun_out/final_r06c/kt.log 2>&1
cp $(find gpurun_out/final_r06c/kt -name '*kernel_stats.csv' | head -1) gpurun_out/final_r06c/opt30b_bench_kernel_stats.csv; rm -rf gpurun_out/final_r06c/kt; grep -i "attn_prefill" gpurun_out/final_r06c/opt30b_bench_kernel_stats.csv | cut -c1-60,200-260
timeout 900 python bench.py --model llama-3-8b --gpu-percentage 100 --batch 128 --prompt 1024 --steps 127 > gpurun_out/final_r06c/llama3_8b.log 2>&1; tail -1 gpurun_out/final_r06c/llama3_8b.log > gpurun_out/final_r06c/llama3_8b.json; cut -c1-200 gpurun_out/final_r06c/llama3_8b.json
timeout 900 python tools/run_matrix.py --only 'offline_opt30b_2016_32_b64' --outdir gpurun_out/final_r06c --timeout 600 2>&1 | tail -3
