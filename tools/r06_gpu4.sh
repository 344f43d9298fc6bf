#!/bin/bash
# r06 fourth GPU call: the repaired tests, bench.py as the driver runs it, Llama GEMMs warm vs cold, matrix batch B (cxl_offloading.sh)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_fused_combine.py tests/test_gpu_generate.py tests/test_gpu_llama.py -q -m gpu > gpurun_out/r06/test_fix4.txt 2>&1; echo "tests rc=$?" > gpurun_out/r06/summary4.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_default.log 2>&1; echo "bench rc=$?" >> gpurun_out/r06/summary4.txt
tail -1 gpurun_out/r06/bench_default.log > gpurun_out/r06/bench_default.json
{ echo "== cold"; LLAMA=1 NOBIAS=1 NORES=1 timeout 120 ./tools/gemm_bench 128; echo "== warm (same buffer every launch: Infinity-Cache hits)"; WARM=1 LLAMA=1 NOBIAS=1 NORES=1 timeout 120 ./tools/gemm_bench 128; } > gpurun_out/r06/llama_gemm_warm.txt 2>&1
timeout 2700 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --only '^cxl_' --timeout 700 --budget-s 2400 > gpurun_out/r06/matrix4.txt 2>&1
tail -n 6 gpurun_out/r06/test_fix4.txt; cat gpurun_out/r06/summary4.txt gpurun_out/r06/llama_gemm_warm.txt gpurun_out/r06/matrix4.txt; cut -c1-600 gpurun_out/r06/bench_default.json
