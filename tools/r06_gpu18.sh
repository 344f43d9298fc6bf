#!/bin/bash
# r06 GPU call 18: the four-loop prefill attention (unmasked fast path for tiles wholly below the diagonal): bit-identity on every shape + time + tests
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
{ timeout 400 ./tools/attn_prefill_bench 5; timeout 200 python tools/attn_time.py; } > gpurun_out/r06/attn_bench4.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_llama.py tests/test_gpu_generate.py -q -m gpu > gpurun_out/r06/test_attn4.txt 2>&1; echo "tests rc=$?"
grep -c MISMATCH gpurun_out/r06/attn_bench4.txt; head -6 gpurun_out/r06/attn_bench4.txt; tail -6 gpurun_out/r06/attn_bench4.txt; tail -3 gpurun_out/r06/test_attn4.txt
