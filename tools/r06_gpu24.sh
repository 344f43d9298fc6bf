#!/bin/bash
# r06 GPU call 24: the prefill attention's output through the LDS (whole-row stores): bit-identity + time + tests
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
{ timeout 400 ./tools/attn_prefill_bench 5; timeout 200 python tools/attn_time.py; } > gpurun_out/r06/attn_bench7.txt 2>&1
echo "mismatches: $(grep -c MISMATCH gpurun_out/r06/attn_bench7.txt)"; grep -c "differing outputs 0 of" gpurun_out/r06/attn_bench7.txt; head -8 gpurun_out/r06/attn_bench7.txt; tail -6 gpurun_out/r06/attn_bench7.txt
echo "== without the stores"; timeout 300 ./tools/attn_prefill_bench_NOSTORE 5 2>&1 | grep -v "first mismatch" | head -5 | cut -c1-110
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_llama.py tests/test_gpu_generate.py -q -m gpu > gpurun_out/r06/test_attn7.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r06/test_attn7.txt
