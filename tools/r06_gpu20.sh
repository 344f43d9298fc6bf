#!/bin/bash
# r06 GPU call 20: K / V staging addresses in the global saddr form (uniform base + 32-bit lane offset): bit-identity + time + tests;
# and tools/pk_rate: what a packed fp32 instruction costs beside two single ones
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
timeout 120 ./tools/pk_rate > gpurun_out/r06/pk_rate.txt 2>&1; cat gpurun_out/r06/pk_rate.txt
{ timeout 400 ./tools/attn_prefill_bench 5; timeout 200 python tools/attn_time.py; } > gpurun_out/r06/attn_bench6.txt 2>&1
echo "mismatches: $(grep -c MISMATCH gpurun_out/r06/attn_bench6.txt)"; head -8 gpurun_out/r06/attn_bench6.txt; tail -6 gpurun_out/r06/attn_bench6.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_llama.py tests/test_gpu_generate.py -q -m gpu > gpurun_out/r06/test_attn6.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r06/test_attn6.txt
