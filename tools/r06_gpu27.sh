#!/bin/bash
# r06 GPU call 27: the attention / generate / llama / full-size tests on the paired-blocks kernel
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_llama.py tests/test_gpu_generate.py tests/test_gpu_configs.py -q -m gpu > gpurun_out/r06/test_attn9.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r06/test_attn9.txt
