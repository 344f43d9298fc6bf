#!/bin/bash
# r06 GPU call 30: tests/test_gpu_ops.py (the new launcher-guard test in it)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu > gpurun_out/r06/test_ops_guard.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r06/test_ops_guard.txt
