#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/gemm_phase8.log
: > $out
for v in 256 262 262; do
  echo "== timing variant $v M=16384" >> $out
  timeout 300 tools/gemm_bench 16384 0 $v >> $out 2>&1
done
echo "== timing variant 262 M=16384 NORES" >> $out
NORES=1 timeout 300 tools/gemm_bench 16384 0 262 >> $out 2>&1
echo "== timing variant 262 M=8192" >> $out
timeout 300 tools/gemm_bench 8192 0 262 >> $out 2>&1
echo "== stamps 262" >> $out
T4STAMPS=262 timeout 300 tools/gemm_bench_stamps 16384 0 262 >> $out 2>&1
grep -v "dummy\| 0 mismatches" $out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_properties.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
LIA_GEMM_TILED_VARIANT=262 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_properties.py tests/test_gpu_llama.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
