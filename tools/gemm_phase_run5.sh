#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/gemm_phase5.log
: > $out
echo "== CHECK 263 vs 256, M=16384" >> $out
CHECK=1 VARIANT=263 timeout 300 tools/gemm_bench 16384 0 >> $out 2>&1
echo "== CHECK 263 vs 256, M=1280" >> $out
CHECK=1 VARIANT=263 timeout 300 tools/gemm_bench 1280 0 >> $out 2>&1
echo "== CHECK 263 vs 256, M=8192 NORES" >> $out
NORES=1 CHECK=1 VARIANT=263 timeout 300 tools/gemm_bench 8192 0 >> $out 2>&1
for v in 262 263 262 263; do
  echo "== timing variant $v M=16384" >> $out
  timeout 300 tools/gemm_bench 16384 0 $v >> $out 2>&1
done
for v in 262 263; do
  echo "== timing variant $v M=16384 NORES NOBIAS" >> $out
  NORES=1 NOBIAS=1 timeout 300 tools/gemm_bench 16384 0 $v >> $out 2>&1
  echo "== timing variant $v M=8192" >> $out
  timeout 300 tools/gemm_bench 8192 0 $v >> $out 2>&1
done
grep -v "dummy\| 0 mismatches" $out
