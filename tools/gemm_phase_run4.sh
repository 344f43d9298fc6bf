#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/gemm_phase4.log
: > $out
for sk in 0 3 6 10 20; do
  echo "== variant 262, XCD skew $sk us" >> $out
  LIA_GEMM_SKEW_US=$sk timeout 300 tools/gemm_bench 16384 0 262 >> $out 2>&1
done
for sk in 0 6; do
  echo "== stamps 262 skew $sk" >> $out
  LIA_GEMM_SKEW_US=$sk T4STAMPS=262 timeout 300 tools/gemm_bench_stamps 16384 0 262 >> $out 2>&1
done
echo "== M=8192 skew 0 / 6" >> $out
LIA_GEMM_SKEW_US=0 timeout 300 tools/gemm_bench 8192 0 262 >> $out 2>&1
LIA_GEMM_SKEW_US=6 timeout 300 tools/gemm_bench 8192 0 262 >> $out 2>&1
grep -v dummy $out
