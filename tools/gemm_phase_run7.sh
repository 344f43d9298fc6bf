#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/gemm_phase7.log
: > $out
for v in 260 262; do
echo "== CHECK $v vs 256, M=16384" >> $out
CHECK=1 VARIANT=$v timeout 300 tools/gemm_bench 16384 0 >> $out 2>&1
done
echo "== CHECK 262 vs 256, M=1280" >> $out
CHECK=1 VARIANT=262 timeout 300 tools/gemm_bench 1280 0 >> $out 2>&1
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
for d in 0 128 256; do
  echo "== variant 262 T4_DBG=$d fc1" >> $out
  T4_DBG=$d SHAPE=28672,7168 timeout 300 tools/gemm_bench_abl 16384 0 262 >> $out 2>&1
done
grep -v "dummy\| 0 mismatches" $out
