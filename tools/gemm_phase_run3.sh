#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/gemm_phase3.log
: > $out
echo "== stamps, variant 262, M=16384" >> $out
T4STAMPS=262 timeout 300 tools/gemm_bench_stamps 16384 0 262 >> $out 2>&1
echo "== stamps, variant 259, M=16384" >> $out
T4STAMPS=259 timeout 300 tools/gemm_bench_stamps 16384 0 259 >> $out 2>&1
for d in 0 64; do
  echo "== variant 262 T4_DBG=$d (64 = no epilogue) out + fc1" >> $out
  T4_DBG=$d SHAPE=7168,7168 timeout 300 tools/gemm_bench_abl 16384 0 262 >> $out 2>&1
  T4_DBG=$d SHAPE=28672,7168 timeout 300 tools/gemm_bench_abl 16384 0 262 >> $out 2>&1
done
echo "== NOBIAS NORES 262" >> $out
NOBIAS=1 NORES=1 timeout 300 tools/gemm_bench 16384 0 262 >> $out 2>&1
grep -v dummy $out
