#!/bin/bash
# one GPU call: bit-identity of the phased 256^2 kernels (variants 259-262) against the one-barrier kernel (256), timings,
# and the ablations of 259 / 262 (tools/gemm_bench_abl, T4_DBG)
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/gemm_phase.log
: > $out
for v in 260 261 262; do
  echo "== CHECK $v vs 256, M=16384" >> $out
  CHECK=1 VARIANT=$v timeout 300 tools/gemm_bench 16384 0 >> $out 2>&1
done
echo "== CHECK 262 vs 256, M=1280" >> $out
CHECK=1 VARIANT=262 timeout 300 tools/gemm_bench 1280 0 >> $out 2>&1
for v in 256 259 260 261 262 259 262; do
  echo "== timing variant $v M=16384" >> $out
  timeout 300 tools/gemm_bench 16384 0 $v >> $out 2>&1
done
for v in 259 262; do
  echo "== timing variant $v M=8192" >> $out
  timeout 300 tools/gemm_bench 8192 0 $v >> $out 2>&1
done
for d in 0 1 4 6 8; do
  echo "== ablation variant 262 T4_DBG=$d (1 no DMA, 2 no reads, 4 no MFMA, 8 no stagger) fc1 M=16384" >> $out
  T4_DBG=$d SHAPE=28672,7168 timeout 300 tools/gemm_bench_abl 16384 0 262 >> $out 2>&1
done
echo "== zero operands 262" >> $out
ZERO=1 timeout 300 tools/gemm_bench 16384 0 262 >> $out 2>&1
cat $out
