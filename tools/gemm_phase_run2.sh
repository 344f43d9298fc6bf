#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/gemm_phase2.log
: > $out
for g in 1 2 8 16; do
  for d in 0 6; do
    echo "== GM=$g variant 262 T4_DBG=$d fc1" >> $out
    T4_DBG=$d SHAPE=28672,7168 timeout 300 tools/gemm_bench_gm$g 16384 0 262 >> $out 2>&1
  done
  echo "== GM=$g variant 262 all shapes" >> $out
  timeout 300 tools/gemm_bench_gm$g 16384 0 262 >> $out 2>&1
done
for d in 0 6 38 32; do
  echo "== GM=4 variant 262 T4_DBG=$d (38 = DMA only, every workgroup on tile 0,0; 32 = full kernel on tile 0,0) fc1" >> $out
  T4_DBG=$d SHAPE=28672,7168 timeout 300 tools/gemm_bench_abl 16384 0 262 >> $out 2>&1
done
echo "== KSWEEP 262" >> $out
KSWEEP=1 timeout 300 tools/gemm_bench 16384 0 262 >> $out 2>&1
grep -v dummy $out
