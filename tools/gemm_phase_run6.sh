#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/gemm_phase6.log
: > $out
for d in 0 1 64 128 192 256 0; do
  echo "== variant 262 T4_DBG=$d (1 no DMA, 64 no epilogue, 128 every piece re-reads K-tile 0, 192 = 128 + 64, 256 tight waits vmcnt 4/2) fc1 + fc2" >> $out
  T4_DBG=$d SHAPE=28672,7168 timeout 300 tools/gemm_bench_abl 16384 0 262 >> $out 2>&1
  T4_DBG=$d SHAPE=7168,28672 timeout 300 tools/gemm_bench_abl 16384 0 262 >> $out 2>&1
done
grep -v "dummy" $out
