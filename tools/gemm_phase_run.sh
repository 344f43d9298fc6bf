#!/bin/bash
# One GPU call for the prefill-GEMM work (run through gpurun from the repo root, after building the three tools):
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I isca-2025-lia_amd/csrc tools/gemm_bench.hip -o tools/gemm_bench
#   ... -DLIA_GEMM_ABLATE ... -o tools/gemm_bench_abl        ... -DLIA_GEMM_STAMPS ... -o tools/gemm_bench_stamps
# 1. bit-identity of the phased kernels (variants 259-262) against the one-barrier kernel (256), incl. a ragged M;
# 2. timings, interleaved in one process each (M = 16384 and 8192; with and without the residual);
# 3. ablations of variant 262 (T4_DBG: 1 no LDS-DMA, 4 no MFMA, 6 LDS-DMA + barriers only, 8 no stagger, 64 no epilogue,
#    128 every piece re-reads K-tile 0, 256 waits tightened to two half-tiles in flight, 38 every workgroup on tile (0,0));
# 4. stamps: prologue / K loop / epilogue / gap per workgroup.
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/gemm_phase.log
: > $out
for v in 259 260 261 262; do echo "== CHECK $v vs 256, M=16384" >> $out; CHECK=1 VARIANT=$v timeout 300 tools/gemm_bench 16384 0 >> $out 2>&1; done
echo "== CHECK 262 vs 256, M=1280" >> $out; CHECK=1 VARIANT=262 timeout 300 tools/gemm_bench 1280 0 >> $out 2>&1
for v in 256 259 260 261 262 256 262; do echo "== timing variant $v M=16384" >> $out; timeout 300 tools/gemm_bench 16384 0 $v >> $out 2>&1; done
echo "== timing variant 262 M=16384 NORES" >> $out; NORES=1 timeout 300 tools/gemm_bench 16384 0 262 >> $out 2>&1
for v in 256 262; do echo "== timing variant $v M=8192" >> $out; timeout 300 tools/gemm_bench 8192 0 $v >> $out 2>&1; done
for d in 0 1 4 6 8 64 128 256 38; do echo "== ablation variant 262 T4_DBG=$d fc1 M=16384" >> $out; T4_DBG=$d SHAPE=28672,7168 timeout 300 tools/gemm_bench_abl 16384 0 262 >> $out 2>&1; done
echo "== zero operands 262" >> $out; ZERO=1 timeout 300 tools/gemm_bench 16384 0 262 >> $out 2>&1
echo "== stamps 262" >> $out; T4STAMPS=262 timeout 300 tools/gemm_bench_stamps 16384 0 262 >> $out 2>&1
grep -v "dummy\| 0 mismatches" $out
