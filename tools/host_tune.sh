#!/bin/bash
# tools/host_tune.sh: the host (policy-1) linear's A/B knobs on the box's cores -- prefetch mode x K-chunk -- through
# tools/host_linear_bench.py and tools/host_layer_bench.py; output under gpurun_out/host_tune/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/host_tune
mkdir -p "$out"
for pf in 0 1 2; do
  for kc in 2048; do
    echo "== PF=$pf KC=$kc"
    LIA_HOST_LINEAR_PF=$pf LIA_HOST_LINEAR_KC=$kc python3 tools/host_linear_bench.py 2>&1 | grep -v amdgpu.ids
  done
done | tee "$out/linear.txt"
for pf in 0 1 2; do
  echo "== layer PF=$pf"
  LIA_HOST_LINEAR_PF=$pf python3 tools/host_layer_bench.py 2>&1 | grep -v amdgpu.ids
done | tee "$out/layer.txt"
