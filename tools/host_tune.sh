#!/bin/bash
# tools/host_tune.sh: the host (policy-1) layer's A/B knobs on the box's cores -- register block width (LIA_HOST_LINEAR_RN) x
# prefetch mode (LIA_HOST_LINEAR_PF) -- through tools/host_layer_bench.py, the configurations interleaved over three rounds so
# that box noise hits all of them; output under gpurun_out/host_tune/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/host_tune
mkdir -p "$out"
for round in 1 2 3; do
  for cfg in ${CFGS:-"4 0" "4 1" "6 1" "6 2"}; do
    set -- $cfg
    echo "== round $round RN=$1 PF=$2: $(LIA_HOST_LINEAR_RN=$1 LIA_HOST_LINEAR_PF=$2 python3 tools/host_layer_bench.py 2>&1 | grep pinned)"
  done
done | tee "$out/layer.txt"
LIA_HOST_LINEAR_RN=4 python3 tools/host_linear_bench.py 2>&1 | grep -v amdgpu.ids | tee "$out/linear_rn4.txt"
LIA_HOST_LINEAR_RN=6 python3 tools/host_linear_bench.py 2>&1 | grep -v amdgpu.ids | tee "$out/linear_rn6.txt"
