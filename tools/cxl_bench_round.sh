#!/bin/bash
# tools/cxl_bench_round.sh <tag>: the DDR-vs-"CXL" host-tier H2D microbenchmark (twin of lia/cxl/benchmark.py + run.sh) on the
# GPU box: pinned DDR, the NUMA-interleaved tier registered with the driver (what --enable-cxl streams from) and unregistered
# (what the reference's numa_alloc tensors are: pageable), each alone and beside the 8192^3 fp32 CPU GEMMs.  Output is kept in
# gpurun_out/<tag>_cxl_benchmark.log (copy into results/).
set -u
tag=${1:-r03}
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONPATH=isca-2025-lia_amd
log=gpurun_out/${tag}_cxl_benchmark.log
: > "$log"
nodes=$(python3 -c "from lia_amd import hostinfo; print(','.join(str(n) for n in hostinfo.numa_nodes()[:2]))")
run() { echo "== python -m lia_amd.cxl.benchmark $*" | tee -a "$log"; timeout 600 python3 -m lia_amd.cxl.benchmark "$@" 2>&1 | grep -v amdgpu.ids | tee -a "$log"; }
run --gpu
run --gpu --cxl --register --nodes "$nodes"
run --gpu --cxl --nodes "$nodes"
run --gpu --cpu
run --gpu --cpu --cxl --register --nodes "$nodes"
