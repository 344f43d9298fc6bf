#!/bin/bash
# tools/final_round.sh <tag>: the headline lines re-measured on the final tree (after results_round.sh): driver flags, protocol flags,
# immediate K/V delivery, all-resident, Llama-3-8B; then the rocprofv3 rounds of the headline and of Llama-3-8B.
set -u
tag=${1:-r03}
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/final_$tag
mkdir -p "$out"
run() { name=$1; shift; echo "== $name: $*"; timeout 1500 "$@" > "$out/$name.log" 2>&1; tail -1 "$out/$name.log" > "$out/$name.json"; cut -c1-200 "$out/$name.json"; echo; }
run bench_driver_flags python3 bench.py --steps 20 --warmup 5
run opt30b_gpu10_p0p2_pack10 python3 bench.py
LIA_DEFER_KV=0 run opt30b_gpu10_p0p2_pack10_immediate_kv python3 bench.py --no-raw-leg --no-cpu-baseline --no-cooperative-leg

run opt30b_gpu100_resident python3 bench.py --gpu-percentage 100 --no-raw-leg --no-cpu-baseline
run llama3_8b_gpu100_b128_t1024_n128 python3 bench.py --model llama-3-8b --gpu-percentage 100 --batch 128 --prompt 1024 --steps 127
tools/profile_round.sh ${tag}_opt30b --no-cooperative-leg 2>&1 | tail -2
tools/profile_round.sh ${tag}_llama3_8b --model llama-3-8b --gpu-percentage 100 --batch 128 --prompt 1024 2>&1 | tail -2
