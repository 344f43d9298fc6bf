#!/bin/bash
# tools/results_round.sh <tag>: every configuration of BASELINE.md section 4 through bench.py on the GPU box, one JSON line each
# under gpurun_out/results_<tag>/ (copy into results/ to keep them).
set -u
tag=${1:-r05}
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/results_$tag
mkdir -p "$out"
export LIA_STATE_DIR="$PWD/$out/state"      # the cooperative controller's converged counts (scheduler.CoopStore) stay on this box, for this round
run() { name=$1; shift; echo "== $name: bench.py $*"; timeout 1500 python3 bench.py "$@" > "$out/$name.log" 2>&1; tail -1 "$out/$name.log" > "$out/$name.json"; cut -c1-220 "$out/$name.json"; echo; }
run opt30b_gpu10_p0p2_pack10
# the same command again in a NEW process: the cooperative legs start on the count the first one converged on (seeded_from_store)
run opt30b_gpu10_p0p2_pack10_second_process --no-raw-leg --no-cpu-baseline
run opt30b_gpu10_p0p2_mb2_pack10 --num-minibatch 2 --no-raw-leg --no-cpu-baseline
run opt30b_gpu10_p3p3_pack10 --prefill-policy 3 --decoding-policy 3 --no-raw-leg --no-cpu-baseline
run opt30b_gpu100_resident --gpu-percentage 100 --no-raw-leg --no-cpu-baseline
run opt30b_gpu10_p0p2_pack10_trained_like --init trained-like --no-cpu-baseline --no-cooperative-leg --no-defer-kv-leg
# cooperative split: the online controller (-1) beside a scan of fixed counts on the SAME box (r03: within 5 % of the best fixed count)
X="--no-raw-leg --no-cpu-baseline --no-cooperative-leg"
run opt30b_gpu10_p0p2_pack10_cpu_online --cpu-layers -1 $X
for c in 19 21 23; do run opt30b_gpu10_p0p2_pack10_cpu$c --cpu-layers $c --steps 12 $X; done
run opt30b_gpu10_p3p3_pack10_cpu_online --prefill-policy 3 --decoding-policy 3 --cpu-layers -1 $X
for c in 21 23 25; do run opt30b_gpu10_p3p3_pack10_cpu$c --prefill-policy 3 --decoding-policy 3 --cpu-layers $c --steps 12 $X; done
run llama3_8b_gpu100_b128_t1024_n128 --model llama-3-8b --gpu-percentage 100 --batch 128 --prompt 1024 --steps 127
run opt66b_gpu5_cxl_pack10 --model opt-66b --gpu-percentage 5 --enable-cxl --cxl-nodes 0,1 --batch 32 --no-raw-leg --no-cpu-baseline
# data-parallel dry runs on the one GPU of the box: the line's schema for N > 1 (two ranks share the GPU over gloo), and real RCCL at world size 1
[ -n "${SKIP_DP2:-}" ] || run dp2_same_gpu_gloo_opt30b_gb64 --dp-same-gpu --dp-backend gloo --gpus 2 --global-batch 64 --steps 8 --warmup 1 --dp-allgather-legs --dp-extra-timeout 900
# four ranks on the one GPU, a global batch that the ranks do not divide (17 + 17 + 16 + 16 rows), 4 host threads each
run dp4_same_gpu_gloo_opt30b_gb66 --dp-same-gpu --dp-backend gloo --gpus 4 --global-batch 66 --steps 4 --warmup 1 --no-dp-extra-legs
[ -n "${SKIP_DP2:-}" ] || run dp1_rccl_world1_opt30b --force-dp --steps 8 --warmup 1 --no-raw-leg --no-cpu-baseline --no-cooperative-leg
run opt175b_gpu5_cxl_pack10_uniform01 --model opt-175b --gpu-percentage 5 --enable-cxl --cxl-nodes 0,1 --batch 32 --init uniform01 --no-raw-leg --no-cpu-baseline
