#!/bin/bash
# tools/final_r06.sh: the bench.py configurations of BASELINE.md section 4 on the final r06 tree, one JSON line each under gpurun_out/final_r06/
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/final_r06
mkdir -p "$out"
export LIA_STATE_DIR="$PWD/$out/state"
run() { name=$1; shift; echo "== $name: bench.py $*"; timeout 1200 python3 bench.py "$@" > "$out/$name.log" 2>&1; tail -1 "$out/$name.log" > "$out/$name.json"; python3 -c "
import json,sys; d=json.load(open('$out/$name.json')); print(round(d['value'],2), round(d['prefill_ms'],1), d.get('prefill_ms_defer_kv_0'), round(d['ms_per_step'],3), (d.get('prefill_detail') or {}).get('gemm_tflops'), d.get('value_cooperative'), d.get('value_cooperative_kv_in_hbm'), (d.get('cpu_baseline') or {}).get('value'))"; }
run bench_driver_flags --steps 20 --warmup 5
run opt30b_gpu10_p0p2_pack10
run opt30b_gpu100_resident --gpu-percentage 100 --no-raw-leg --no-cpu-baseline
run llama3_8b_gpu100_b128_t1024_n128 --model llama-3-8b --gpu-percentage 100 --batch 128 --prompt 1024 --steps 127
run opt30b_b900_t32_gpu0_p0p2_mb2 --batch 900 --prompt 32 --gpu-percentage 0 --num-minibatch 2 --steps 20 --no-raw-leg --no-cpu-baseline --no-cooperative-leg --no-defer-kv-leg --no-auto-plan
run dp2_same_gpu_gloo_opt30b_gb64 --dp-same-gpu --dp-backend gloo --gpus 2 --global-batch 64 --steps 8 --warmup 1 --dp-extra-timeout 900
run dp1_rccl_world1_opt30b --force-dp --steps 8 --warmup 1 --no-raw-leg --no-cpu-baseline --no-cooperative-leg
