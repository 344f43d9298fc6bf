cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/final_r05
mkdir -p "$out"
export LIA_STATE_DIR="$PWD/$out/state"
run() { name=$1; shift; echo "== $name: bench.py $*"; timeout 1200 python3 bench.py "$@" > "$out/$name.log" 2>&1; tail -1 "$out/$name.log" > "$out/$name.json"; python3 -c "
import json,sys; d=json.load(open('$out/$name.json')); print(round(d['value'],2), round(d['prefill_ms'],1), d.get('prefill_ms_defer_kv_0'), round(d['ms_per_step'],3), (d.get('prefill_detail') or {}).get('gemm_tflops'))"; }
run bench_driver_flags --steps 20 --warmup 5
run opt30b_gpu10_p0p2_pack10
run opt30b_gpu10_p3p3_pack10 --prefill-policy 3 --decoding-policy 3 --no-raw-leg --no-cpu-baseline
run opt30b_gpu100_resident --gpu-percentage 100 --no-raw-leg --no-cpu-baseline
run llama3_8b_gpu100_b128_t1024_n128 --model llama-3-8b --gpu-percentage 100 --batch 128 --prompt 1024 --steps 127
run opt30b_gpu10_p0p2_mb2_pack10 --num-minibatch 2 --no-raw-leg --no-cpu-baseline
