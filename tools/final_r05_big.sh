cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/final_r05
mkdir -p "$out"
export LIA_STATE_DIR="$PWD/$out/state"
run() { name=$1; shift; echo "== $name: bench.py $*"; timeout 1700 python3 bench.py "$@" > "$out/$name.log" 2>&1; tail -1 "$out/$name.log" > "$out/$name.json"; python3 -c "
import json,sys; d=json.load(open('$out/$name.json')); print(round(d['value'],2), round(d['prefill_ms'],1), round(d['ms_per_step'],3), d.get('value_cooperative'), d.get('value_cooperative_kv_in_hbm'))"; }
run opt66b_gpu5_cxl_pack10 --model opt-66b --gpu-percentage 5 --enable-cxl --cxl-nodes 0,1 --batch 32 --no-raw-leg --no-cpu-baseline
run opt175b_gpu5_cxl_pack10_uniform01 --model opt-175b --gpu-percentage 5 --enable-cxl --cxl-nodes 0,1 --batch 32 --init uniform01 --no-raw-leg --no-cpu-baseline
