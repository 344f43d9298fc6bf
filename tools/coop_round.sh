#!/bin/bash
# tools/coop_round.sh <tag>: the online cooperative-split controller (--cpu-layers -1) beside a scan of fixed counts on the SAME
# box, the online run first and last so box drift shows; one bench.py JSON line each under gpurun_out/coop_<tag>/.
set -u
tag=${1:-r03}
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/coop_$tag
mkdir -p "$out"
run() { name=$1; shift; echo "== $name: bench.py $*"; timeout 900 python3 bench.py "$@" > "$out/$name.log" 2>&1; tail -1 "$out/$name.log" > "$out/$name.json"; cut -c1-160 "$out/$name.json"; echo; }
X="--no-raw-leg --no-cpu-baseline --no-cooperative-leg"
run p0p2_online_a --cpu-layers -1 $X
for c in 17 19 21 23; do run p0p2_cpu$c --cpu-layers $c --steps 12 $X; done
run p0p2_online_b --cpu-layers -1 $X
P="--prefill-policy 3 --decoding-policy 3"
run p3p3_online_a $P --cpu-layers -1 $X
for c in 19 21 23 25; do run p3p3_cpu$c $P --cpu-layers $c --steps 12 $X; done
run p3p3_online_b $P --cpu-layers -1 $X
python3 - "$out" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read())
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
        continue
    c = d.get("cooperative_controller") or {}
    print("%-18s %7.1f tok/s %7.1f ms/step  %s" % (os.path.basename(f)[:-5], d["value"], d["ms_per_step"],
          {k: c[k] for k in ("host_layers", "centre", "converged", "moves", "searches", "ms_by_count") if k in c}))
PY
