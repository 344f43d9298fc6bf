#!/bin/bash
# r06 GPU call 14: FETCH_SIZE of the prefill GEMM (fc1, M = 16384) under four tile orders (T2_GM = 2 / 4 / 8 / 16 row tiles per XCD group):
# the measurement behind "the 10.8 GB are the floor of a 4 MB L2, not an accident of the order" (LABNOTES r06)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/r06/pmc_gm
mkdir -p "$out"
: > gpurun_out/r06/gemm_tile_order_fetch.txt
for g in 2 4 8 16; do
  bin=./tools/gemm_bench_gm$g; [ "$g" = 4 ] && bin=./tools/gemm_bench
  echo "== T2_GM = $g: time" >> gpurun_out/r06/gemm_tile_order_fetch.txt
  SHAPE=28672,7168 timeout 120 $bin 16384 2>&1 | grep custom >> gpurun_out/r06/gemm_tile_order_fetch.txt
  SHAPE=28672,7168 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/gm$g" -o pmc -- $bin 16384 > "$out/gm$g.log" 2>&1
  f=$(find "$out/gm$g" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'P' >> gpurun_out/r06/gemm_tile_order_fetch.txt
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "tiled256p" not in r["Kernel_Name"] or r["Counter_Name"] != "FETCH_SIZE": continue
    grid = r.get("Grid_Size", "")
    acc[grid] += float(r["Counter_Value"]); n[grid] += 1
for g, v in acc.items():
    print(f"   lia_gemm_tiled256p_kernel grid {g}: FETCH_SIZE {v / n[g] / 1e6:.3f} M KiB per launch -> x 2 (gfx950) = {2 * v / n[g] * 1024 / 1e9:.2f} GB  ({n[g]} launches)")
P
done
cat gpurun_out/r06/gemm_tile_order_fetch.txt
rm -rf "$out"
