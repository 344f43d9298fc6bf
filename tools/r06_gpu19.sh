#!/bin/bash
# r06 GPU call 19: lia_attention.o built with -fno-honor-nans (see csrc/Makefile): bit-identity + time, then the whole GPU suite, smoke and
# the driver's bench line on the final tree
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
{ timeout 400 ./tools/attn_prefill_bench 5; timeout 200 python tools/attn_time.py; } > gpurun_out/r06/attn_bench5.txt 2>&1
echo "mismatches: $(grep -c MISMATCH gpurun_out/r06/attn_bench5.txt)"; head -8 gpurun_out/r06/attn_bench5.txt; tail -6 gpurun_out/r06/attn_bench5.txt
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06/test_gpu_final.txt 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r06/test_gpu_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06/smoke_final.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r06/smoke_final.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_final.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r06/bench_final.log > gpurun_out/r06/bench_final.json; tail -1 gpurun_out/r06/bench_final.log | cut -c1-600
