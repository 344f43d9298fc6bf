#!/bin/bash
# r06 GPU call 6a: attention XCD-order A/B (bit-identity + time), the planner's repaired picks, the batch-1150 line under the 0.92 cache check
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
{ echo "== XCD-aware grid (product)"; QUICK=1 timeout 300 ./tools/attn_prefill_bench; echo "== r05's grid (-DLIA_ATTN_LEGACY_GRID)"; QUICK=1 timeout 300 ./tools/attn_prefill_bench_legacy_grid; echo "== full sweep, product"; timeout 300 ./tools/attn_prefill_bench 5; timeout 200 python tools/attn_time.py; timeout 100 ./tools/attn_occ_test; } > gpurun_out/r06/attn_bench2.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_llama.py -q -m gpu -k "attention or prefill or llama" > gpurun_out/r06/test_attn.txt 2>&1; echo "attn tests rc=$?" > gpurun_out/r06/summary6a.txt
timeout 900 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --only 'cxl_opt30b_32_128_b1150' --timeout 800 > gpurun_out/r06/matrix6a.txt 2>&1
timeout 2400 python tools/run_matrix.py --outdir gpurun_out/r06/matrix --auto-plan-only --only 'offline_opt30b_32_256_b900|cxl_opt30b_32_128_b1150|cxl_opt30b_32_256_b1050' --timeout 750 >> gpurun_out/r06/matrix6a.txt 2>&1
cat gpurun_out/r06/attn_bench2.txt | head -30; tail -4 gpurun_out/r06/test_attn.txt; cat gpurun_out/r06/summary6a.txt gpurun_out/r06/matrix6a.txt
