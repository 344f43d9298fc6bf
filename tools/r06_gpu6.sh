#!/bin/bash
# r06 sixth GPU call: the round's rocprofv3 evidence (headline, batch 900, Llama decode timeline) and the final bench.py lines
cd "$(dirname "$0")/.." || exit 1
bash tools/profile_round.sh r06_opt30b > gpurun_out/r06_prof_opt30b.txt 2>&1
bash tools/profile_round.sh r06_opt30b_b900 --batch 900 --prompt 32 --gpu-percentage 0 --num-minibatch 2 > gpurun_out/r06_prof_b900.txt 2>&1
bash tools/profile_decode.sh r06_llama3_8b --model llama-3-8b --gpu-percentage 100 --batch 128 --prompt 1024 > gpurun_out/r06_prof_llama.txt 2>&1
bash tools/profile_decode.sh r06_opt30b --steps 8 > gpurun_out/r06_prof_opt30b_timeline.txt 2>&1
bash tools/final_r06.sh > gpurun_out/r06_final.txt 2>&1
tail -n 5 gpurun_out/r06_prof_opt30b.txt gpurun_out/r06_prof_b900.txt gpurun_out/r06_prof_llama.txt; cat gpurun_out/r06_final.txt
