#!/bin/bash
# r06 GPU call 22: tools/issue_model (how a SIMD shares its time between matrix, vector, scalar instructions and s_nop), and the
# attention experiments again with every shape's line
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
timeout 200 ./tools/issue_model > gpurun_out/r06/issue_model.txt 2>&1; cat gpurun_out/r06/issue_model.txt


