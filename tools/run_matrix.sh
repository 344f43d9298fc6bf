#!/bin/bash
# The reference's benchmark matrix (scripts/run_performance.sh -> lia_offline.sh / lia_online.sh, scripts/cxl_offloading.sh, and the
# README quick example) through this build: one `python run.py <the reference's flags>` per line, results under results/r06_matrix_*.
# Arguments are passed on to tools/run_matrix.py (--list, --only REGEX, --auto-plan-too, --timeout S).
cd "$(dirname "$0")/.." && exec python3 tools/run_matrix.py "$@"
