cd "${GRAFT_REPO_ROOT:-.}"
for skv in 0 1 2 3 4; do echo "== LLAMA M=128 SKV=$skv"; SKV=$skv LLAMA=1 tools/gemm_bench 128 | grep -v dummy; done
for sp in 1 2 4; do echo "== LLAMA M=128 SKV=3 split=$sp"; SKV=3 LLAMA=1 tools/gemm_bench 128 $sp | grep -v dummy; done
for skv in 0 1 2 3; do echo "== OPT M=128 SKV=$skv"; SKV=$skv tools/gemm_bench 128 | grep -v dummy; done
