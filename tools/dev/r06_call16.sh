cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/dev/pmc_passes.sh s64 gemm_stream64 -- python $PWD/tools/ab_algos.py --rounds 2 --algos stream64 64x8192x8192 32x8192x8192 16x8192x8192 2>&1 | tail -120
