set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --no-header --timeout 600 -p no:cacheprovider --maxfail=30 -x -k "gemm or full_size or select" > gpurun_out/r06_gemm_pytest.log 2>&1
echo "pytest exit $?"; tail -n 25 gpurun_out/r06_gemm_pytest.log
timeout 300 python tools/ab_algos.py --rounds 3 --algos lp256w4,lp256m16,lp256qm 6144x4096x8192 6144x6144x2048 > gpurun_out/r06_qm_misc_ab.txt 2>&1; cat gpurun_out/r06_qm_misc_ab.txt
