set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm.py -q --no-header -p no:cacheprovider -x -k "lp256qm" --timeout 300 > gpurun_out/r06_qm_pytest.log 2>&1
echo "pytest exit $?"; tail -n 15 gpurun_out/r06_qm_pytest.log
{
for algo in 15 7 15 14 15 7; do timeout 120 python tools/c5_probe.py 6 nt 512 $algo; done
for algo in 15 7 15; do timeout 120 python tools/c5_probe.py 8 nt 64 $algo; done
} > gpurun_out/r06_c5_qm.txt 2>&1
cat gpurun_out/r06_c5_qm.txt
