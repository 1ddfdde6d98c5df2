set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm.py -q --no-header -p no:cacheprovider -x -k "lp256qm" --timeout 300 > gpurun_out/r06_qm_pytest.log 2>&1
echo "pytest exit $?"; tail -n 5 gpurun_out/r06_qm_pytest.log
{
for rep in 1 2 3; do
for v in "" _qmbd0; do
  so=cubecl_amd/csrc/libmi355cube.so; [ -n "$v" ] && so=cubecl_amd/csrc/variants/libmi355cube$v.so
  echo "== $so"; MI355CUBE_LIB=$PWD/$so timeout 120 python tools/c5_probe.py 6 nt 512 15
done
timeout 120 python tools/c5_probe.py 6 nt 512 7
done
} > gpurun_out/r06_qm_bdrip.txt 2>&1
cat gpurun_out/r06_qm_bdrip.txt
