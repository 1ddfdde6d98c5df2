set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for v in qmtrace qmtrace1 qmtrace3; do
echo "== $v"
MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_$v.so timeout 200 python tools/dev/qm_trace.py 512 2048
done
MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_qmtrace.so timeout 200 python tools/dev/qm_trace.py 1 8192
MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_qmstag.so timeout 300 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "lp256qm" 2>&1 | tail -3
for rep in 1 2 3; do
for v in "" _qmstag; do
  so=cubecl_amd/csrc/libmi355cube.so; [ -n "$v" ] && so=cubecl_amd/csrc/variants/libmi355cube$v.so
  echo "== $so"
  MI355CUBE_LIB=$PWD/$so timeout 120 python tools/c5_probe.py 6 nt 512 15
  MI355CUBE_LIB=$PWD/$so timeout 120 python tools/c5_probe.py 6 nn 512 15
done; done
} > gpurun_out/r06_qm_stagger.txt 2>&1
cat gpurun_out/r06_qm_stagger.txt
