set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "qm or lp256qm or identity or auto" 2>&1 | tail -5
for rep in 1 2 3; do
for v in "" _qmold; do
  so=cubecl_amd/csrc/libmi355cube.so; [ -n "$v" ] && so=cubecl_amd/csrc/variants/libmi355cube$v.so
  echo "== $so"; PROBE_M=8192 MI355CUBE_LIB=$PWD/$so timeout 120 python tools/c5_probe.py 10 nt 1 15
  MI355CUBE_LIB=$PWD/$so timeout 120 python tools/c5_probe.py 6 nt 512 15
  MI355CUBE_LIB=$PWD/$so timeout 120 python tools/c5_probe.py 6 nn 512 15
done; done
} > gpurun_out/r06_qm_pre_ab.txt 2>&1
cat gpurun_out/r06_qm_pre_ab.txt
