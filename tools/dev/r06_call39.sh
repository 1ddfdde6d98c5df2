set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_qmbuf.so timeout 300 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "lp256qm" 2>&1 | tail -3
for rep in 1 2 3; do
for v in "" _qmabl1 _qmbuf; do
  so=cubecl_amd/csrc/libmi355cube.so; [ -n "$v" ] && so=cubecl_amd/csrc/variants/libmi355cube$v.so
  echo "== $so"
  MI355CUBE_LIB=$PWD/$so timeout 120 python tools/c5_probe.py 6 nt 512 15
done; done
} > gpurun_out/r06_qm_buffer_store.txt 2>&1
cat gpurun_out/r06_qm_buffer_store.txt
