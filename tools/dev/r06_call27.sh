set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for rep in 1 2 3; do
for v in "" _qmpad1 _qmpad2 _qmpad4; do
  so=cubecl_amd/csrc/libmi355cube.so; [ -n "$v" ] && so=cubecl_amd/csrc/variants/libmi355cube$v.so
  echo "== $so"; PROBE_M=8192 MI355CUBE_LIB=$PWD/$so timeout 120 python tools/c5_probe.py 10 nt 1 15
done; done
} > gpurun_out/r06_qm_pad_cost.txt 2>&1
cat gpurun_out/r06_qm_pad_cost.txt
