set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_qmtrace.so timeout 200 python tools/dev/qm_trace.py 512
for rep in 1 2; do
for v in "" _qmabl1 _qmabl2 _qmabl3 _qmlag2 _qmsf0; do
  so=cubecl_amd/csrc/libmi355cube.so; [ -n "$v" ] && so=cubecl_amd/csrc/variants/libmi355cube$v.so
  echo "== $so"; MI355CUBE_LIB=$PWD/$so timeout 120 python tools/c5_probe.py 6 nt 512 15
done; done
} > gpurun_out/r06_qm_trace_abl.txt 2>&1
cat gpurun_out/r06_qm_trace_abl.txt
