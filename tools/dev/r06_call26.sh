set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for v in qmtrace qmtrace3; do
echo "== $v"
MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_$v.so timeout 200 python tools/dev/qm_trace.py 512 2048
MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_$v.so timeout 200 python tools/dev/qm_trace.py 1 8192
done
} > gpurun_out/r06_qm_handover_wait.txt 2>&1
cat gpurun_out/r06_qm_handover_wait.txt
