set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
echo "== trace, polled"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_redtrace.so timeout 300 python tools/dev/shard_probe.py
echo "== trace, polled, 2 wg per cu"; MI355_REDUCE_WG_PER_CU=2 MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_redtrace.so timeout 300 python tools/dev/shard_probe.py
} > gpurun_out/r06_shard_trace2.txt 2>&1
grep -v "^sum \|^argmax\|^fused " gpurun_out/r06_shard_trace2.txt
