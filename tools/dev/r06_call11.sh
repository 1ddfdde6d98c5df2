set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for rot in 0 1 3 9 1 0; do echo "== MI355_REDUCE_ROT=$rot"; MI355_REDUCE_ROT=$rot timeout 300 python tools/dev/shard_probe.py; done
echo "== trace rot 1"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_redtrace.so timeout 300 python tools/dev/shard_probe.py | grep -v "^sum \|^argmax\|^fused "
echo "== 1 GiB rot 1"; timeout 300 python tools/reduce_probe.py
echo "== 1 GiB rot 0"; MI355_REDUCE_ROT=0 timeout 300 python tools/reduce_probe.py
} > gpurun_out/r06_shard_rot.txt 2>&1
cat gpurun_out/r06_shard_rot.txt
