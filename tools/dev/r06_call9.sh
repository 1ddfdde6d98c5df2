set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_reduce.py tests/test_gpu_layout_reduce_fuzz.py tests/test_gpu_full_size.py -q --no-header -p no:cacheprovider -x -k "reduce or sum or argm or c4 or c1" --timeout 600 > gpurun_out/r06_reduce_pytest.log 2>&1
echo "pytest exit $?"; tail -n 4 gpurun_out/r06_reduce_pytest.log
{
echo "== polled records (default)"; timeout 300 python tools/dev/shard_probe.py
echo "== tickets (MI355_REDUCE_POLL=0)"; MI355_REDUCE_POLL=0 timeout 300 python tools/dev/shard_probe.py
echo "== polled records (default), again"; timeout 300 python tools/dev/shard_probe.py
echo "== trace, polled"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_redtrace.so timeout 300 python tools/dev/shard_probe.py
echo "== trace, tickets"; MI355_REDUCE_POLL=0 MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_redtrace.so timeout 300 python tools/dev/shard_probe.py
echo "== 1 GiB"; timeout 300 python tools/reduce_probe.py
} > gpurun_out/r06_shard_probe.txt 2>&1
cat gpurun_out/r06_shard_probe.txt
