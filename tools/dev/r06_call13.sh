set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_reduce.py tests/test_gpu_layout_reduce_fuzz.py tests/test_gpu_full_size.py tests/test_gpu_runtime.py -q --no-header -p no:cacheprovider -x -k "reduce or sum or argm or c4 or c1 or exchange or all_reduce" --timeout 600 > gpurun_out/r06_reduce_pytest.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/r06_reduce_pytest.log | tail -2
{
for rep in 1 2; do
echo "== product"; timeout 300 python tools/dev/shard_probe.py
echo "== tickets (MI355_REDUCE_POLL=0)"; MI355_REDUCE_POLL=0 timeout 300 python tools/dev/shard_probe.py
done
echo "== trace"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_redtrace.so timeout 300 python tools/dev/shard_probe.py | grep -v "^sum \|^argmax\|^fused "
echo "== 1 GiB"; timeout 300 python tools/reduce_probe.py
} > gpurun_out/r06_shard_dpp.txt 2>&1
grep -v "entry per XCD\|by dispatch\|by quarter\|duration per XCD\|correlation" gpurun_out/r06_shard_dpp.txt
