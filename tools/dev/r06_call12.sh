set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_reduce.py tests/test_gpu_layout_reduce_fuzz.py tests/test_gpu_full_size.py -q --no-header -p no:cacheprovider -x -k "reduce or sum or argm or c4 or c1" --timeout 600 > gpurun_out/r06_reduce_pytest.log 2>&1
echo "pytest exit $?"; tail -n 3 gpurun_out/r06_reduce_pytest.log
{
for pm in 0 1000 915 880 940 915 0; do echo "== MI355_REDUCE_ODD_PERMILLE=$pm"; MI355_REDUCE_ODD_PERMILLE=$pm timeout 300 python tools/dev/shard_probe.py; done
echo "== trace 915"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_redtrace.so timeout 300 python tools/dev/shard_probe.py | grep -v "^sum \|^argmax\|^fused "
for pm in 0 1000 915 880; do echo "== 1 GiB MI355_REDUCE_ODD_PERMILLE=$pm"; MI355_REDUCE_ODD_PERMILLE=$pm timeout 300 python tools/reduce_probe.py; done
} > gpurun_out/r06_shard_weighted.txt 2>&1
cat gpurun_out/r06_shard_weighted.txt
