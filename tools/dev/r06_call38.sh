set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "stream64_f32" 2>&1 | tail -4
for p in 0 1 0 1; do echo "== MI355_S64F_PIPE=$p"; MI355_S64F_PIPE=$p timeout 300 python tools/ab_algos.py --f32 --rounds 5 --algos stream64,f32 64x8192x8192 8192x64x8192 64x14336x4096 56x16384x2048 64x16384x8192; done
} > gpurun_out/r06_stream64_f32_pipelined.txt 2>&1
cat gpurun_out/r06_stream64_f32_pipelined.txt
