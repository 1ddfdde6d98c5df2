set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python tools/ab_algos.py --rounds 5 --algos stream64,lp128 64x8192x10240 64x8192x12288 64x8192x14336 64x8192x16384 64x8192x24576 64x8192x32768 48x8192x12288 48x8192x14336 48x8192x24576 36x8192x16384 64x7168x16384 64x9216x16384 8192x64x14336 8192x64x16384 8192x48x16384 64x8192x65536 > gpurun_out/r06_stream64_longk_ab.txt 2>&1
cat gpurun_out/r06_stream64_longk_ab.txt
