set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for algo in 7 14 5 7 14; do timeout 120 python tools/c5_probe.py 6 nt 512 $algo; done
for algo in 7 14; do timeout 120 python tools/c5_probe.py 8 nt 64 $algo; done
} > gpurun_out/r06_c5_baseline.txt 2>&1
cat gpurun_out/r06_c5_baseline.txt
