set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
echo "== up to 16 rows on very large streaming grids (the LM head): [N][K]"
S=""; for k in 2048 4096 8192; do for m in 4 8 16; do for n in 65536 100000 128256 152064; do S="$S ${m}x${n}x${k}"; done; done; done
timeout 1500 python tools/ab_algos.py --rounds 5 --algos auto,lp128,stream64,skinny $S
echo "== 17-32 rows past 768 streaming workgroups"
S=""; for k in 2048 4096 8192; do for m in 20 24 32; do for n in 28672 40960 57344 80000; do S="$S ${m}x${n}x${k}"; done; done; done
timeout 1500 python tools/ab_algos.py --rounds 5 --algos auto,lp128,stream64,lp192x192 $S
} > gpurun_out/r06_stream_large_grids_ab.txt 2>&1
cat gpurun_out/r06_stream_large_grids_ab.txt
