set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for k in 2048 3072 4096; do
  shapes=""
  for m in 24 31 47 64; do for n in 8192 8704 9216 9528 10240 11264 12288 14336 16568; do shapes="$shapes ${m}x${n}x${k}"; done; done
  timeout 900 python tools/ab_algos.py --rounds 5 --algos auto,stream64,lp128 $shapes
done
} > gpurun_out/r06_stream64_second_round_ab.txt 2>&1
cat gpurun_out/r06_stream64_second_round_ab.txt
