set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for rep in 1 2 3; do
for so in cubecl_amd/csrc/libmi355cube.so cubecl_amd/csrc/variants/libmi355cube_s64p32.so; do
  echo "== $(basename $so)"; MI355CUBE_LIB=$PWD/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos stream64 64x8192x8192 8192x64x8192 48x8192x8192 32x8192x8192 16x8192x8192 | tail -n 5
done; done
echo "== long K"; timeout 300 python tools/ab_algos.py --rounds 5 --algos stream64,lp128 64x8192x16384 64x8192x32768 48x8192x16384 64x4096x16384 40x8192x12288
} > gpurun_out/r06_stream64_pitch_ab.txt 2>&1
cat gpurun_out/r06_stream64_pitch_ab.txt
