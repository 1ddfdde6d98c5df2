set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
rocm-smi --showclocks 2>/dev/null | grep -i "mclk\|sclk" | head -4
for rep in 1 2; do
timeout 900 python tools/dev/batched_audit.py 8x2048x2048x128 8x4096x4096x128 32x1024x1024x128 32x2048x2048x128 32x4096x4096x128 64x1024x1024x128 64x2048x2048x128 64x4096x4096x128 1x8192x8192x128 1x16384x8192x128 1x8192x8192x192 1x16384x16384x128 16x2048x2048x64 16x4096x4096x64 | grep -v "^==" 
done
} > gpurun_out/r06_batched_short_k_ab.txt 2>&1
cut -c1-150 gpurun_out/r06_batched_short_k_ab.txt
