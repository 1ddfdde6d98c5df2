set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/gpu_check.sh r06 > gpurun_out/r06_gpu_check_final.log 2>&1
tail -n 40 gpurun_out/r06_gpu_check_final.log
