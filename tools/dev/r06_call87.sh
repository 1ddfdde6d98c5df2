set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/pmc_all.sh 396fc62 > gpurun_out/r06_pmc_final.log 2>&1
tail -n 14 gpurun_out/r06_pmc_final.log | cut -c1-200
