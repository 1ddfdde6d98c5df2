cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_check.sh r06a 2>&1 | tail -60
bash tools/pmc_all.sh c511e3e 2>&1 | tail -30
