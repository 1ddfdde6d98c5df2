set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for rep in 1 2 3; do
  timeout 1500 python -m pytest tests -x -q -m gpu --no-header --timeout 300 -p no:cacheprovider > gpurun_out/r06_pytest_rep$rep.log 2>&1
  echo "rep $rep exit $?"; grep -E "passed|failed" gpurun_out/r06_pytest_rep$rep.log | tail -1; grep -E "^FAILED|^ERROR" gpurun_out/r06_pytest_rep$rep.log | head -5
done
for rep in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_gpu_select_audit.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -1; done
