set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for rep in 1 2 3; do
  timeout 1200 python -m pytest tests -x -q -m gpu --no-header -p no:cacheprovider 2>&1 | grep -a " passed\| failed\|FAILED\|Error" | head -5
done > gpurun_out/r06_three_suites.txt 2>&1
cat gpurun_out/r06_three_suites.txt
