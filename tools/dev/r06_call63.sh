set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for seed in 801 802 803 804 901 902 903 904; do timeout 1500 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_held_out_final2.txt 2>&1
grep -c "AUTO ->" gpurun_out/r06_random_audit_held_out_final2.txt; grep "BEHIND" gpurun_out/r06_random_audit_held_out_final2.txt
echo "== fresh"
for seed in 701 702 703 704 705 706 707 708; do timeout 1500 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_fresh_seeds_final2.txt 2>&1
grep -c "BEHIND" gpurun_out/r06_random_audit_fresh_seeds_final2.txt
echo "== tuned"
for seed in 601 602 603 604; do timeout 1500 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_final4.txt 2>&1
grep -c "BEHIND" gpurun_out/r06_random_audit_final4.txt
timeout 900 python tools/dev/llm_audit.py > gpurun_out/r06_llm_audit_after.txt 2>&1; tail -1 gpurun_out/r06_llm_audit_after.txt; grep BEHIND gpurun_out/r06_llm_audit_after.txt
