set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --no-header --timeout 300 -p no:cacheprovider --maxfail=60 > gpurun_out/r06_pytest.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/r06_pytest.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/r06_pytest.log | head -20
for seed in 601 602 603 604; do timeout 1500 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_final3.txt 2>&1
grep -c "AUTO ->" gpurun_out/r06_random_audit_final3.txt; grep "BEHIND" gpurun_out/r06_random_audit_final3.txt
echo "== fresh 701-708"
for seed in 701 702 703 704 705 706 707 708; do timeout 1500 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_fresh_seeds_final.txt 2>&1
grep -c "AUTO ->" gpurun_out/r06_random_audit_fresh_seeds_final.txt; grep -c "BEHIND" gpurun_out/r06_random_audit_fresh_seeds_final.txt
echo "== held out 801-804, 901-904"
for seed in 801 802 803 804 901 902 903 904; do timeout 1500 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_held_out_final.txt 2>&1
grep -c "AUTO ->" gpurun_out/r06_random_audit_held_out_final.txt; grep "BEHIND" gpurun_out/r06_random_audit_held_out_final.txt
python tools/kres.py cubecl_amd/csrc/*.hip > gpurun_out/r06_kres.txt 2>&1
bash tools/gpu_rehearse_n2.sh 2>&1 | tail -6
