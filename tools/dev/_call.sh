python -m pytest tests/test_gpu_gemm.py tests/test_gpu_full_size.py tests/test_gpu_gemm_fuzz.py tests/test_gpu_select_audit.py -q -m gpu -k "select or auto or dispatch or chosen or as_benched or audit or tile or takes" 2>&1 | tail -8
for seed in 21 22 23; do timeout 600 python tools/dev/random_audit.py $seed 96; done > gpurun_out/random_audit_r05b.log 2>&1
grep -c AUTO gpurun_out/random_audit_r05b.log; grep BEHIND gpurun_out/random_audit_r05b.log
