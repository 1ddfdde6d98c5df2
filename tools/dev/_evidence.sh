bash tools/pmc_all.sh 1c8e9db > gpurun_out/r05_pmc.log 2>&1; tail -3 gpurun_out/r05_pmc.log | cut -c1-400
# the PMC entries must be in place for the bench line of the same call
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json; cp gpurun_out/pmc_mfma_util.json profiles/pmc_mfma_util.json
bash tools/gpu_check.sh r05 > gpurun_out/r05_gpu_check.log 2>&1; grep -E "exit|passed|failed" gpurun_out/r05_gpu_check.log | head -12
bash tools/gpu_rehearse_n2.sh > gpurun_out/r05_rehearse.log 2>&1; head -5 gpurun_out/r05_rehearse.log
