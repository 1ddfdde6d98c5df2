#!/bin/bash
# LDS bank conflicts of every staging form (rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE, one pass each): the 256x256 kernel with
# B stored [N][K] and row-major (transposing reads), the 128x128 kernel with [N][K], row-major B, and A stored [K][M].
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{ GEMM_PROBE_LAYOUT=nt bash tools/dev/pmc_gemm.sh w4_nt 5 8192,8192,8192 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  GEMM_PROBE_LAYOUT=nn bash tools/dev/pmc_gemm.sh w4_nn 5 8192,8192,8192 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  GEMM_PROBE_LAYOUT=nt bash tools/dev/pmc_gemm.sh lp128_nt 3 2048,2048,2048 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  GEMM_PROBE_LAYOUT=nn bash tools/dev/pmc_gemm.sh lp128_nn 3 2048,2048,2048 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  GEMM_PROBE_LAYOUT=tn bash tools/dev/pmc_gemm.sh lp128_tn 3 2048,2048,2048 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  GEMM_PROBE_LAYOUT=nn bash tools/dev/pmc_gemm.sh lp128_nn_fewrows 3 16,8192,8192 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  GEMM_PROBE_LAYOUT=nt bash tools/dev/pmc_gemm.sh q_nt 7 2048,2048,2048 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; } > gpurun_out/r03_lds_bank_conflicts.txt 2>&1
cat gpurun_out/r03_lds_bank_conflicts.txt; tail -2 gpurun_out/pg_w4_nn_1.log
