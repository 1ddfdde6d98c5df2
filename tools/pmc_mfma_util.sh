#!/bin/bash
# Runs on the GPU box: matrix-pipe utilisation of the headline GEMM kernel from rocprofv3 PMC counters (BASELINE.json
# config C3: "rocprof MFMA-util reported").  One pass: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over every SIMD:
# = 32 x the number of 32x32x16 bf16 MFMAs, MI355X_MICROARCH.md "s_memtime tick vs SQ PMC units") and GRBM_GUI_ACTIVE
# (shader-clock cycles the launch was resident, summed over the 8 XCDs).  ROCm 7.2 has no gfx950 formula for the derived
# `MfmaUtil` metric (same guide, "rocprofv3 PMC slots"), so it is formed here:
#     mfma_util = (MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs)
# Writes gpurun_out/pmc_mfma_util.json (copy to profiles/pmc_mfma_util.json).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
SIZE=${1:-8192}
export TMPDIR=/tmp; R=$PWD
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma_util -o p -- \
    python $R/bench.py --no-extras --no-cpu-baseline --size $SIZE --steps 10 --warmup 3 > $R/gpurun_out/pmc_mfma_util.log 2>&1 )
python - "$SIZE" <<'PY'
import csv, glob, json, statistics, sys
size = int(sys.argv[1])
f = glob.glob("gpurun_out/pmc_mfma_util/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
med = lambda c: statistics.median(float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == c)
busy, active = med("SQ_VALU_MFMA_BUSY_CYCLES"), med("GRBM_GUI_ACTIVE")
n_mfma = size ** 3 / (32 * 32 * 16)
out = {f"gemm_bf16_{size}": {"kernel": rows[0]["Kernel_Name"], "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": active,
       "mfma_busy_cycles_per_simd": busy / 1024, "resident_cycles_per_xcd": active / 8,
       "mfma_util": round((busy / 1024) / (active / 8), 4), "expected_busy_cycles_32_per_mfma": 32 * n_mfma,
       "launches": len(rows) // 2,
       "note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace, medians over the launches of one bench.py --no-extras run"}}
json.dump(out, open("gpurun_out/pmc_mfma_util.json", "w"), indent=1)
print(json.dumps(out))
PY
