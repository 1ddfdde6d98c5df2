#!/bin/bash
# Runs on the GPU box: HBM traffic of the headline GEMM kernel from rocprofv3 PMC counters, one pass per
# counter (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Correction for gfx950: FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read
# -> doubled; WRITE_SIZE is taken as reported (uncalibrated, noted in the output).  Units: KiB.
# Writes gpurun_out/pmc_traffic.json (copy to profiles/pmc_traffic.json).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
SIZE=${1:-8192}; ALGO=${2:-0}
export TMPDIR=/tmp; R=$PWD
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmc_traffic_$ctr -o p -- \
      python $R/bench.py --no-extras --no-cpu-baseline --size $SIZE --algo $ALGO --steps 10 --warmup 3 > $R/gpurun_out/pmc_traffic_$ctr.log 2>&1 )
done
python - "$SIZE" <<'PY'
import csv, glob, json, sys, statistics
size = int(sys.argv[1])
vals, name = {}, None
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_traffic_{ctr}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    vals[ctr] = statistics.median(float(r["Counter_Value"]) for r in rows)
    name = rows[0]["Kernel_Name"]
algo = {"lp256w4": 5, "lp256_": 4, "lp128": 3}
a = next(v for k, v in algo.items() if k in name)
fetch = vals["FETCH_SIZE"] * 1024 * 2      # gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x
write = vals["WRITE_SIZE"] * 1024
alg = 3 * size * size * 2
out = {f"gemm_bf16_{size}_algo{a}": {"kernel": name, "hbm_bytes_per_launch": int(fetch + write), "fetch_bytes": int(fetch),
       "write_bytes": int(write), "algorithmic_bytes": alg, "FETCH_SIZE_KiB_raw": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": vals["WRITE_SIZE"],
       "note": "rocprofv3 --pmc, one counter per pass; FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated"}}
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
