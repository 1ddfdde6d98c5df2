#!/bin/bash
# Runs on the GPU box: HBM traffic of the 1 GiB reductions from rocprofv3 PMC counters (FETCH_SIZE x2 on gfx950).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; R=$PWD
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_reduce -o p -- python $R/tools/dev/reduce_probe.py > $R/gpurun_out/pmc_reduce.log 2>&1 )
python - <<'PY'
import csv, glob, json, statistics, collections
f = glob.glob("gpurun_out/pmc_reduce/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "reduce_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in agg.items():
    name = "sum" if "<true, false>" in k else "argmax" if "<false, true>" in k else "sum_argmax"
    out[f"reduce_1GiB_{name}"] = {"kernel": k[:90], "fetch_bytes": int(statistics.median(v) * 1024 * 2), "algorithmic_bytes": 1 << 30,
                                   "FETCH_SIZE_KiB_raw": statistics.median(v)}
print(json.dumps(out, indent=1)); json.dump(out, open("gpurun_out/pmc_reduce.json", "w"), indent=1)
PY
