#!/bin/bash
# usage (GPU box): tools/dev/cycles.sh TAG "<gemm_probe args>"  -- GRBM_GUI_ACTIVE (cycles) + duration per GEMM dispatch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; ARGS=$2
export TMPDIR=/tmp; R=$PWD
( cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/cyc_$TAG -o p -- python $R/tools/dev/gemm_probe.py $ARGS > $R/gpurun_out/cyc_$TAG.log 2>&1 )
python - "$TAG" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
cc = list(csv.DictReader(open(glob.glob(f"gpurun_out/cyc_{tag}/**/*counter_collection.csv", recursive=True)[0])))
agg = collections.OrderedDict()
for r in cc:
    if "gemm" not in r["Kernel_Name"]: continue
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg.setdefault((r["Kernel_Name"][28:48], r["Grid_Size"]), []).append((float(r["Counter_Value"]) / 8, dur))
for k, v in agg.items():
    v = v[3:]
    cyc = sorted(x[0] for x in v); dur = sorted(x[1] for x in v)
    print(f"{tag:10s} {k[0]} grid {k[1]:>8s} n={len(v):3d} cycles med {cyc[len(cyc)//2]/1e3:9.1f}K min {cyc[0]/1e3:9.1f}K   dur med {dur[len(dur)//2]/1e3:8.1f} us  clock {cyc[len(cyc)//2]/dur[len(dur)//2]:.3f} GHz")
PY
