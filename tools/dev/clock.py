#!/usr/bin/env python3
"""Dev: effective shader clock per kernel = GRBM_GUI_ACTIVE / duration, from a rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace run dir."""
import csv, sys, glob, collections
d = sys.argv[1]
cc = list(csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])))
agg = collections.defaultdict(list)
for r in cc:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) if "End_Timestamp" in r else None
    agg[r["Kernel_Name"][:70]].append((float(r["Counter_Value"]), dur))
for k, v in agg.items():
    v = v[len(v)//2:]
    cyc = sum(x[0] for x in v) / len(v)
    if v[0][1]:
        dur = sum(x[1] for x in v) / len(v)
        print(f"{k:70s} n={len(v):3d} GUI_ACTIVE {cyc:12.0f}  dur {dur/1e3:9.1f} us  clock {cyc/dur:6.3f} GHz")
    else:
        print(f"{k:70s} n={len(v):3d} GUI_ACTIVE {cyc:12.0f}")
