#!/usr/bin/env python3
"""Per-kernel means of a rocprofv3 --pmc run:  python3 tools/pmc_kernels.py <rocprof output dir> <out.json> <kernel substring> [...]
Reads every *counter_collection.csv (and *kernel_trace.csv for the durations) under the directory; one entry per kernel substring."""
import csv
import glob
import json
import os
import sys

d, out, names = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {}
for name in names:
    per, dur = {}, []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if name in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
                per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if name in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    e = {k: {"launches": len(v), "mean_per_launch": sum(v.values()) / len(v)} for k, v in per.items()}
    if dur:
        e["kernel_us_under_pmc"] = sum(dur) / len(dur)
    res[name] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
