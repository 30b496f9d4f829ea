#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV passes (tools/pmc_profile.sh) per hipfeat kernel dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> [per-dispatch values]
dur = defaultdict(list)
for path in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "")
            if "hipfeat" not in k:
                continue
            vals[k.split("(")[0]][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
for path in sorted(glob.glob(os.path.join(out, "pass*", "**", "*kernel_trace.csv"), recursive=True)):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "")
            if "hipfeat" in k:
                dur[k.split("(")[0]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k, cs in vals.items():
    print(f"== {k}")
    if dur[k]:
        d = sorted(dur[k])
        print(f"   dispatch duration under profiling: median {d[len(d)//2]:.1f} us (n={len(d)})")
    for c, lst in sorted(cs.items()):
        # a counter may be reported once per dispatch (already summed over XCDs/SEs) or in several rows
        per = defaultdict(float)
        for did, v in lst:
            per[did] += v
        v = sorted(per.values())
        print(f"   {c:28s} per-dispatch median {v[len(v)//2]:.6g}  (n={len(v)})")
