#!/usr/bin/env python3
"""rocprofv3 counter_collection.csv → {kernel: {counter: average per launch, "launches": n, "avg_ms": t}} for the kernels
whose names start with one of the given prefixes.  usage: pmc_counters.py <csv> <prefix> [<prefix> …]"""
import csv
import json
import sys
from collections import defaultdict

vals = defaultdict(lambda: defaultdict(list))
times = defaultdict(dict)
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"]
        p = next((p for p in sys.argv[2:] if k.startswith(p)), None)
        if p is None:
            continue
        vals[p][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "End_Timestamp" in r and r["End_Timestamp"]:
            times[p][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = {}
for p, cs in vals.items():
    out[p] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[p]["launches"] = max(len(v) for v in cs.values())
    if times[p]:
        out[p]["avg_ms"] = sum(times[p].values()) / len(times[p]) / 1e6
print(json.dumps(out, indent=1))
