#!/usr/bin/env python3
"""rocprofv3 counter_collection.csv of tools/probe_bench.py → per-variant averages of every counter for
the count-only probe kernel (dispatch order: clustered x6, unclustered x6, selective x6, selective unclustered x6;
the first dispatch of each six is the warm-up)."""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)  # counter -> [(dispatch id, value, ns)]
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        if not r["Kernel_Name"].startswith("k_join_probe_count"):
            continue
        rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
names = ["clustered", "unclustered", "selective", "selective_unclustered"]
for c, vals in rows.items():
    vals.sort()
    per = [vals[i:i + 6] for i in range(0, len(vals), 6)]
    out = []
    for k, grp in enumerate(per[:4]):
        g = grp[1:] or grp
        out.append("%s %.4g (%.2f ms)" % (names[k], sum(v for _, v, _ in g) / len(g), sum(t for _, _, t in g) / len(g) / 1e6))
    print(c.ljust(36), " | ".join(out))
