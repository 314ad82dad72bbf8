#!/usr/bin/env python3
"""Cut a rocprofv3 kernel trace of tools/query_timeline.py into per-query timelines.

For every query (median run by span): span = first kernel start → next marker start, busy = sum of
kernel durations inside it, idle = span − busy, launches, and the time per kernel name."""
import argparse
import csv
import json
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--out", required=True)
    ap.add_argument("--top", type=int, default=12)
    a = ap.parse_args()
    rows = []
    with open(a.trace, newline="") as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0]
            if name.startswith("void "):
                name = name[5:]
            grid = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
            wg = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 1)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, grid // max(wg, 1)))
    rows.sort()
    marks = [(i, r[3]) for i, r in enumerate(rows) if r[2] == "k_ldb_marker"]
    runs = defaultdict(list)
    for (i, mid), (j, _) in zip(marks, marks[1:]):
        if mid == 9999 or j <= i + 1:
            continue
        seg = rows[i + 1:j]
        span = rows[j][0] - rows[i][1]
        busy = sum(e - s for s, e, _, _ in seg)
        per = defaultdict(lambda: [0, 0])
        for s, e, n, _ in seg:
            per[n][0] += 1
            per[n][1] += e - s
        runs[mid // 100].append({"span_ns": span, "busy_ns": busy, "launches": len(seg), "per": per})
    out = {"unit": "ms", "queries": {}}
    tot_span = tot_busy = 0.0
    for q in sorted(runs):
        rs = sorted(runs[q], key=lambda r: r["span_ns"])
        m = rs[len(rs) // 2]
        top = sorted(m["per"].items(), key=lambda kv: -kv[1][1])
        out["queries"]["Q%d" % q] = {
            "span": round(m["span_ns"] / 1e6, 4), "busy": round(m["busy_ns"] / 1e6, 4), "idle": round((m["span_ns"] - m["busy_ns"]) / 1e6, 4),
            "busy_share": round(m["busy_ns"] / max(m["span_ns"], 1), 4), "launches": m["launches"], "runs": len(rs),
            "kernels": {n: {"launches": c, "ms": round(t / 1e6, 4)} for n, (c, t) in top[:a.top]},
            "rest_ms": round(sum(t for _, (c, t) in top[a.top:]) / 1e6, 4)}
        tot_span += m["span_ns"]
        tot_busy += m["busy_ns"]
    out["total"] = {"span": round(tot_span / 1e6, 3), "busy": round(tot_busy / 1e6, 3), "busy_share": round(tot_busy / max(tot_span, 1), 4)}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["total"]))
    for q, v in out["queries"].items():
        print(q, v["span"], v["busy"], v["idle"], v["launches"])


if __name__ == "__main__":
    main()
