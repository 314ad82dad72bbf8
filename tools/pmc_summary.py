#!/usr/bin/env python3
"""Per-kernel HBM traffic from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected in SEPARATE
runs, as MI355X_MICROARCH.md prescribes), calibrated on kernels whose byte count is known.

  python tools/pmc_summary.py --fetch <..._counter_collection.csv> [--write <...csv>]
         --calib-kernel k_scan_count_spec --calib-bytes 9600000000,2400000000 --out profiles/r01_pmc_q1_sf100.json

rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE under-reports streaming
reads (½ for 16 B/lane loads; other widths "uncalibrated"), so the factor is measured, not
assumed: calib = known bytes of a calibration launch / its raw FETCH_SIZE.  The calibration
launches are bench.py's hbm_ceiling() scans — the library's count-only scan over one dense column
(every cache line of the column is fetched exactly once): a 16-byte decimal column read with
8-byte loads (bytes = rows x 16) and a 4-byte date column read with dword loads (rows x 4).  The
launches of the calibration kernel are clustered by raw value and matched to --calib-bytes in
descending order.  `fetch_bytes` of every other kernel uses the FIRST factor (the 8-byte-load one,
the dominant access width of the TPC-H decimal columns); both factors are recorded."""
import argparse
import csv
import json
from collections import defaultdict


def load(path, counter):
    per = defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0]
            per[name].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return per


def clusters(vals):
    """group launch values that agree within 5 %; returns the cluster means, descending"""
    out = []
    for v in sorted(vals, reverse=True):
        if out and abs(out[-1][0] / out[-1][1] - v) <= 0.05 * v:
            out[-1][0] += v
            out[-1][1] += 1
        else:
            out.append([v, 1])
    return [s / n for s, n in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write")
    ap.add_argument("--calib-kernel", default="k_scan_count_spec")
    ap.add_argument("--calib-bytes", required=True, help="known byte counts of the calibration launches, descending, comma separated")
    ap.add_argument("--min-kb", type=float, default=1e5, help="ignore kernels below this FETCH_SIZE (KiB) per launch")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    fetch = load(a.fetch, "FETCH_SIZE")
    write = load(a.write, "WRITE_SIZE") if a.write else {}
    cal = fetch.get(a.calib_kernel)
    if not cal:
        raise SystemExit(f"calibration kernel {a.calib_kernel} not in {a.fetch}")
    known = [float(x) for x in a.calib_bytes.split(",")]
    raw = clusters([v for v, _ in cal])
    if len(raw) < len(known):
        raise SystemExit(f"{len(known)} calibration sizes but {len(raw)} distinct launch groups of {a.calib_kernel}")
    points = [{"known_bytes": k, "raw_fetch_kib": r, "factor_bytes_per_kib": round(k / r, 2), "x_over_1024": round(k / r / 1024.0, 4)} for k, r in zip(known, raw)]
    factor = known[0] / raw[0]
    out = {"counter_unit": "KiB", "calibration": {"kernel": a.calib_kernel, "points": points}, "kernels": {}}
    for name, vals in sorted(fetch.items()):
        r = sum(v for v, _ in vals) / len(vals)
        if r < a.min_kb:
            continue
        e = {"launches": len(vals), "fetch_raw_kib": r, "fetch_bytes": r * factor, "avg_ns_under_pmc": sum(t for _, t in vals) / len(vals)}
        if name in write:
            w = write[name]
            e["write_raw_bytes"] = sum(v for v, _ in w) / len(w) * 1024.0
        out["kernels"][name] = e
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(points), {k: round(v["fetch_bytes"] / 1e9, 2) for k, v in out["kernels"].items()})


if __name__ == "__main__":
    main()
