#!/usr/bin/env python3
"""Micro-benchmark of string conjuncts over p_name (TPC-H part, generated on the device):
   python tools/like_bench.py [n_orders]        (LDB_JIT=0 for the generic kernels)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lingo-db_amd"))
import lingodb_amd as ldb  # noqa: E402
from lingodb_amd import api, capi  # noqa: E402

n_orders = int(sys.argv[1]) if len(sys.argv) > 1 else 150_000_000
ctx = ldb.Context(0)
part = ctx.tpch_generate(3, n_orders, cols=[0, 3])
rel = part.rel()
ctx.prof_enable(True)
for name, plist in [("like %green%", [api.pred((0, 1), capi.F_LIKE, "%green%")]), ("like green%", [api.pred((0, 1), capi.F_LIKE, "green%")]),
                    ("like %", [api.pred((0, 1), capi.F_LIKE, "%")]), ("eq", [api.pred((0, 1), capi.F_EQ, "green")]), ("key >= 0", [api.pred((0, 0), capi.F_GTE, 0)])]:
    for kind in ("count", "filter"):
        ctx.prof_reset()
        for _ in range(3):
            r = rel.scan_count(plist) if kind == "count" else rel.scan_filter(plist).rows
        prof = ctx.prof_all()
        print(f"{name:14s} {kind:6s} rows={r:10d} " + " ".join(f"{k}={v[1] / v[0]:.3f}ms" for k, v in prof.items()), flush=True)
