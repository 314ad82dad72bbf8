"""debug: Q10 (and Q3 / Q4 / Q21) at a given SF with the row-id checks on, generic and specialised kernels"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lingo-db_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import lingodb_amd as ldb
import tpch_plans
from lingodb_amd import capi

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
qs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "10,3,4,21").split(",")]
ctx = ldb.Context(0)
lib = capi.gpu_lib()
db = tpch_plans.Database(ctx, int(sf * 1_500_000), 0, 1, qs, 0)
runner = tpch_plans.Runner(ctx, db, 1, None, None)
lib.ldb_gpu_set_option(b"debug_check", 1)
for mode in ("generic-async", "spec"):
    for q in qs:
        try:
            for rep in range(3):
                t = runner.run(q).to_arrow()
                ctx.sync()
            print(mode, "Q%d" % q, "rows", t.num_rows, "ok", flush=True)
        except Exception as e:
            print(mode, "Q%d" % q, "FAILED", str(e)[:300], flush=True)
            sys.exit(1)
    tpch_plans._jit_wait()
