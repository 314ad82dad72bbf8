"""f3 evidence: a range predicate on the SORTED l_orderkey of SF100 lineitem (600 M rows) plus one on l_quantity,
count-only scan, with and without the zone map of l_orderkey (selectivity 1 % … 100 % of the key range).
Prints JSON: per selectivity the scan kernel's milliseconds with zones on / off and the matching row counts.
usage: python tools/zone_bench.py [SF=100]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lingo-db_amd"), ROOT]
import lingodb_amd as ldb  # noqa: E402
from lingodb_amd import api, capi  # noqa: E402

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
n = int(sf * 1_500_000)
ctx = ldb.Context(0)
ctx.prof_enable(True)
L = capi.gpu_lib()
li = ctx.tpch_generate(0, n, cols=[0, 4])  # l_orderkey, l_quantity
kcol, qcol = li.col("l_orderkey"), li.col("l_quantity")
kmax = 4 * n  # sparse order keys
out = {"sf": sf, "rows": li.rows, "zones": L.ldb_gpu_table_zones(ctx.h, li.h, kcol), "runs": []}
for frac in (0.01, 0.1, 0.5, 1.0):
    row = {"key_fraction": frac}
    for zm in (1, 0):
        L.ldb_gpu_set_option(b"zone_maps", zm)
        preds = [api.pred((0, kcol), capi.F_LT, int(kmax * frac)), api.pred((0, qcol), capi.F_LT, 2400)]
        cnt = li.rel().scan_filter(preds).rows
        ctx.prof_reset()
        for _ in range(5):
            cnt = li.rel().scan_filter(preds).rows
        pr = ctx.prof_all()
        row["zones_on" if zm else "zones_off"] = {k: round(v[1] / 5, 3) for k, v in pr.items()}
        row["rows_on" if zm else "rows_off"] = cnt
    assert row["rows_on"] == row["rows_off"]
    out["runs"].append(row)
L.ldb_gpu_set_option(b"zone_maps", 1)
print(json.dumps(out, indent=1))
