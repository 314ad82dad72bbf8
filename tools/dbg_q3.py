import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lingo-db_amd"), ROOT]
import lingodb_amd as ldb
sf = float(sys.argv[1]); n = int(sf * 1_500_000)
ctx = ldb.Context(0)
li = ctx.tpch_generate(0, n, cols=[0, 4, 5, 6, 7, 8, 9, 10]); od = ctx.tpch_generate(1, n, cols=[0, 1, 4, 6]); cu = ctx.tpch_generate(2, n, cols=[0, 3])
print("gen ok", flush=True)
step = sys.argv[2] if len(sys.argv) > 2 else "q3"
from lingodb_amd import api, capi
if step == "q3":
    print(ctx.plan_q3(cu, od, li).to_arrow().num_rows, flush=True)
else:
    c1 = cu.rel().scan_filter([api.pred((0, 1), capi.F_EQ, "BUILDING")]); print("c1", flush=True)
    o1 = od.rel().scan_filter([api.pred((0, 2), capi.F_LT, 9204)]); print("o1", flush=True)
    hc = c1.join_build([(0, 0)], unique=True); ctx.sync(); print("hc", hc.slots, flush=True)
    co = hc.probe(o1, [(0, 1)]); print("co", co.rows, flush=True)
    ho = co.join_build([(0, 0)], unique=True); ctx.sync(); print("ho", ho.slots, flush=True)
    l1 = li.rel().scan_filter([api.pred((0, 7), capi.F_GT, 9204)])
    lco = ho.probe(l1, [(0, 0)]); print("lco", lco.rows, flush=True)
