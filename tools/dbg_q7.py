import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lingo-db_amd"), ROOT]
import numpy as np
import lingodb_amd as ldb
from lingodb_amd import api, capi
sf = float(sys.argv[1]); n = int(sf * 1_500_000)
pre = sys.argv[2]
ctx = ldb.Context(0)
li = ctx.tpch_generate(0, n, cols=[0, 2, 5, 6, 10]); od = ctx.tpch_generate(1, n, cols=[0, 1, 4, 6]); cu = ctx.tpch_generate(2, n, cols=[0, 1, 3])
su = ctx.tpch_generate(4, n, cols=[0, 1])
if pre == "q3":
    print("q3", ctx.plan_q3(cu, od, li).to_arrow().num_rows, flush=True)
def first_probe(tag):
    s1 = su.rel().scan_filter([api.pred((0, 1), capi.F_IN, values=[6, 7])])
    supps = s1.materialize([(0, 0), (0, 1)])
    hs = supps.rel().join_build([(0, 0)], unique=True)
    l1 = li.rel().scan_filter([api.pred((0, li.col("l_shipdate")), capi.F_GTE, 9131), api.pred((0, li.col("l_shipdate")), capi.F_LTE, 9861)])
    ls = hs.probe(l1, [(0, li.col("l_suppkey"))])
    a, b = ls.rowids(0), ls.rowids(1)
    print(tag, "rows", ls.rows, "max probe id", a.max(), "of", li.rows, "max build id", b.max(), "of", supps.rows, flush=True)
    return a, b
a1, b1 = first_probe("lazy")
capi.gpu_lib().ldb_gpu_set_option(b"lazy_filter", 0)
a2, b2 = first_probe("forced")
print("equal", np.array_equal(a1, a2), np.array_equal(b1, b2))
if not np.array_equal(b1, b2):
    bad = np.nonzero(b1 != b2)[0]; print("n bad", len(bad), bad[:10], b1[bad[:10]], b2[bad[:10]], a1[bad[:10]])
