import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lingo-db_amd"), ROOT]
import numpy as np
import lingodb_amd as ldb
from lingodb_amd import api, capi
sf = float(sys.argv[1]); n = int(sf * 1_500_000)
if "torch" in sys.argv:
    import torch; torch.cuda.set_device(0); x = torch.zeros(10, device="cuda")
ctx = ldb.Context(0)
if "prof" in sys.argv: ctx.prof_enable(True)
li = ctx.tpch_generate(0, n, cols=[0, 2, 5, 6, 10]); od = ctx.tpch_generate(1, n, cols=[0, 1, 4, 6]); cu = ctx.tpch_generate(2, n, cols=[0, 1, 3])
su = ctx.tpch_generate(4, n, cols=[0, 1]); na = ctx.tpch_generate(6, n, cols=[0, 1, 2])
if "q3" in sys.argv: print("q3", ctx.plan_q3(cu, od, li).to_arrow().num_rows, flush=True)
H = capi.host_lib()
def plan(fn, *tabs):
    t = C.c_void_p(); capi.check_plan(fn(ctx.h, *[x.h for x in tabs], C.byref(t))); return api.Table(ctx, t)
if "cust" in sys.argv: custs = plan(H.ldb_plan_tpch_q7_customers, cu, na); print("custs", custs.rows, flush=True)
supps = plan(H.ldb_plan_tpch_q7_suppliers, su, na); print("supps", supps.rows, flush=True)
sk = np.frombuffer(supps.read_fixed(0).tobytes(), dtype=np.int32); print("supp keys", sk.min(), sk.max(), flush=True)
hs = supps.rel().join_build([(0, 0)], unique=True)
l1 = li.rel().scan_filter([api.pred((0, li.col("l_shipdate")), capi.F_GTE, 9131), api.pred((0, li.col("l_shipdate")), capi.F_LTE, 9861)])
ls = hs.probe(l1, [(0, li.col("l_suppkey"))])
a, b = ls.rowids(0), ls.rowids(1)
print("rows", ls.rows, "max probe id", a.max(), "of", li.rows, "max build id", b.max(), "of", supps.rows, flush=True)
bad = np.nonzero(b >= supps.rows)[0]; print("bad", len(bad), bad[:8], b[bad[:8]], a[bad[:8]])
# reference result with the filter forced (different kernel path)
capi.gpu_lib().ldb_gpu_set_option(b"lazy_filter", 0)
l2 = li.rel().scan_filter([api.pred((0, li.col("l_shipdate")), capi.F_GTE, 9131), api.pred((0, li.col("l_shipdate")), capi.F_LTE, 9861)])
sel = l2.rowids(0)
good = hs.probe(l2, [(0, li.col("l_suppkey"))])
ga, gb = good.rowids(0), good.rowids(1)
print("good rows", good.rows)
gmap = dict(zip(ga.tolist(), gb.tolist()))
missing = sorted(set(ga.tolist()) - set(a.tolist()))
print("missing", len(missing), missing[:10], [m % 2048 for m in missing[:20]])
wrong = [(int(x), int(y), gmap.get(int(x))) for x, y in zip(a.tolist(), b.tolist()) if gmap.get(int(x)) != int(y)]
print("wrong", len(wrong), wrong[:10], [w[0] % 2048 for w in wrong[:20]])
# position of the wrong rows inside their tile's queue
selset = np.zeros(li.rows, dtype=bool); selset[sel] = True
for w in wrong[:6] + [(m, 0, 0) for m in missing[:6]]:
    row = w[0]; base = row // 2048 * 2048
    tile_sel = np.nonzero(selset[base:base + 2048])[0]
    print("row", row, "rel", row - base, "tile passing", len(tile_sel), "rank in tile", int(np.searchsorted(tile_sel, row - base)))
