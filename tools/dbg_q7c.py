import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lingo-db_amd"), ROOT]
import numpy as np
import torch; torch.cuda.set_device(0); x = torch.zeros(10, device="cuda")
import lingodb_amd as ldb
from lingodb_amd import api, capi
n = 15_000_000
ctx = ldb.Context(0)
li = ctx.tpch_generate(0, n, cols=[0, 2, 5, 6, 10]); su = ctx.tpch_generate(4, n, cols=[0, 1]); na = ctx.tpch_generate(6, n, cols=[0, 1, 2])
od = ctx.tpch_generate(1, n, cols=[0, 1, 4, 6]); cu = ctx.tpch_generate(2, n, cols=[0, 1, 3])
mode = os.environ.get("PRE", "q3")
L = capi.gpu_lib()
if mode == "q3": ctx.plan_q3(cu, od, li).to_arrow()
elif mode == "q3_nojit":
    L.ldb_gpu_set_option(b"jit", 0); ctx.plan_q3(cu, od, li).to_arrow(); L.ldb_gpu_set_option(b"jit", 1)
elif mode == "q3_nolazy":
    L.ldb_gpu_set_option(b"lazy_filter", 0); ctx.plan_q3(cu, od, li).to_arrow(); L.ldb_gpu_set_option(b"lazy_filter", 1)
elif mode == "probe_only":
    o1 = od.rel().scan_filter([api.pred((0, 2), capi.F_LT, 9204)])
    ho = o1.join_build([(0, 0)], unique=True)
    l0 = li.rel().scan_filter([api.pred((0, 4), capi.F_GT, 9204)])
    print("pre rows", ho.probe(l0, [(0, 0)]).rows)
elif mode == "alloc_only":
    big = [ctx.tpch_generate(0, n, cols=[0]) for _ in range(3)]; del big
H = capi.host_lib()
t = C.c_void_p(); capi.check_plan(H.ldb_plan_tpch_q7_customers(ctx.h, cu.h, na.h, C.byref(t))); custs = api.Table(ctx, t)
t = C.c_void_p(); capi.check_plan(H.ldb_plan_tpch_q7_suppliers(ctx.h, su.h, na.h, C.byref(t))); supps = api.Table(ctx, t)
hs = supps.rel().join_build([(0, 0)], unique=True)
res = []
for it in range(3):
    l1 = li.rel().scan_filter([api.pred((0, 4), capi.F_GTE, 9131), api.pred((0, 4), capi.F_LTE, 9861)])
    ls = hs.probe(l1, [(0, 1)])
    b = ls.rowids(1)
    res.append((ls.rows, int((b >= supps.rows).sum())))
print("runs", res)
