import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lingo-db_amd"), ROOT]
import lingodb_amd as ldb
from lingodb_amd import api, capi
n = int(float(sys.argv[1]) * 1_500_000)
ctx = ldb.Context(0); ctx.prof_enable(True); L = capi.gpu_lib()
od = ctx.tpch_generate(1, n, cols=[0]); pk = ctx.tpch_generate(8, n, cols=[0])
ht = od.rel().join_build([(0, 0)], unique=True)
L.ldb_gpu_set_option(b"join_radix", 1)
for pb in [1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26, 1 << 28]:
    L.ldb_gpu_set_option(b"join_radix_part_bytes", pb)
    ht.probe_count(pk.rel(), [(0, 0)]); ctx.prof_reset()
    for _ in range(3): m = ht.probe_count(pk.rel(), [(0, 0)])
    pr = ctx.prof_all()
    print(pb >> 10, "KB/part", {k: round(v[1] / 3, 3) for k, v in pr.items()}, "total", round(sum(v[1] for v in pr.values()) / 3, 3), m, flush=True)
