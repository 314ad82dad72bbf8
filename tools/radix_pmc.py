"""The unclustered FK probe of the bench (600 M random order keys → the 150 M-order rank table) three times, for PMC passes
over its kernels (k_wc_hist, k_wc_scatter, k_join_probe_lds with join_radix = -1; k_join_probe_count with LDB_JOIN_RADIX=0):
  cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python tools/radix_pmc.py 100"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lingo-db_amd"), ROOT]
import lingodb_amd as ldb  # noqa: E402

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
n = int(sf * 1_500_000)
ctx = ldb.Context(0)
od = ctx.tpch_generate(1, n, cols=[0])
pk = ctx.tpch_generate(8, n, cols=[0])
ht = od.rel().join_build([(0, 0)], unique=True)
for _ in range(3):
    m = ht.probe_count(pk.rel(), [(0, 0)])
assert m == pk.rows, (m, pk.rows)
ctx.sync()
