"""g1 evidence: the radix-partitioned probe against the direct probe, UNCLUSTERED foreign keys (600 M random
l_orderkey-like keys → 150 M orders at SF100), for both table layouts (ordered 8-byte slots; the rank table),
over partition sizes from LDS-sized to L2/MALL-sized.  Prints one JSON document: per configuration the
per-kernel milliseconds (k_radix_hist, k_radix_scatter, the probe kernel, helpers), their sum, the algorithmic
bytes of each pass and the matches (must equal the direct probe's).
usage: python tools/radix_sweep.py [SF=100] [tables=0,1]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lingo-db_amd"), ROOT]
import lingodb_amd as ldb  # noqa: E402
from lingodb_amd import capi  # noqa: E402

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
n = int(sf * 1_500_000)
ctx = ldb.Context(0)
ctx.prof_enable(True)
L = capi.gpu_lib()
od = ctx.tpch_generate(1, n, cols=[0])
pk = ctx.tpch_generate(8, n, cols=[0])  # lineitem-sized column of uniformly random order keys
rows = pk.rows
out = {"sf": sf, "probe_rows": rows, "build_rows": od.rows, "runs": []}


def timed(reps=3):
    ht.probe_count(pk.rel(), [(0, 0)])
    ctx.prof_reset()
    for _ in range(reps):
        m = ht.probe_count(pk.rel(), [(0, 0)])
    pr = ctx.prof_all()
    ks = {k: round(v[1] / reps, 3) for k, v in pr.items()}
    return ks, round(sum(ks.values()), 3), m


tables = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]  # 0 = ordered 8-byte slots, 1 = rank table
for rank in tables:
    L.ldb_gpu_set_option(b"join_rank", rank)
    L.ldb_gpu_set_option(b"join_direct", rank)
    ht = od.rel().join_build([(0, 0)], unique=True)
    tb = ht.table_bytes
    L.ldb_gpu_set_option(b"join_radix", 0)
    ks, total, m0 = timed()
    out["runs"].append({"table": "rank" if rank else "ordered", "table_bytes": tb, "radix": False, "kernels_ms": ks, "total_ms": total, "matches": m0,
                        "bytes": {"probe": rows * 4 + tb}})
    L.ldb_gpu_set_option(b"join_radix", 1)
    for pb in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26):
        if tb // pb > 4096 * 1:  # RX_MAX_PARTS partitions at most
            continue
        L.ldb_gpu_set_option(b"join_radix_part_bytes", pb)
        for wc, lds in (((1, 1), (1, 0), (0, 0)) if rank else ((0, 0),)):  # round 4: write-combining partition (ldb_wc.hip) [+ LDS-staged probe] beside the one-pass cursor scatter
            L.ldb_gpu_set_option(b"join_radix_wc", wc)
            L.ldb_gpu_set_option(b"join_radix_lds", lds)
            ks, total, m = timed()
            assert m == m0, (m, m0)
            parts = 16
            while parts < 4096 and tb // parts > pb:
                parts *= 2
            passes = 2 if (wc and parts > 64) else 1
            # bytes: histogram reads the keys; scatter pass 1 reads keys, writes (key, row); pass 2 reads the keys again for its histogram, then reads and writes (key, row)
            by = {"hist": rows * 4 * passes, "scatter": rows * (4 + 8) + (rows * 16 if passes == 2 else 0), "probe": rows * 8 + tb}
            out["runs"].append({"table": "rank" if rank else "ordered", "table_bytes": tb, "radix": True, "write_combining": bool(wc), "lds_staged_probe": bool(lds and pb <= 65536), "partitions": parts, "passes": passes, "part_bytes": pb,
                                "kernels_ms": ks, "total_ms": total, "matches": m, "bytes": by})
        L.ldb_gpu_set_option(b"join_radix_wc", 1)
        L.ldb_gpu_set_option(b"join_radix_lds", 1)
    L.ldb_gpu_set_option(b"join_radix", 0)
    del ht
print(json.dumps(out, indent=1))
