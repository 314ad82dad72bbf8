"""debug: pair32 tables with / without a residual conjunct (GPU run 4)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lingo-db_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np, pyarrow as pa
import lingodb_amd as ldb
from lingodb_amd import api, capi

ctx = ldb.Context(0)
lib = capi.gpu_lib()
rng = np.random.default_rng(5)
nb, npr = 5000, 20000
combos = rng.permutation(100 * 100)[:nb]
ba, bb = combos // 100, combos % 100
pa_, pb_ = rng.integers(0, 110, npr), rng.integers(0, 100, npr)
for typ_b in (pa.int32(), pa.date32()):
    b = ctx.register("b", pa.table({"a": pa.array(ba, pa.int32()), "b": pa.array(bb.astype(np.int32), pa.int32()).cast(typ_b), "x": pa.array(rng.integers(0, 10, nb), pa.int32())}))
    p = ctx.register("p", pa.table({"a": pa.array(pa_, pa.int32()), "b": pa.array(pb_.astype(np.int32), pa.int32()).cast(typ_b), "x": pa.array(rng.integers(0, 10, npr), pa.int32())}))
    keys = [(0, 0), (0, 1)]
    for unique in (True, False):
        for resid in ([], [((0, 2), capi.F_NEQ, (0, 2))], [((0, 2), capi.F_EQ, (0, 2))], [((0, 0), capi.F_EQ, (0, 0))]):
            out = {}
            for layout in (1, 0):
                lib.ldb_gpu_set_option(b"join_pair32", layout)
                ht = b.rel().join_build(keys, unique=unique)
                out[layout] = (ht.probe(p.rel(), keys, capi.JOIN_INNER, residual=resid).rows, ht.probe(p.rel(), keys, capi.JOIN_SEMI, residual=resid).rows, ht.table_bytes)
            print(str(typ_b), "unique" if unique else "general", "resid", [(r[0], r[1], r[2]) for r in resid], "pair32:", out[1], "plain:", out[0], "OK" if out[1][:2] == out[0][:2] else "DIFFERENT", flush=True)
lib.ldb_gpu_set_option(b"join_pair32", 1)
