"""Round 6 (verdict r5 #8): the compressed resident format, narrow level 2 — decimals of precision < 19 at the narrowest of 1 / 2 / 4 / 8 bytes their column's
value range allows, char(1) at one byte.  Kernels widen in registers and compute in i64 / i128 exactly as before (LowerToStd.cpp:128-132,1479-1486), so
every result must be bit-identical to the Arrow-width run: the registration path on hand-made columns (widths chosen from the values, NULLs, negative
values, export widening), and all 22 TPC-H plans at SF 0.2 on generated tables against the oracle legs, on specialised kernels."""
import decimal

import numpy as np
import pyarrow as pa
import pytest

import tpch_legs
import tpch_plans
from lingodb_amd import api, capi
from test_gpu_sf1_oracle import assert_matches_legs, canon

pytestmark = pytest.mark.gpu


def test_registration_picks_the_narrowest_width_and_results_do_not_change(ctx):
    rng = np.random.default_rng(62)
    n = 100_000
    D = lambda vals, p=12, s=2: pa.array([None if v is None else decimal.Decimal(int(v)).scaleb(-s) for v in vals], pa.decimal128(p, s))
    tiny = rng.integers(0, 11, n)  # fits one byte
    small = rng.integers(-30000, 30000, n)  # two bytes, negative values
    mid = rng.integers(-2_000_000_000, 2_000_000_000, n)  # four
    big = rng.integers(-(1 << 40), 1 << 40, n)  # eight
    wide = [int(v) * 10**12 for v in rng.integers(-(1 << 40), 1 << 40, n)]  # decimal(38, 2): never narrowed
    nul = [None if i % 7 == 0 else int(v) for i, v in enumerate(rng.integers(0, 100, n))]
    flag = np.frombuffer(bytes(b for x in rng.integers(0, 3, n) for b in (b"AFO"[x], 0, 0, 0)), dtype=np.uint8)
    t = pa.table({"k": pa.array(rng.integers(0, 50, n).astype(np.int32)), "tiny": D(tiny), "small": D(small), "mid": D(mid), "big": D(big, 18, 2),
                  "wide": pa.array([decimal.Decimal(v).scaleb(-2) for v in wide], pa.decimal128(38, 2)), "nul": D(nul),
                  "flag": pa.FixedSizeBinaryArray.from_buffers(pa.binary(4), n, [None, pa.py_buffer(flag.tobytes())])})
    full = ctx.register("n2_full", t, 0)
    comp = ctx.register("n2_comp", t, 2)
    widths = {name: comp.col_width(comp.col(name)) for name in t.column_names}
    assert widths == {"k": 4, "tiny": 1, "small": 2, "mid": 4, "big": 8, "wide": 16, "nul": 1, "flag": 1}, widths
    assert [full.col_width(i) for i in range(full.n_cols)] == [4, 16, 16, 16, 16, 16, 16, 4]
    assert comp.to_arrow().equals(full.to_arrow()) and comp.to_arrow().equals(t)  # export widens back to the Arrow widths
    plan = {"steps": [{"op": "filter", "in": "t", "out": "f", "preds": [{"col": "tiny", "op": "LTE", "value": "0.07"}, {"col": "small", "op": "GT", "value": "-250.00"}, {"col": "flag", "op": "NEQ", "value": "O"}]},
                      {"op": "groupby", "in": "f", "keys": ["k", "flag"], "aggs": [{"fn": "sum", "expr": {"mul": ["mid", {"sub": [1, "tiny"]}]}, "as": "a"}, {"fn": "sum", "expr": "big", "as": "b"},
                                                                                   {"fn": "sum", "expr": "wide", "as": "w"}, {"fn": "min", "expr": "small", "as": "lo"}, {"fn": "max", "expr": "nul", "as": "hi"},
                                                                                   {"fn": "count", "expr": "nul", "as": "c"}, {"fn": "avg", "expr": "small", "as": "avg"}], "est_groups": 200, "out": "g"},
                      {"op": "sort", "in": "g", "by": ["k", "flag"], "out": "s"},
                      {"op": "materialize", "in": "s", "cols": ["k", "flag", "a", "b", "w", "lo", "hi", "c", "avg"], "out": "result"}], "result": "result"}
    import json

    a = ctx.run_plan(json.dumps(plan), {"t": full}).to_arrow()
    b = ctx.run_plan(json.dumps(plan), {"t": comp}).to_arrow()
    assert a.num_rows > 50 and a.equals(b)
    # a narrowed decimal as a join key and as a sort key
    jp = {"steps": [{"op": "groupby", "in": "t", "keys": ["tiny"], "aggs": [{"fn": "count_star", "as": "n"}], "est_groups": 16, "key_names": ["gt"], "out": "g"},
                    {"op": "join_build", "in": "g", "keys": ["gt"], "unique": True, "out": "h"}, {"op": "join_probe", "ht": "h", "in": "t", "keys": ["tiny"], "kind": "inner", "out": "j"},
                    {"op": "topk", "in": "j", "by": [{"col": "small", "desc": True}, "mid"], "k": 50, "out": "top"},
                    {"op": "materialize", "in": "top", "cols": ["small", "mid", "tiny", "n", "flag"], "out": "result"}], "result": "result"}
    assert ctx.run_plan(json.dumps(jp), {"t": full}).to_arrow().equals(ctx.run_plan(json.dumps(jp), {"t": comp}).to_arrow())


N_ORDERS = 300_000  # SF 0.2


@pytest.fixture(scope="module")
def world(ctx):
    lib = capi.gpu_lib()
    lib.ldb_gpu_set_option(b"jit_min_rows", 0)  # specialised kernels: the widths are compile-time constants there
    lib.ldb_gpu_set_option(b"lazy_min_rows", 0)
    db = tpch_plans.Database(ctx, N_ORDERS, 0, 1, list(range(1, 23)), 2)
    assert db.lineitem.col_width(db.lineitem.col("l_discount")) == 1 and db.lineitem.col_width(db.lineitem.col("l_quantity")) == 2
    assert db.lineitem.col_width(db.lineitem.col("l_extendedprice")) == 4 and db.lineitem.col_width(db.lineitem.col("l_returnflag")) == 1
    yield tpch_plans.Runner(ctx, db, 1, None, None), tpch_legs.Legs(N_ORDERS)
    lib.ldb_gpu_set_option(b"jit_min_rows", 4000000)
    lib.ldb_gpu_set_option(b"lazy_min_rows", 1 << 20)


@pytest.mark.parametrize("q", list(range(1, 23)))
def test_tpch_on_the_compressed_format_matches_the_oracle(world, q):
    runner, legs = world
    got = canon(runner.run(q).to_arrow())
    want = legs.run(q)
    if q == 11 and not want:
        assert not got
        return
    assert want, "empty oracle result: the check would be vacuous"
    assert_matches_legs(q, got, want)
