"""Plan-interpreter features the round-1 plans needed when they became data (lingo-db_amd/plans/tpch):
group estimates taken from other values, a scalar subquery that returns no row.  The plans
themselves are checked against independent evaluations of the SQL text in test_gpu_parity /
test_gpu_tpch_more / test_gpu_z_tpch_q10 / test_gpu_tpch_new (small scale, both kernel modes) and
against the oracle legs at SF1 in test_gpu_sf1_oracle."""
import json

import pyarrow as pa
import pyarrow.compute  # noqa: F401
import pytest

import tpch_data

pytestmark = pytest.mark.gpu


def test_q15_without_lineitems_in_the_quarter(ctx):
    """the scalar subquery returns no row: MAX is NULL, `= NULL` keeps nothing (the plan's scalar filter must not fail)"""
    T = tpch_data
    li = T.host_table(T.LINEITEM, 20_000, cols=[2, 5, 6, 10])
    su = T.host_table(T.SUPPLIER, 20_000, cols=[0, 1])
    from test_gpu_tpch_more import days

    none = li.filter(pa.compute.less(li.column("l_shipdate"), pa.scalar(days("1996-01-01"), pa.int32()).cast(pa.date32())))
    res = ctx.run_plan("tpch/q15.json", {"supplier": ctx.register("q15j_su", su), "lineitem": ctx.register("q15j_li", none)})
    assert res.rows == 0


def test_group_estimates_from_other_values(ctx):
    """est_groups as {"rows_of", "div", "min"}: a far too low estimate only costs a retry"""
    T = tpch_data
    li = T.host_table(T.LINEITEM, 50_000, cols=[0, 4])
    na = T.host_table(T.NATION, 50_000, cols=[0, 1, 2])
    plan = {"steps": [{"op": "groupby", "in": "lineitem", "keys": ["l_orderkey"], "aggs": [{"fn": "count_star", "as": "n"}],
                       "est_groups": {"rows_of": "nation", "div": 5, "min": 2}, "out": "g"},
                      {"op": "groupby", "in": "g", "keys": [], "aggs": [{"fn": "sum", "expr": "n", "as": "rows"}, {"fn": "count_star", "as": "groups"}], "est_groups": 1, "out": "result"}],
            "result": "result"}
    got = ctx.run_plan(json.dumps(plan), {"lineitem": ctx.register("eg_li", li), "nation": ctx.register("eg_na", na)}).to_arrow()
    assert got.column(0).to_pylist() == [li.num_rows] and got.column(1).to_pylist() == [50_000]


def test_scalar_subquery_over_no_rows_is_null(ctx):
    """a key-less aggregate over an empty input yields ONE row holding NULL (SimpleState): a comparison with that
    NULL keeps nothing — Q22 with no customer above 0.00, Q11 with no supplier of the nation"""
    t = pa.table({"k": pa.array([1, 2, 3, 4], pa.int32()), "v": pa.array([10, 20, 30, 40], pa.int64())})
    plan = {"steps": [{"op": "filter", "in": "t", "out": "none", "preds": [{"col": "v", "op": "GT", "value": 1000}]},
                      {"op": "groupby", "in": "none", "keys": [], "aggs": [{"fn": "sum", "expr": "v", "as": "s"}], "est_groups": 1, "out": "total"},
                      {"op": "filter", "in": "t", "out": "kept", "preds": [{"col": "v", "op": "GT", "scalar": {"from": "total", "col": "s"}}]},
                      {"op": "materialize", "in": "kept", "cols": ["k"], "out": "result"}], "result": "result"}
    dev = ctx.register("scalar_null_t", t)
    assert ctx.run_plan(json.dumps(plan), {"t": dev}).rows == 0
    plan["steps"][0]["preds"][0]["value"] = 25  # a real scalar: sum(30, 40) = 70 keeps nothing either; 35 > … check a passing one
    plan["steps"][2]["preds"][0]["op"] = "LT"
    assert ctx.run_plan(json.dumps(plan), {"t": dev}).to_arrow().column(0).to_pylist() == [1, 2, 3, 4]


def test_avg_merges_partial_sums_and_counts(ctx):
    """{"fn": "avg", "expr": sum, "count": n}: the merge of exchanged (sum, count) partials equals AVG over the rows"""
    import decimal

    rows = pa.table({"g": pa.array([1, 1, 2, 2, 2, 3], pa.int32()),
                     "d": pa.array([decimal.Decimal(x) / 100 for x in (101, 250, 399, 1, 77, 5000)], pa.decimal128(12, 2)), "part": pa.array([0, 1, 0, 0, 1, 1], pa.int32())})
    direct = {"steps": [{"op": "groupby", "in": "t", "keys": ["g"], "aggs": [{"fn": "avg", "expr": "d", "as": "a"}], "est_groups": 4, "out": "x"},
                        {"op": "sort", "in": "x", "by": ["g"], "out": "xs"}, {"op": "materialize", "in": "xs", "cols": ["g", "a"], "out": "result"}], "result": "result"}
    merged = {"steps": [{"op": "groupby", "in": "t", "keys": ["g", "part"], "aggs": [{"fn": "sum", "expr": "d", "as": "s"}, {"fn": "count_star", "as": "n"}], "est_groups": 8, "out": "p"},
                        {"op": "groupby", "in": "p", "keys": ["g"], "aggs": [{"fn": "avg", "expr": "s", "count": "n", "as": "a"}], "est_groups": 4, "out": "x"},
                        {"op": "sort", "in": "x", "by": ["g"], "out": "xs"}, {"op": "materialize", "in": "xs", "cols": ["g", "a"], "out": "result"}], "result": "result"}
    dev = ctx.register("avg_merge_t", rows)
    a, b = ctx.run_plan(json.dumps(direct), {"t": dev}).to_arrow(), ctx.run_plan(json.dumps(merged), {"t": dev}).to_arrow()
    assert a.schema.field(1).type == b.schema.field(1).type and a.to_pylist() == b.to_pylist() and a.num_rows == 3


def test_unique_join_over_a_base_table_uses_the_tables_index(ctx):
    """join_build over a bare base table with a unique key = the table's persistent hash index (ldb_gpu_table_index, the
    counterpart of LingoDBHashIndex + index nested-loop joins): built by the first plan run, reused by the next ones"""
    T = tpch_data
    od = ctx.register("ix_orders", T.host_table(T.ORDERS, 30_000, cols=[0, 1]))
    li = ctx.register("ix_lineitem", T.host_table(T.LINEITEM, 30_000, cols=[0, 4]))
    plan = json.dumps({"steps": [{"op": "join_build", "in": "orders", "keys": ["o_orderkey"], "unique": True, "out": "h"},
                                 {"op": "join_probe", "ht": "h", "in": "lineitem", "keys": ["l_orderkey"], "kind": "inner", "out": "j"},
                                 {"op": "groupby", "in": "j", "keys": [], "aggs": [{"fn": "count_star", "as": "n"}, {"fn": "sum", "expr": "o_custkey", "as": "s"}], "est_groups": 1, "out": "result"}],
                       "result": "result"})
    ctx.prof_enable(True)
    ctx.prof_reset()
    first = ctx.run_plan(plan, {"orders": od, "lineitem": li}).to_arrow().to_pylist()
    builds_first = ctx.prof_all().get("k_join_build", (0, 0.0))[0]
    ctx.prof_reset()
    second = ctx.run_plan(plan, {"orders": od, "lineitem": li}).to_arrow().to_pylist()
    builds_second = ctx.prof_all().get("k_join_build", (0, 0.0))[0]
    ctx.prof_enable(False)
    assert first == second and first[0]["n"] == li.rows
    assert builds_first >= 1 and builds_second == 0
    no_index = json.loads(plan)
    no_index["steps"][0]["index"] = False  # an emitter may opt out (e.g. a table about to change)
    assert ctx.run_plan(json.dumps(no_index), {"orders": od, "lineitem": li}).to_arrow().to_pylist() == first


def test_long_in_lists_and_long_conjunctions(ctx):
    """round 6 (verdict r5 missing #6): the reference's Restrictions take an IN list of any size (a hash set, Restrictions.cpp:481-515) and any number of
    conjuncts; one scan_filter call takes eight of each.  The interpreter applies a long conjunction eight conjuncts at a time and turns a long
    integer IN list into a semi join against the table of its constants — same rows as numpy, also as a prepared plan executed three times (the second
    and third execution replay)"""
    import numpy as np

    rng = np.random.default_rng(6)
    n = 300_000
    cols = {"c%d" % i: rng.integers(0, 100, n).astype(np.int32) for i in range(11)}
    cols["d"] = rng.integers(8000, 11000, n).astype(np.int32)
    t = pa.table({**{k: pa.array(v, pa.int32()) for k, v in cols.items() if k != "d"}, "d": pa.array(cols["d"], pa.int32()).cast(pa.date32()), "i": pa.array(np.arange(n, dtype=np.int64))})
    dev = ctx.register("long_in_t", t)
    members = [3, 5, 8, 13, 21, 34, 55, 89, 1, 2, 97, 96, 95, 5, 5, 40, 41, 42, 43]  # 19 constants, duplicates among them
    days = sorted(set(int(x) for x in rng.integers(8000, 11000, 30)))
    import datetime

    iso = [(datetime.date(1970, 1, 1) + datetime.timedelta(days=x)).isoformat() for x in days]
    preds = [{"col": "c0", "op": "IN", "values": members}, {"col": "d", "op": "IN", "values": iso}] + [{"col": "c%d" % i, "op": "GTE" if i % 2 else "LTE", "value": 5 if i % 2 else 95} for i in range(1, 11)]
    plan = {"steps": [{"op": "filter", "in": "t", "out": "f", "preds": preds}, {"op": "materialize", "in": "f", "cols": ["i"], "out": "result"}], "result": "result"}
    keep = np.isin(cols["c0"], members) & np.isin(cols["d"], days)
    for i in range(1, 11):
        keep &= (cols["c%d" % i] >= 5) if i % 2 else (cols["c%d" % i] <= 95)
    want = np.nonzero(keep)[0].tolist()
    assert 0 < len(want) < n // 20
    assert ctx.run_plan(json.dumps(plan), {"t": dev}).to_arrow().column(0).to_pylist() == want
    prepared = ctx.prepare_plan(json.dumps(plan))
    for _ in range(3):
        assert prepared.execute({"t": dev}).to_arrow().column(0).to_pylist() == want
    assert prepared.stats()["replays"] >= 1
    prepared.release()
    # only the long lists alone, and an empty result
    only = {"steps": [{"op": "filter", "in": "t", "out": "f", "preds": [{"col": "c0", "op": "IN", "values": list(range(200, 212))}]}, {"op": "materialize", "in": "f", "cols": ["i"], "out": "result"}], "result": "result"}
    assert ctx.run_plan(json.dumps(only), {"t": dev}).rows == 0


def test_inner_join_with_more_than_two_residual_conjuncts(ctx):
    """round 6: the probe kernel checks two residual conjuncts on the candidate pair; an inner join's further conjuncts are applied to the joined rows"""
    import numpy as np

    rng = np.random.default_rng(61)
    nb, n = 5_000, 200_000
    build = pa.table({"bk": pa.array(np.arange(nb, dtype=np.int32)), "b1": pa.array(rng.integers(0, 100, nb).astype(np.int32)), "b2": pa.array(rng.integers(0, 100, nb).astype(np.int32)),
                      "b3": pa.array(rng.integers(0, 100, nb).astype(np.int32)), "b4": pa.array(rng.integers(0, 100, nb).astype(np.int32))})
    probe = pa.table({"pk": pa.array(rng.integers(0, nb + 500, n).astype(np.int32)), "p1": pa.array(rng.integers(0, 100, n).astype(np.int32)), "p2": pa.array(rng.integers(0, 100, n).astype(np.int32)),
                      "p3": pa.array(rng.integers(0, 100, n).astype(np.int32)), "p4": pa.array(rng.integers(0, 100, n).astype(np.int32)), "i": pa.array(np.arange(n, dtype=np.int64))})
    plan = {"steps": [{"op": "join_build", "in": "b", "keys": ["bk"], "unique": True, "index": False, "out": "h"},
                      {"op": "join_probe", "ht": "h", "in": "p", "keys": ["pk"], "kind": "inner",
                       "residual": [{"probe": "p1", "op": "LT", "build": "b1"}, {"probe": "p2", "op": "GTE", "build": "b2"}, {"probe": "p3", "op": "NEQ", "build": "b3"}, {"probe": "p4", "op": "LTE", "build": "b4"}],
                       "out": "j"},
                      {"op": "materialize", "in": "j", "cols": ["i", "bk"], "out": "result"}], "result": "result"}
    got = ctx.run_plan(json.dumps(plan), {"b": ctx.register("res_b", build), "p": ctx.register("res_p", probe)}).to_arrow()
    pk = probe.column("pk").to_numpy()
    ok = pk < nb
    bi = np.where(ok, pk, 0)
    col = lambda t, c: t.column(c).to_numpy()
    keep = ok & (col(probe, "p1") < col(build, "b1")[bi]) & (col(probe, "p2") >= col(build, "b2")[bi]) & (col(probe, "p3") != col(build, "b3")[bi]) & (col(probe, "p4") <= col(build, "b4")[bi])
    want = np.nonzero(keep)[0]
    assert 0 < len(want) < n // 4
    assert got.column(0).to_pylist() == want.tolist() and got.column(1).to_pylist() == pk[want].tolist()
