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
