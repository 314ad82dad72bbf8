"""bench.py's cpu_baseline (kind "reference", round 6): TPC-H Q1 / Q6 / Q3 as compiled morsel loops around the reference's REAL runtime objects
(oracle/_ref: Restrictions, PreAggregationHashtable, GrowingBuffer, HashIndexedView — oracle/ref_build/ref_glue.cpp) return what the oracle legs
return, on 1 and on several threads.  Needs oracle/_ref (built from /root/reference by __graft_entry__.build())."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle")]
import ref_baseline  # noqa: E402
import tpch_data as T  # noqa: E402
import tpch_legs  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_baseline.available(), reason="oracle/_ref/libldb_ref.so not built (no /root/reference here)")


@pytest.mark.parametrize("threads", [1, 4])
def test_reference_object_legs_equal_the_oracle_legs(threads):
    n_orders = 60_000
    legs = tpch_legs.Legs(n_orders, queries=[1, 3, 6])
    li = ref_baseline.columns_from_arrow(T.host_table(T.LINEITEM, n_orders, cols=[0, 4, 5, 6, 7, 8, 9, 10]),
                                         ["l_orderkey", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"])
    od = ref_baseline.columns_from_arrow(T.host_table(T.ORDERS, n_orders, cols=[0, 1, 4, 6]), ["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
    cu_t = T.host_table(T.CUSTOMER, n_orders, cols=[0, 3])
    cu = ref_baseline.columns_from_arrow(cu_t, ["c_custkey"])
    cu["c_segment4"] = ref_baseline.segment4(cu_t.column("c_mktsegment"))
    with ref_baseline.Session(threads) as s:
        ms1, partials = s.q1(li)
        assert ms1 > 0 and tpch_legs.Legs.q1_finish(partials) == legs.q1()
        ms6, v6 = s.q6(li)
        assert ms6 > 0 and [(v6,)] == legs.q6()
        ms3, rows3 = s.q3(cu, od, li)
        want3 = legs.q3()
        assert ms3 > 0 and len(rows3) == len(want3) > 100
        assert [(r[1], r[2]) for r in rows3] == [(r[1], r[2]) for r in want3] and sorted(rows3) == sorted(want3)
