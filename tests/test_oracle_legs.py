"""The oracle legs (oracle/tpch_legs.py: TPC-H over the C restatement of the reference's CPU path —
the checker of tests/test_gpu_sf1_oracle.py and bench.py's cpu_baseline) return what the SQL text
says: the eight queries with a pandas evaluation (tests/tpch_sql.py) are compared with it, the
others with invariants of the generated data and with each other."""
import numpy as np
import pytest

import tpch_legs
import tpch_sql

N = 30_000  # SF 0.02


@pytest.fixture(scope="module")
def legs():
    return tpch_legs.Legs(N, threads=4)


@pytest.fixture(scope="module")
def host():
    return tpch_sql.Tables(N)


@pytest.mark.parametrize("q", sorted(tpch_sql.SQL))
def test_leg_matches_sql_text(legs, host, q):
    assert legs.run(q) == tpch_sql.evaluate(q, host)


def test_q1_q6_invariants(legs):
    q1 = legs.run(1)
    assert [r[:2] for r in q1] == sorted(r[:2] for r in q1) and 3 <= len(q1) <= 6
    li = legs.frame(0)
    assert sum(r[9] for r in q1) == int((li.np("l_shipdate") <= tpch_legs.days("1998-09-02")).sum())
    for r in q1:
        assert r[6] == (r[2] * 10**19) // r[9]
    (q6,) = legs.run(6)[0]
    m = (li.np("l_shipdate") >= tpch_legs.days("1994-01-01")) & (li.np("l_shipdate") < tpch_legs.days("1995-01-01")) & (li.np("l_discount") >= 5) & (li.np("l_discount") <= 7) & \
        (li.np("l_quantity") < 2400)
    assert q6 == int((li.np("l_extendedprice")[m] * li.np("l_discount")[m]).sum())


def test_join_queries_against_numpy(legs):
    """Q3 / Q12 / Q14 / Q18 recomputed with numpy set operations over the same host columns"""
    li, od, cu = legs.frame(0), legs.frame(1), legs.frame(2)
    d = tpch_legs.days
    # Q3
    seg = cu.strs("c_mktsegment") == "BUILDING"
    okc = np.isin(od.np("o_custkey"), cu.np("c_custkey")[seg]) & (od.np("o_orderdate") < d("1995-03-15"))
    sel = (li.np("l_shipdate") > d("1995-03-15")) & np.isin(li.np("l_orderkey"), od.np("o_orderkey")[okc])
    rev = {}
    for k, v in zip(li.np("l_orderkey")[sel].tolist(), (li.np("l_extendedprice")[sel] * (100 - li.np("l_discount")[sel])).tolist()):
        rev[k] = rev.get(k, 0) + v
    q3 = legs.run(3)
    assert {r[0]: r[1] for r in q3} == rev and [(-r[1], r[2]) for r in q3] == sorted((-r[1], r[2]) for r in q3)
    # Q14
    pt = legs.frame(3)
    promo = pt.np("p_partkey")[np.array([t.startswith("PROMO") for t in pt.strs("p_type")])]
    m = (li.np("l_shipdate") >= d("1995-09-01")) & (li.np("l_shipdate") < d("1995-10-01"))
    r = li.np("l_extendedprice")[m] * (100 - li.np("l_discount")[m])
    a, b = int(r[np.isin(li.np("l_partkey")[m], promo)].sum()), int(r.sum())
    assert legs.run(14) == [((a * 10000) * 10**4 // b,)]
    # Q18: every reported order sums to more than 300 units and the list is ordered by (o_totalprice desc, o_orderdate)
    q18 = legs.run(18)
    qty = {}
    for k, v in zip(li.np("l_orderkey").tolist(), li.np("l_quantity").tolist()):
        qty[k] = qty.get(k, 0) + v
    assert {r[2]: r[5] for r in q18} == {k: v for k, v in qty.items() if v > 30000}
    assert [(-r[4], r[3]) for r in q18] == sorted((-r[4], r[3]) for r in q18)


def test_remaining_legs_run_and_are_ordered(legs):
    for q, key in ((4, lambda r: r[0]), (5, lambda r: -r[1]), (7, lambda r: r[:3]), (8, lambda r: r[0]), (9, lambda r: (r[0], -r[1])), (11, lambda r: -r[1]), (12, lambda r: r[0])):
        rows = legs.run(q)
        assert rows and [key(r) for r in rows] == sorted(key(r) for r in rows), q
    assert all(len(r) == 5 for r in legs.run(10)) and legs.run(15)
