"""The oracle legs bench.py times as `cpu_baseline` for Q10 and Q15 (tpch_plans.cpu_baseline: the
reference CPU path restated over the oracle's operators) return what the SQL text says — checked
against a plain-Python evaluation of resources/sql/tpch/{10,15}.sql over the same generated tables."""
import collections

import tpch_data
import tpch_plans
from test_gpu_tpch_more import days, np_col


def test_oracle_q10_q15_legs(monkeypatch):
    monkeypatch.setenv("LDB_CPU_BASELINE_RESULTS", "1")
    sf = 0.01
    n = int(round(sf * 1_500_000))
    got = tpch_plans.cpu_baseline([10, 15], sf)
    T = tpch_data
    li = T.host_table(T.LINEITEM, n, cols=[0, 2, 5, 6, 8, 10])
    od = T.host_table(T.ORDERS, n, cols=[0, 1, 4])
    ocust = {ok: ck for ok, ck, d in zip(np_col(od, "o_orderkey").tolist(), np_col(od, "o_custkey").tolist(), np_col(od, "o_orderdate").tolist())
             if days("1993-10-01") <= d < days("1994-01-01")}
    flags = [v.as_py()[:1] for v in li.column("l_returnflag").combine_chunks()]
    q10, q15 = collections.defaultdict(int), collections.defaultdict(int)
    for ok, sk, ext, disc, fl, d in zip(*[np_col(li, c).tolist() for c in ("l_orderkey", "l_suppkey", "l_extendedprice", "l_discount")], flags, np_col(li, "l_shipdate").tolist()):
        if fl == b"R" and ok in ocust:
            q10[ocust[ok]] += ext * (100 - disc)
        if days("1996-01-01") <= d < days("1996-04-01"):
            q15[sk] += ext * (100 - disc)
    assert len(q10) > 100 and len(q15) > 50
    # the legs return result ROWS now: Q10 = the best customers (c_custkey, c_name, revenue, c_acctbal, n_name), Q15 = the best suppliers
    ranked = sorted(q10.items(), key=lambda r: -r[1])
    assert [r[2] for r in got[10][:20]] == [r[1] for r in ranked[:20]] and all(q10[r[0]] == r[2] for r in got[10])
    best = max(q15.values())
    assert got[15] == sorted((k, v) for k, v in q15.items() if v == best)
