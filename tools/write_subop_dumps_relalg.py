#!/usr/bin/env python3
"""The TPC-H queries tools/write_subop_dumps.py does not author (Q2, Q7 – Q11, Q13 – Q17, Q19 – Q22) as relational-algebra trees over
tools/subop_lower.py, which prints them as sub-operator dumps in the schema of the reference's `tools/ct/mlir-subop-to-json.cpp`.
The trees are hand-ordered like the plan files in lingo-db_amd/plans/tpch (join orders; semi joins where only existence matters),
in relational algebra only: what those plans do with interpreter tricks arrives here in the reference's own shapes —
  scalar subqueries   → constant single joins (Q11, Q15, Q22) and a filter comparing two columns
  correlated subqueries → the decorrelated aggregation joined back on the correlation key (Q2, Q17, Q20)
  count(distinct)     → a distinct projection below the aggregation (Q16)
  the outer join      → anyTuple + null / as-nullable maps + union (Q13)
  EXISTS / NOT EXISTS with a residual → semi / anti joins with `reverseSides` and a second conjunct (Q21, Q22)
  extract(year …), substring, LIKE → db.runtime_call leaves (ExtractYearFromDate, Substring, ConstLike)
Result columns are the ones the oracle legs (oracle/tpch_legs.py) produce over the generated schema (tests/tpch_data.py).
Writes tests/golden/subop_tpch_q{2,7,8,9,10,11,13,14,15,16,17,19,20,21,22}.json."""
from subop_lower import (Aggregate, C, ConstJoin, Cx, Distinct, Join, Map, Rename, Select, Sort, Table, Tmp, TopK, add, and_, between, call, cast, const, date, dec, div, eq, gt, gte,
                         if_, lt, lte, mul, neq, not_, one_of, or_, result, sconst, sub)

ONE = const(1, "decimal(12,2)")


def volume(t): return mul(t["l_extendedprice"].j, sub(ONE, t["l_discount"].j))
def like(c, pattern): return call("ConstLike", c.j, sconst(pattern))
def year(c): return call("ExtractYearFromDate", c.j)


def q2():
    cx = Cx("tpch_q2")
    r, n, s, ps = Table("region", filters=[("r_name", "EQ", "EUROPE")]), Table("nation"), Table("supplier"), Table("partsupp")
    p = Select(Table("part", filters=[("p_size", "EQ", 15)]), like(Table("part")["p_type"], "%BRASS"))
    P = Table("part")
    n1 = Join("semi", n, r, [(n["n_regionkey"], r["r_regionkey"])])
    sn = Join("inner", s, n1, [(s["s_nationkey"], n["n_nationkey"])])
    psp = Join("inner", ps, p, [(ps["ps_partkey"], P["p_partkey"])])
    offers = Join("inner", psp, sn, [(ps["ps_suppkey"], s["s_suppkey"])])
    cols = [ps["ps_partkey"], ps["ps_supplycost"], s["s_acctbal"], s["s_name"], n["n_name"], P["p_partkey"], P["p_mfgr"], s["s_address"], s["s_phone"], s["s_comment"]]
    t = Tmp(offers, cols)
    min_cost, m_partkey = C("aggr0::min_cost", "decimal(12,2)"), C("min0::ps_partkey", "int32")
    m = Rename(Aggregate(t, [ps["ps_partkey"]], [("min", ps["ps_supplycost"], min_cost)]), [(m_partkey, ps["ps_partkey"])])
    best = Join("inner", t, m, [(ps["ps_partkey"], m_partkey)], residual=[eq(ps["ps_supplycost"].j, min_cost.j)])
    top = TopK(best, [(s["s_acctbal"], "desc"), (n["n_name"], "asc"), (s["s_name"], "asc"), (P["p_partkey"], "asc")], 100)
    return result(cx, top, [("s_acctbal", s["s_acctbal"]), ("s_name", s["s_name"]), ("n_name", n["n_name"]), ("p_partkey", P["p_partkey"]), ("p_mfgr", P["p_mfgr"]),
                            ("s_address", s["s_address"]), ("s_phone", s["s_phone"]), ("s_comment", s["s_comment"])])


def q7():
    cx = Cx("tpch_q7")
    n1, n2, s, c, o = Table("nation", "n1"), Table("nation", "n2"), Table("supplier"), Table("customer"), Table("orders")
    l = Table("lineitem", filters=[("l_shipdate", "GTE", "1995-01-01"), ("l_shipdate", "LTE", "1996-12-31")])
    sn = Join("inner", s, n1, [(s["s_nationkey"], n1["n_nationkey"])])
    cn = Join("inner", c, n2, [(c["c_nationkey"], n2["n_nationkey"])])
    ls = Join("inner", l, sn, [(l["l_suppkey"], s["s_suppkey"])])
    oc = Join("inner", o, cn, [(o["o_custkey"], c["c_custkey"])])
    j = Join("inner", ls, oc, [(l["l_orderkey"], o["o_orderkey"])])
    fr, ge = sconst("FRANCE"), sconst("GERMANY")
    sel = Select(j, or_(and_(eq(n1["n_name"].j, fr), eq(n2["n_name"].j, ge)), and_(eq(n1["n_name"].j, ge), eq(n2["n_name"].j, fr))))
    l_year, vol, revenue = C("map0::l_year", "int64"), C("map0::volume", "decimal(24,4)"), C("aggr0::revenue", "decimal(38,4)")
    mp = Map(sel, [(l_year, year(l["l_shipdate"])), (vol, volume(l))])
    g = Aggregate(mp, [n1["n_name"], n2["n_name"], l_year], [("sum", vol, revenue)])
    srt = Sort(g, [(n1["n_name"], "asc"), (n2["n_name"], "asc"), (l_year, "asc")])
    return result(cx, srt, [("supp_nation", n1["n_name"]), ("cust_nation", n2["n_name"]), ("l_year", l_year), ("revenue", revenue)])


def q8():
    cx = Cx("tpch_q8")
    p = Table("part", filters=[("p_type", "EQ", "ECONOMY ANODIZED STEEL")])
    r, n1, n2, c, s, l = Table("region", filters=[("r_name", "EQ", "AMERICA")]), Table("nation", "n1"), Table("nation", "n2"), Table("customer"), Table("supplier"), Table("lineitem")
    o = Table("orders", filters=[("o_orderdate", "GTE", "1995-01-01"), ("o_orderdate", "LTE", "1996-12-31")])
    lp = Join("semi", l, p, [(l["l_partkey"], p["p_partkey"])])
    ls = Join("inner", lp, s, [(l["l_suppkey"], s["s_suppkey"])])
    n1r = Join("semi", n1, r, [(n1["n_regionkey"], r["r_regionkey"])])
    cr = Join("semi", c, n1r, [(c["c_nationkey"], n1["n_nationkey"])])
    oc = Join("semi", o, cr, [(o["o_custkey"], c["c_custkey"])])
    j = Join("inner", ls, oc, [(l["l_orderkey"], o["o_orderkey"])])
    jn = Join("inner", j, n2, [(s["s_nationkey"], n2["n_nationkey"])])
    o_year, vol, bvol = C("map0::o_year", "int64"), C("map0::volume", "decimal(24,4)"), C("map0::brazil_volume", "decimal(24,4)")
    mp = Map(jn, [(o_year, year(o["o_orderdate"])), (vol, volume(l)), (bvol, if_(eq(n2["n_name"].j, sconst("BRAZIL")), volume(l), const(0, "decimal(24,4)")))])
    brazil, total, share = C("aggr0::brazil", "decimal(38,4)"), C("aggr0::total", "decimal(38,4)"), C("map1::mkt_share", "decimal(38,4)")
    g = Aggregate(mp, [o_year], [("sum", bvol, brazil), ("sum", vol, total)])
    srt = Sort(Map(g, [(share, div(brazil.j, total.j))]), [(o_year, "asc")])
    return result(cx, srt, [("o_year", o_year), ("mkt_share", share)])


def q9():
    cx = Cx("tpch_q9")
    P, l, ps, s, o, n = Table("part"), Table("lineitem"), Table("partsupp"), Table("supplier"), Table("orders"), Table("nation")
    green = Select(P, like(P["p_name"], "%green%"))
    lp = Join("semi", l, green, [(l["l_partkey"], P["p_partkey"])])
    psp = Join("semi", ps, green, [(ps["ps_partkey"], P["p_partkey"])])
    lps = Join("inner", lp, psp, [(l["l_partkey"], ps["ps_partkey"]), (l["l_suppkey"], ps["ps_suppkey"])])
    lpss = Join("inner", lps, s, [(l["l_suppkey"], s["s_suppkey"])])
    lo = Join("inner", lpss, o, [(l["l_orderkey"], o["o_orderkey"])])
    ln = Join("inner", lo, n, [(s["s_nationkey"], n["n_nationkey"])])
    o_year, amount, profit = C("map0::o_year", "int64"), C("map0::amount", "decimal(24,4)"), C("aggr0::sum_profit", "decimal(38,4)")
    mp = Map(ln, [(o_year, year(o["o_orderdate"])), (amount, sub(volume(l), mul(ps["ps_supplycost"].j, l["l_quantity"].j)))])
    g = Aggregate(mp, [n["n_name"], o_year], [("sum", amount, profit)])
    srt = Sort(g, [(n["n_name"], "asc"), (o_year, "desc")])
    return result(cx, srt, [("nation", n["n_name"]), ("o_year", o_year), ("sum_profit", profit)])


def q10():
    cx = Cx("tpch_q10")
    o = Table("orders", filters=[("o_orderdate", "GTE", "1993-10-01"), ("o_orderdate", "LT", "1994-01-01")])
    l, c, n = Table("lineitem", filters=[("l_returnflag", "EQ", "R")]), Table("customer"), Table("nation")
    lo = Join("inner", l, o, [(l["l_orderkey"], o["o_orderkey"])])
    loc = Join("inner", lo, c, [(o["o_custkey"], c["c_custkey"])])
    locn = Join("inner", loc, n, [(c["c_nationkey"], n["n_nationkey"])])
    vol, revenue = C("map0::volume", "decimal(24,4)"), C("aggr0::revenue", "decimal(38,4)")
    g = Aggregate(Map(locn, [(vol, volume(l))]), [c["c_custkey"], c["c_name"], c["c_acctbal"], n["n_name"]], [("sum", vol, revenue)])
    top = TopK(g, [(revenue, "desc")], 20)
    return result(cx, top, [("c_custkey", c["c_custkey"]), ("c_name", c["c_name"]), ("revenue", revenue), ("c_acctbal", c["c_acctbal"]), ("n_name", n["n_name"])])


def q11():
    cx = Cx("tpch_q11")
    n, s, ps = Table("nation", filters=[("n_name", "EQ", "GERMANY")]), Table("supplier"), Table("partsupp")
    s1 = Join("semi", s, n, [(s["s_nationkey"], n["n_nationkey"])])
    ps1 = Join("semi", ps, s1, [(ps["ps_suppkey"], s["s_suppkey"])])
    v = C("map0::stock", "decimal(24,2)")
    t = Tmp(Map(ps1, [(v, mul(ps["ps_supplycost"].j, cast(ps["ps_availqty"].j)))]), [ps["ps_partkey"], v])
    value, total, thr, thr_n = C("aggr0::value", "decimal(38,2)"), C("aggr1::total", "decimal(38,2)"), C("map1::threshold", "decimal(38,6)"), C("sj0::threshold", "decimal(38,6)")
    g = Aggregate(t, [ps["ps_partkey"]], [("sum", v, value)])
    tot = Map(Aggregate(t, [], [("sum", v, total)]), [(thr, mul(total.j, dec("0.0001", 5, 4)))])
    sel = Select(ConstJoin(g, tot, [(thr_n, thr)]), gt(value.j, thr_n.j))
    srt = Sort(sel, [(value, "desc")])
    return result(cx, srt, [("ps_partkey", ps["ps_partkey"]), ("value", value)])


def q13():
    cx = Cx("tpch_q13")
    c, o = Table("customer"), Table("orders")
    orders = Select(o, not_(like(o["o_comment"], "%special%requests%")))
    oj_orderkey = C("oj0::o_orderkey", "nullable(int32)")
    co = Join("outer", c, orders, [(c["c_custkey"], o["o_custkey"])], mapping=[(oj_orderkey, o["o_orderkey"])])
    c_count, custdist = C("aggr0::c_count", "int64"), C("aggr1::custdist", "int64")
    g1 = Aggregate(co, [c["c_custkey"]], [("count", oj_orderkey, c_count)], nullable_args=[oj_orderkey])
    g2 = Aggregate(g1, [c_count], [("count_star", None, custdist)])
    srt = Sort(g2, [(custdist, "desc"), (c_count, "desc")])
    return result(cx, srt, [("c_count", c_count), ("custdist", custdist)])


def q14():
    cx = Cx("tpch_q14")
    l, p = Table("lineitem", filters=[("l_shipdate", "GTE", "1995-09-01"), ("l_shipdate", "LT", "1995-10-01")]), Table("part")
    lp = Join("inner", l, p, [(l["l_partkey"], p["p_partkey"])])
    vol, pvol = C("map0::volume", "decimal(24,4)"), C("map0::promo_volume", "decimal(24,4)")
    mp = Map(lp, [(vol, volume(l)), (pvol, if_(like(p["p_type"], "PROMO%"), volume(l), const(0, "decimal(24,4)")))])
    a, b, out = C("aggr0::promo", "decimal(38,4)"), C("aggr0::total", "decimal(38,4)"), C("map1::promo_revenue", "decimal(38,6)")
    g = Aggregate(mp, [], [("sum", pvol, a), ("sum", vol, b)])
    return result(cx, Map(g, [(out, div(mul(dec("100.00", 5, 2), a.j), b.j))]), [("promo_revenue", out)])


def q15():
    cx = Cx("tpch_q15")
    l, s = Table("lineitem", filters=[("l_shipdate", "GTE", "1996-01-01"), ("l_shipdate", "LT", "1996-04-01")]), Table("supplier")
    vol, total, best, best_n = C("map0::volume", "decimal(24,4)"), C("aggr0::total_revenue", "decimal(38,4)"), C("aggr1::best", "decimal(38,4)"), C("sj0::best", "decimal(38,4)")
    rev = Tmp(Aggregate(Map(l, [(vol, volume(l))]), [l["l_suppkey"]], [("sum", vol, total)]), [l["l_suppkey"], total])
    mx = Aggregate(rev, [], [("max", total, best)])
    winners = Select(ConstJoin(rev, mx, [(best_n, best)]), eq(total.j, best_n.j))
    j = Join("inner", s, winners, [(s["s_suppkey"], l["l_suppkey"])])
    srt = Sort(j, [(s["s_suppkey"], "asc")])
    return result(cx, srt, [("s_suppkey", s["s_suppkey"]), ("total_revenue", total)])


def q16():
    cx = Cx("tpch_q16")
    s, ps, P = Table("supplier"), Table("partsupp"), Table("part")
    complaints = Select(s, like(s["s_comment"], "%Customer%Complaints%"))
    p = Select(Table("part", filters=[("p_brand", "NEQ", "Brand#45"), ("p_size", "IN", [49, 14, 23, 45, 19, 3, 36, 9])]), not_(like(P["p_type"], "MEDIUM POLISHED%")))
    pp = Join("inner", ps, p, [(ps["ps_partkey"], P["p_partkey"])])
    pp2 = Join("anti", pp, complaints, [(ps["ps_suppkey"], s["s_suppkey"])])
    d = Distinct(pp2, [P["p_brand"], P["p_type"], P["p_size"], ps["ps_suppkey"]])
    cnt = C("aggr0::supplier_cnt", "int64")
    g = Aggregate(d, [P["p_brand"], P["p_type"], P["p_size"]], [("count_star", None, cnt)])
    srt = Sort(g, [(cnt, "desc"), (P["p_brand"], "asc"), (P["p_type"], "asc"), (P["p_size"], "asc")])
    return result(cx, srt, [("p_brand", P["p_brand"]), ("p_type", P["p_type"]), ("p_size", P["p_size"]), ("supplier_cnt", cnt)])


def q17():
    cx = Cx("tpch_q17")
    p, l = Table("part", filters=[("p_brand", "EQ", "Brand#23"), ("p_container", "EQ", "MED BOX")]), Table("lineitem")
    t = Tmp(Join("semi", l, p, [(l["l_partkey"], p["p_partkey"])]), [l["l_partkey"], l["l_quantity"], l["l_extendedprice"]])
    sq, cq, avg_qty, a_partkey = C("aggr0::sum_qty", "decimal(38,2)"), C("aggr0::cnt", "int64"), C("map0::avg_qty", "decimal(38,8)"), C("avg0::l_partkey", "int32")
    a = Rename(Map(Aggregate(t, [l["l_partkey"]], [("sum", l["l_quantity"], sq), ("count_star", None, cq)]), [(avg_qty, div(sq.j, cast(cq.j)))]), [(a_partkey, l["l_partkey"])])
    la = Join("inner", t, a, [(l["l_partkey"], a_partkey)])
    small = Select(la, lt(l["l_quantity"].j, mul(dec("0.2", 2, 1), avg_qty.j)))
    s_, out = C("aggr1::sum_price", "decimal(38,2)"), C("map1::avg_yearly", "decimal(38,8)")
    g = Aggregate(small, [], [("sum", l["l_extendedprice"], s_)])
    return result(cx, Map(g, [(out, div(s_.j, dec("7.0", 2, 1)))]), [("avg_yearly", out)])


def q19():
    cx = Cx("tpch_q19")
    l, p = Table("lineitem", filters=[("l_shipinstruct", "EQ", "DELIVER IN PERSON"), ("l_shipmode", "IN", ["AIR", "AIR REG"])]), Table("part")
    qty = lambda lo, hi: [gte(l["l_quantity"].j, const(lo, "decimal(12,2)")), lte(l["l_quantity"].j, const(hi, "decimal(12,2)"))]
    alt = lambda brand, boxes, lo, hi, size: and_(eq(p["p_brand"].j, sconst(brand)), one_of(p["p_container"].j, [sconst(b) for b in boxes]), *qty(lo, hi),
                                                  between(p["p_size"].j, const(1, "int32"), const(size, "int32")))
    alts = or_(alt("Brand#12", ["SM CASE", "SM BOX", "SM PACK", "SM PKG"], 1, 11, 5), alt("Brand#23", ["MED BAG", "MED BOX", "MED PKG", "MED PACK"], 10, 20, 10),
               alt("Brand#34", ["LG CASE", "LG BOX", "LG PACK", "LG PKG"], 20, 30, 15))
    j = Join("inner", l, p, [(l["l_partkey"], p["p_partkey"])], residual=[alts])
    vol, revenue = C("map0::volume", "decimal(24,4)"), C("aggr0::revenue", "decimal(38,4)")
    g = Aggregate(Map(j, [(vol, volume(l))]), [], [("sum", vol, revenue)])
    return result(cx, g, [("revenue", revenue)])


def q20():
    cx = Cx("tpch_q20")
    P, ps, s, n = Table("part"), Table("partsupp"), Table("supplier"), Table("nation", filters=[("n_name", "EQ", "CANADA")])
    l = Table("lineitem", filters=[("l_shipdate", "GTE", "1994-01-01"), ("l_shipdate", "LT", "1995-01-01")])
    forest = Select(P, like(P["p_name"], "forest%"))
    ps1 = Join("semi", ps, forest, [(ps["ps_partkey"], P["p_partkey"])])
    l1 = Join("semi", l, forest, [(l["l_partkey"], P["p_partkey"])])
    sum_qty, q_partkey, q_suppkey = C("aggr0::sum_qty", "decimal(38,2)"), C("q0::l_partkey", "int32"), C("q0::l_suppkey", "int32")
    lq = Rename(Aggregate(l1, [l["l_partkey"], l["l_suppkey"]], [("sum", l["l_quantity"], sum_qty)]), [(q_partkey, l["l_partkey"]), (q_suppkey, l["l_suppkey"])])
    pq = Join("inner", ps1, lq, [(ps["ps_partkey"], q_partkey), (ps["ps_suppkey"], q_suppkey)])
    enough = Select(pq, gt(cast(ps["ps_availqty"].j), mul(dec("0.5", 2, 1), sum_qty.j)))
    s1 = Join("semi", s, n, [(s["s_nationkey"], n["n_nationkey"])])
    s2 = Join("semi", s1, enough, [(s["s_suppkey"], ps["ps_suppkey"])])
    srt = Sort(s2, [(s["s_name"], "asc")])
    return result(cx, srt, [("s_name", s["s_name"]), ("s_address", s["s_address"])])


def q21():
    cx = Cx("tpch_q21")
    n, s, o = Table("nation", filters=[("n_name", "EQ", "SAUDI ARABIA")]), Table("supplier"), Table("orders", filters=[("o_orderstatus", "EQ", "F")])
    l1, l2, l3 = Table("lineitem", "l1"), Table("lineitem", "l2"), Table("lineitem", "l3")
    late = lambda t: Select(t, gt(t["l_receiptdate"].j, t["l_commitdate"].j))
    s1 = Join("semi", s, n, [(s["s_nationkey"], n["n_nationkey"])])
    l1s = Join("inner", late(l1), s1, [(l1["l_suppkey"], s["s_suppkey"])])
    l1o = Join("semi", l1s, o, [(l1["l_orderkey"], o["o_orderkey"])])
    # EXISTS (another supplier's line of the order) / NOT EXISTS (another supplier's LATE line): the candidate lines are the flagged build side
    ex = Join("semi", l2, l1o, [(l2["l_orderkey"], l1["l_orderkey"])], residual=[neq(l2["l_suppkey"].j, l1["l_suppkey"].j)], reverse=True)
    nex = Join("anti", late(l3), ex, [(l3["l_orderkey"], l1["l_orderkey"])], residual=[neq(l3["l_suppkey"].j, l1["l_suppkey"].j)], reverse=True)
    numwait = C("aggr0::numwait", "int64")
    g = Aggregate(nex, [s["s_name"]], [("count_star", None, numwait)])
    top = TopK(g, [(numwait, "desc"), (s["s_name"], "asc")], 100)
    return result(cx, top, [("s_name", s["s_name"]), ("numwait", numwait)])


def q22():
    cx = Cx("tpch_q22")
    c, o = Table("customer"), Table("orders")
    code = C("map0::cntrycode", "str")
    codes = Select(Map(c, [(code, call("Substring", c["c_phone"].j, const(1, "int32"), const(2, "int32")))]), one_of(code.j, [sconst(x) for x in ("13", "31", "23", "29", "30", "18", "17")]))
    t = Tmp(codes, [c["c_custkey"], c["c_acctbal"], code])
    sa, ca, avg, avg_n = C("aggr0::sum_bal", "decimal(38,2)"), C("aggr0::cnt", "int64"), C("map1::avg_bal", "decimal(38,8)"), C("sj0::avg_bal", "decimal(38,8)")
    positive = Select(t, gt(c["c_acctbal"].j, dec("0.00", 3, 2)))
    av = Map(Aggregate(positive, [], [("sum", c["c_acctbal"], sa), ("count_star", None, ca)]), [(avg, div(sa.j, cast(ca.j)))])
    rich = Select(ConstJoin(t, av, [(avg_n, avg)]), gt(c["c_acctbal"].j, avg_n.j))
    no_orders = Join("anti", o, rich, [(o["o_custkey"], c["c_custkey"])], reverse=True)  # NOT EXISTS (orders): the few customers are the flagged build side
    numcust, tot = C("aggr1::numcust", "int64"), C("aggr1::totacctbal", "decimal(38,2)")
    g = Aggregate(no_orders, [code], [("count_star", None, numcust), ("sum", c["c_acctbal"], tot)])
    srt = Sort(g, [(code, "asc")])
    return result(cx, srt, [("cntrycode", code), ("numcust", numcust), ("totacctbal", tot)])


ALL = {2: q2, 7: q7, 8: q8, 9: q9, 10: q10, 11: q11, 13: q13, 14: q14, 15: q15, 16: q16, 17: q17, 19: q19, 20: q20, 21: q21, 22: q22}

if __name__ == "__main__":
    for q, f in ALL.items():
        print(f())
