#!/usr/bin/env python3
"""Writes lingo-db_amd/plans/tpch/q{1,3,4,5,6,7,8,9,10,11,12,14,15,18}.json — the single-GPU plans
that were C++ functions in round 1 (lingo-db_amd/host/ldb_host.cpp), now data for the plan
interpreter.  The files are the source of truth once written; this script only keeps them
consistently formatted (one step per line)."""
import json
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lingo-db_amd", "plans", "tpch")

REV = {"mul": ["l_extendedprice", {"sub": [1, "l_discount"]}]}


def f(col, op, value=None, **kw):
    d = {"col": col, "op": op}
    if value is not None:
        d["value"] = value
    d.update(kw)
    return d


def scan_filter(src, out, preds):
    return {"op": "filter", "in": src, "out": out, "preds": preds}


def build(src, keys, out, unique=True):
    return {"op": "join_build", "in": src, "keys": keys, "unique": unique, "out": out}


def probe(ht, src, keys, out, kind="inner"):
    return {"op": "join_probe", "ht": ht, "in": src, "keys": keys, "kind": kind, "out": out}


def groupby(src, keys, aggs, out, est=None, **kw):
    d = {"op": "groupby", "in": src, "keys": keys, "aggs": aggs}
    if est is not None:
        d["est_groups"] = est
    d.update(kw)
    d["out"] = out
    return d


def agg(fn, expr=None, name=None, **kw):
    d = {"fn": fn}
    if expr is not None:
        d["expr"] = expr
    d.update(kw)
    d["as"] = name
    return d


def sort(src, by, out):
    return {"op": "sort", "in": src, "by": by, "out": out}


def topk(src, by, k, out):
    return {"op": "topk", "in": src, "by": by, "k": k, "out": out}


def mat(src, cols, out="result"):
    return {"op": "materialize", "in": src, "cols": cols, "out": out}


def desc(col):
    return {"col": col, "desc": True}


def members(dim, key, nation_col, prefix, nation_filter, region=None):
    """rows of `dim` whose nation passes the filter → table (key, nation_col); the small reduced
    dimension table is what a multi-GPU run all-gathers"""
    steps = []
    nat = "nation"
    if region:
        steps += [scan_filter("region", prefix + "_r", [f("r_name", "EQ", region)]), build(prefix + "_r", ["r_regionkey"], prefix + "_hr"),
                  probe(prefix + "_hr", "nation", ["n_regionkey"], prefix + "_n", "semi")]
        nat = prefix + "_n"
    elif nation_filter:
        steps += [scan_filter("nation", prefix + "_n", nation_filter)]
        nat = prefix + "_n"
    steps += [build(nat, ["n_nationkey"], prefix + "_hn"), probe(prefix + "_hn", dim, [nation_col], prefix + "_sel", "semi"), mat(prefix + "_sel", [key, nation_col], prefix)]
    return steps


PLANS = {}

PLANS[1] = dict(ref="resources/sql/tpch/1.sql", inputs=["lineitem"],
                doc="the pushed-down l_shipdate restriction is fused into the aggregation kernel (scan → filter → aggregate in one pass); date '1998-12-01' - interval '90' day is folded by the frontend",
                steps=[
                    groupby("lineitem", ["l_returnflag", "l_linestatus"],
                            [agg("sum", "l_quantity", "sum_qty"), agg("sum", "l_extendedprice", "sum_base_price"), agg("sum", REV, "sum_disc_price"),
                             agg("sum", {"mul": ["l_extendedprice", {"sub": [1, "l_discount"]}, {"add": [1, "l_tax"]}]}, "sum_charge"), agg("avg", "l_quantity", "avg_qty"),
                             agg("avg", "l_extendedprice", "avg_price"), agg("avg", "l_discount", "avg_disc"), agg("count_star", None, "count_order")],
                            "g", est=6, preds=[f("l_shipdate", "LTE", "1998-09-02")]),
                    sort("g", ["l_returnflag", "l_linestatus"], "gs"),
                    mat("gs", ["l_returnflag", "l_linestatus", "sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc", "count_order"])])

PLANS[6] = dict(ref="resources/sql/tpch/6.sql", inputs=["lineitem"], result="revenue",
                doc="pure scan + key-less SUM (SimpleState); 0.06 - 0.01 / 0.06 + 0.01 are folded to decimal constants, BETWEEN is two inclusive restrictions",
                steps=[groupby("lineitem", [], [agg("sum", {"mul": ["l_extendedprice", "l_discount"]}, "revenue")], "revenue", est=1,
                               preds=[f("l_shipdate", "GTE", "1994-01-01"), f("l_shipdate", "LT", "1995-01-01"), f("l_discount", "GTE", "0.05"), f("l_discount", "LTE", "0.07"),
                                      f("l_quantity", "LT", 24)])])

PLANS[3] = dict(ref="resources/sql/tpch/3.sql", inputs=["customer", "orders", "lineitem"],
                doc="customer ⋈ orders ⋈ lineitem, GROUP BY three keys, top 10; the filtered customers (primary key) are the first hash table, the joined orders the second",
                steps=[scan_filter("customer", "c1", [f("c_mktsegment", "EQ", "BUILDING")]), scan_filter("orders", "o1", [f("o_orderdate", "LT", "1995-03-15")]),
                       scan_filter("lineitem", "l1", [f("l_shipdate", "GT", "1995-03-15")]), build("c1", ["c_custkey"], "hc"), probe("hc", "o1", ["o_custkey"], "co"),
                       build("co", ["o_orderkey"], "ho"), probe("ho", "l1", ["l_orderkey"], "lco"),
                       groupby("lco", ["l_orderkey", "o_orderdate", "o_shippriority"], [agg("sum", REV, "revenue")], "g", est_groups_from="rows"),
                       topk("g", [desc("revenue"), "o_orderdate"], 10, "top"), mat("top", ["l_orderkey", "revenue", "o_orderdate", "o_shippriority"])])

PLANS[4] = dict(ref="resources/sql/tpch/4.sql", inputs=["orders", "lineitem"],
                doc="EXISTS → semi join that keeps the hash-table side (the quarter's orders); l_commitdate < l_receiptdate is a residual column-vs-column conjunct",
                steps=[scan_filter("orders", "o1", [f("o_orderdate", "GTE", "1993-07-01"), f("o_orderdate", "LT", "1993-10-01")]),
                       scan_filter("lineitem", "l1", [f("l_commitdate", "LT", rhs_col="l_receiptdate")]), build("o1", ["o_orderkey"], "ho"),
                       probe("ho", "l1", ["l_orderkey"], "osel", "semi_build"), groupby("osel", ["o_orderpriority"], [agg("count_star", None, "order_count")], "g", est=5),
                       sort("g", ["o_orderpriority"], "gs"), mat("gs", ["o_orderpriority", "order_count"])])

PLANS[12] = dict(ref="resources/sql/tpch/12.sql", inputs=["orders", "lineitem"],
                 doc="the few late lineitems of two ship modes (0.5 %) probe the ORDERS primary-key index with their own l_orderkey, as Q9's lines do (round 6: they were a non-unique hash-table side that all 150 M orders probed — pair counting, pairs and the expansion were 2 of the query's 5 ms); the two CASE sums are conditional aggregates (integer literals are int32, SUM keeps the type)",
                 steps=[scan_filter("lineitem", "l1", [f("l_shipmode", "IN", values=["MAIL", "SHIP"]), f("l_receiptdate", "GTE", "1994-01-01"), f("l_receiptdate", "LT", "1995-01-01"),
                                                       f("l_commitdate", "LT", rhs_col="l_receiptdate"), f("l_shipdate", "LT", rhs_col="l_commitdate")]),
                        build("orders", ["o_orderkey"], "ho"), probe("ho", "l1", ["l_orderkey"], "ol"),
                        groupby("ol", ["l_shipmode"],
                                [agg("sum", 1, "high_line_count", when=[f("o_orderpriority", "IN", values=["1-URGENT", "2-HIGH"])], type="int32"),
                                 agg("sum", 1, "low_line_count", when=[f("o_orderpriority", "NEQ", "1-URGENT"), f("o_orderpriority", "NEQ", "2-HIGH")], type="int32")], "g", est=2),
                        sort("g", ["l_shipmode"], "gs"), mat("gs", ["l_shipmode", "high_line_count", "low_line_count"])])

PLANS[18] = dict(ref="resources/sql/tpch/18.sql", inputs=["customer", "orders", "lineitem"],
                 doc="the IN subquery is a group-by with one group per order (1.5 M x SF groups) + HAVING; its few keys are a semi-join hash table; the outer GROUP BY runs over the lineitems of those orders",
                 steps=[groupby("lineitem", ["l_orderkey"], [agg("sum", "l_quantity", "sum_qty")], "per_order", est={"rows_of": "orders"}),
                        scan_filter("per_order", "big", [f("sum_qty", "GT", 300)]), build("big", ["l_orderkey"], "hk"), probe("hk", "orders", ["o_orderkey"], "o1", "semi"),
                        build("o1", ["o_custkey"], "ho", unique=False), probe("ho", "customer", ["c_custkey"], "co"), build("co", ["o_orderkey"], "hco"),
                        probe("hco", "lineitem", ["l_orderkey"], "lco"),
                        groupby("lco", ["c_name", "c_custkey", "o_orderkey", "o_orderdate", "o_totalprice"], [agg("sum", "l_quantity", "sum_quantity")], "g", est={"rows_of": "co"}),
                        topk("g", [desc("o_totalprice"), "o_orderdate"], 100, "top"), mat("top", ["c_name", "c_custkey", "o_orderkey", "o_orderdate", "o_totalprice", "sum_quantity"])])

PLANS[9] = dict(ref="resources/sql/tpch/9.sql", inputs=["part", "supplier", "lineitem", "partsupp", "orders", "nation"],
                doc="join order by cardinality: the LIKE keeps 5.4 % of part and reduces lineitem and partsupp first; eager aggregation: the joined rows are summed per (s_nationkey, o_year) — integer keys — and only the <= 175 partial rows meet nation, the final GROUP BY (n_name, o_year) re-aggregates them",
                steps=[scan_filter("part", "p1", [f("p_name", "LIKE", "%green%")]), build("p1", ["p_partkey"], "hp"), probe("hp", "lineitem", ["l_partkey"], "lp", "semi"),
                       probe("hp", "partsupp", ["ps_partkey"], "ps1", "semi"), build("ps1", ["ps_partkey", "ps_suppkey"], "hps"), probe("hps", "lp", ["l_partkey", "l_suppkey"], "lps"),
                       build("supplier", ["s_suppkey"], "hs"), probe("hs", "lps", ["l_suppkey"], "lpss"),
                       mat("lpss", ["l_orderkey", "l_extendedprice", "l_discount", "l_quantity", "ps_supplycost", "s_nationkey"], "m"), build("m", ["l_orderkey"], "hm", unique=False),
                       probe("hm", "orders", ["o_orderkey"], "om"), {"op": "map", "in": "om", "fn": "extract_year", "col": "o_orderdate", "as": "o_year", "out": "omy"},
                       groupby("omy", ["s_nationkey", "o_year"], [agg("sum", {"sub": [REV, {"mul": ["ps_supplycost", "l_quantity"]}]}, "amount")], "partial", est=200),
                       build("nation", ["n_nationkey"], "hn"), probe("hn", "partial", ["s_nationkey"], "pn"),
                       groupby("pn", ["n_name", "o_year"], [agg("sum", "amount", "sum_profit")], "g", est=200), sort("g", ["n_name", desc("o_year")], "gs"),
                       mat("gs", ["n_name", "o_year", "sum_profit"])])

PLANS[5] = dict(ref="resources/sql/tpch/5.sql", inputs=["customer", "orders", "lineitem", "supplier", "nation", "region"],
                doc="customers and suppliers of the region's nations are reduced to (key, nationkey) tables first; c_nationkey = s_nationkey makes the supplier join a two-column semi join; SUM per nationkey, names joined afterwards",
                steps=members("customer", "c_custkey", "c_nationkey", "custs", None, "ASIA") + members("supplier", "s_suppkey", "s_nationkey", "supps", None, "ASIA") + [
                    scan_filter("orders", "o1", [f("o_orderdate", "GTE", "1994-01-01"), f("o_orderdate", "LT", "1995-01-01")]), build("custs", ["c_custkey"], "hc"),
                    probe("hc", "o1", ["o_custkey"], "oc"), build("oc", ["o_orderkey"], "ho"), probe("ho", "lineitem", ["l_orderkey"], "loc"),
                    build("supps", ["s_suppkey", "s_nationkey"], "hs"), probe("hs", "loc", ["l_suppkey", "c_nationkey"], "locs", "semi"),
                    groupby("locs", ["c_nationkey"], [agg("sum", REV, "revenue")], "partial", est=25), build("nation", ["n_nationkey"], "hn"),
                    probe("hn", "partial", ["c_nationkey"], "pn"), groupby("pn", ["n_name"], [agg("sum", "revenue", "revenue")], "g", est=25), sort("g", [desc("revenue")], "gs"),
                    mat("gs", ["n_name", "revenue"])])

PLANS[7] = dict(ref="resources/sql/tpch/7.sql", inputs=["customer", "orders", "lineitem", "supplier", "nation"],
                doc="(n1 = A and n2 = B) or (n1 = B and n2 = A) = both nations in {A, B} (pushed into the two dimension tables) and n1 <> n2 (a residual column-vs-column conjunct); the orders of the two nations' customers (8 %) are the UNIQUE build side the reduced lineitem side probes (round 6: 150 M orders probing a non-unique table of the lineitem side cost 3.6 ms of pair counting, pairs and build, and the lineitem side was materialised for it)",
                steps=members("customer", "c_custkey", "c_nationkey", "custs", [f("n_name", "IN", values=["FRANCE", "GERMANY"])]) +
                members("supplier", "s_suppkey", "s_nationkey", "supps", [f("n_name", "IN", values=["FRANCE", "GERMANY"])]) + [
                    scan_filter("lineitem", "l1", [f("l_shipdate", "GTE", "1995-01-01"), f("l_shipdate", "LTE", "1996-12-31")]), build("supps", ["s_suppkey"], "hs"),
                    probe("hs", "l1", ["l_suppkey"], "ls"), build("custs", ["c_custkey"], "hc"), probe("hc", "orders", ["o_custkey"], "oc"),
                    build("oc", ["o_orderkey"], "ho"), probe("ho", "ls", ["l_orderkey"], "omc"),
                    scan_filter("omc", "diff", [f("s_nationkey", "NEQ", rhs_col="c_nationkey")]),
                    {"op": "map", "in": "diff", "fn": "extract_year", "col": "l_shipdate", "as": "l_year", "out": "dy"},
                    groupby("dy", ["s_nationkey", "c_nationkey", "l_year"], [agg("sum", REV, "volume")], "partial", est=16), build("nation", ["n_nationkey"], "hn"),
                    probe("hn", "partial", ["s_nationkey"], "p1"), probe("hn", "p1", ["c_nationkey"], "p2"),
                    groupby("p2", ["1:n_name", "2:n_name", "l_year"], [agg("sum", "volume", "revenue")], "g", est=16, key_names=["supp_nation", "cust_nation", "l_year"]),
                    sort("g", ["supp_nation", "cust_nation", "l_year"], "gs"), mat("gs", ["supp_nation", "cust_nation", "l_year", "revenue"])])

PLANS[8] = dict(ref="resources/sql/tpch/8.sql", inputs=["part", "supplier", "lineitem", "orders", "customer", "nation", "region"],
                doc="the part-type filter keeps 1/150 of part and reduces lineitem first; two years of orders, semi-joined with the customers of the region, are the UNIQUE build side the reduced lineitem side probes (round 6, as in Q7); the CASE is a conditional SUM on the supplier's nation name",
                steps=[scan_filter("part", "p1", [f("p_type", "EQ", "ECONOMY ANODIZED STEEL")]), mat("p1", ["p_partkey"], "parts")] + members("customer", "c_custkey", "c_nationkey", "custs", None, "AMERICA") + [
                    build("parts", ["p_partkey"], "hp"), probe("hp", "lineitem", ["l_partkey"], "lp", "semi"), build("supplier", ["s_suppkey"], "hs"), probe("hs", "lp", ["l_suppkey"], "ls"),
                    scan_filter("orders", "o1", [f("o_orderdate", "GTE", "1995-01-01"), f("o_orderdate", "LTE", "1996-12-31")]), build("custs", ["c_custkey"], "hc"),
                    probe("hc", "o1", ["o_custkey"], "oc", "semi"), build("oc", ["o_orderkey"], "ho"), probe("ho", "ls", ["l_orderkey"], "omc"), build("nation", ["n_nationkey"], "hn"),
                    probe("hn", "omc", ["s_nationkey"], "omn"), {"op": "map", "in": "omn", "fn": "extract_year", "col": "o_orderdate", "as": "o_year", "out": "oy"},
                    groupby("oy", ["o_year"], [agg("sum", REV, "brazil", when=[f("n_name", "EQ", "BRAZIL")]), agg("sum", REV, "total")], "g", est=8),
                    {"op": "map", "in": "g", "expr": {"div": ["brazil", "total"]}, "as": "mkt_share", "out": "gz"}, sort("gz", ["o_year"], "gs"), mat("gs", ["o_year", "mkt_share"])])

PLANS[10] = dict(ref="resources/sql/tpch/10.sql", inputs=["customer", "orders", "lineitem", "nation"],
                 doc="c_custkey = o_custkey and c_nationkey = n_nationkey are foreign keys, so neither join drops a group: the aggregation runs on o_custkey before the customer join and only the 20 winners meet customer and nation (the other GROUP BY columns are functionally dependent on c_custkey)",
                 steps=[scan_filter("orders", "o1", [f("o_orderdate", "GTE", "1993-10-01"), f("o_orderdate", "LT", "1994-01-01")]), scan_filter("lineitem", "l1", [f("l_returnflag", "EQ", "R")]),
                        build("o1", ["o_orderkey"], "ho"), probe("ho", "l1", ["l_orderkey"], "lo"), groupby("lo", ["o_custkey"], [agg("sum", REV, "revenue")], "groups", est_groups_from="rows"),
                        topk("groups", [desc("revenue")], 20, "t20"), mat("t20", ["o_custkey", "revenue"], "top"), build("top", ["o_custkey"], "ht", unique=False),
                        probe("ht", "customer", ["c_custkey"], "ct"), build("nation", ["n_nationkey"], "hn"), probe("hn", "ct", ["c_nationkey"], "ctn"),
                        mat("ctn", ["c_custkey", "c_name", "revenue", "c_acctbal", "n_name"], "named"), topk("named", [desc("revenue")], 20, "fin"),
                        mat("fin", ["c_custkey", "c_name", "revenue", "c_acctbal", "n_name"])])

PLANS[11] = dict(ref="resources/sql/tpch/11.sql", inputs=["partsupp", "supplier", "nation"],
                 doc="ps_supplycost decimal(12,2) x ps_availqty (int32 → decimal(19,0)); the scalar subquery is the SUM over the same groups; sum x 0.0001 has scale 2+4, so after the cast to the common scale the HAVING is value x 10^4 > total in integers, i.e. value > floor(total / 10^4)",
                 steps=[scan_filter("nation", "n1", [f("n_name", "EQ", "GERMANY")]), build("n1", ["n_nationkey"], "hn"), probe("hn", "supplier", ["s_nationkey"], "s1", "semi"),
                        mat("s1", ["s_suppkey"], "supps"), build("supps", ["s_suppkey"], "hs"), probe("hs", "partsupp", ["ps_suppkey"], "ps1", "semi"),
                        groupby("ps1", ["ps_partkey"], [agg("sum", {"mul": ["ps_supplycost", "ps_availqty"]}, "value")], "groups", est={"rows_of": "partsupp", "div": 16, "min": 1024}),
                        groupby("groups", [], [agg("sum", "value", "total")], "total", est=1),
                        scan_filter("groups", "kept", [f("value", "GT", scalar={"from": "total", "col": "total", "div_pow10": 4})]), sort("kept", [desc("value")], "ks"),
                        mat("ks", ["ps_partkey", "value"])])

PLANS[14] = dict(ref="resources/sql/tpch/14.sql", inputs=["part", "lineitem"],
                 doc="LIKE runs once per part, not per lineitem: the PROMO part keys are a hash table, the CASE is a left outer join with it + a NOT NULL condition on its key; 100.00 is decimal(5,2), the product decimal(38,6), the quotient typeAfterDiv → decimal(38,6)",
                 steps=[scan_filter("part", "p1", [f("p_type", "LIKE", "PROMO%")]), mat("p1", [{"col": "p_partkey", "as": "promo_partkey"}], "promo"),
                        scan_filter("lineitem", "l1", [f("l_shipdate", "GTE", "1995-09-01"), f("l_shipdate", "LT", "1995-10-01")]), build("part", ["p_partkey"], "hp"),
                        probe("hp", "l1", ["l_partkey"], "lp", "semi"), build("promo", ["promo_partkey"], "hpromo"), probe("hpromo", "lp", ["l_partkey"], "lpp", "left_outer"),
                        groupby("lpp", [], [agg("sum", REV, "promo_rev", when=[f("promo_partkey", "NOTNULL")]), agg("sum", REV, "total_rev")], "sums", est=1),
                        {"op": "map", "in": "sums", "expr": {"div": [{"mul": ["100.00", "promo_rev"]}, "total_rev"]}, "as": "promo_revenue", "out": "sz"}, mat("sz", ["promo_revenue"])])

PLANS[15] = dict(ref="resources/sql/tpch/15.sql", inputs=["supplier", "lineitem"],
                 doc="the revenue view (SUM per l_suppkey over one quarter), its maximum (top-1: MAX over a 128-bit decimal as an ordered select) read back as the constant of an EQ filter — the reference materialises the scalar subquery first — then ⋈ supplier on the key",
                 steps=[scan_filter("lineitem", "l1", [f("l_shipdate", "GTE", "1996-01-01"), f("l_shipdate", "LT", "1996-04-01")]),
                        groupby("l1", ["l_suppkey"], [agg("sum", REV, "total_revenue")], "groups", est={"rows_of": "lineitem", "div": 512, "min": 1024}),
                        topk("groups", [desc("total_revenue")], 1, "b1"), mat("b1", ["l_suppkey", "total_revenue"], "best"),
                        scan_filter("groups", "w1", [f("total_revenue", "EQ", scalar={"from": "best", "col": "total_revenue"})]), mat("w1", ["l_suppkey", "total_revenue"], "winners"),
                        build("winners", ["l_suppkey"], "hw"), probe("hw", "supplier", ["s_suppkey"], "sw"), mat("sw", ["s_suppkey", "total_revenue"], "joined"),
                        sort("joined", ["s_suppkey"], "js"), mat("js", ["s_suppkey", "total_revenue"])])


def dump_step(st):
    return json.dumps(st, ensure_ascii=False)


HAND_MAINTAINED = {9}  # q9.json was re-ordered by hand in round 4 (the lines probe the orders primary-key index): this script leaves it alone


def main():
    for q, p in sorted(PLANS.items()):
        if q in HAND_MAINTAINED:
            continue
        head = {"name": "tpch_q%d" % q, "ref": p["ref"], "doc": p["doc"], "inputs": p["inputs"]}
        lines = ["{" + json.dumps(head, ensure_ascii=False)[1:-1] + ",", ' "steps": [']
        lines += ["  " + dump_step(s) + ("," if i + 1 < len(p["steps"]) else "") for i, s in enumerate(p["steps"])]
        lines.append(' ], "result": %s}' % json.dumps(p.get("result", "result")))
        with open(os.path.join(OUT, "q%d.json" % q), "w") as f:
            f.write("\n".join(lines) + "\n")
        json.load(open(os.path.join(OUT, "q%d.json" % q)))
    print("wrote", sorted(set(PLANS) - HAND_MAINTAINED))


if __name__ == "__main__":
    main()
