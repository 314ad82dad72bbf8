#!/usr/bin/env python3
"""Writes the SHARDED TPC-H plans (lingo-db_amd/plans/tpch/dist/qN.json): the same step vocabulary as the
single-GPU plans plus the two exchange steps of SURVEY §8(e),

  {"op": "allgather", "in": table, "out": name}                      replicate a small table on every rank
  {"op": "shuffle", "in": rel, "keys": [...], "cols": [...], "out": name}   hash-radix re-partition on db.hash(keys)

Every rank runs the SAME plan text over its shard (ldb_plan_run_json_comm).  Sharding of the inputs
(include/ldb_tpchgen.h): orders and lineitem by order ranges (co-located), customer / part / partsupp /
supplier by row ranges, nation / region replicated.  "replicated_inputs" names static dimension tables the
host program all-gathers ONCE per database (a one-step plan) and passes in like any other input.

The reference has no multi-GPU path (src/runtime/GPU/CUDA/CMakeLists.txt:9); the shapes below are this
repo's design: partial aggregation + all-gather + merge for low-cardinality results, one shuffle per
repartitioned input for joins / group-bys that are not co-partitioned, small build sides replicated.
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SINGLE = os.path.join(ROOT, "lingo-db_amd", "plans", "tpch")
OUT = os.path.join(SINGLE, "dist")

REV = {"mul": ["l_extendedprice", {"sub": [1, "l_discount"]}]}


def single(q):
    with open(os.path.join(SINGLE, "q%d.json" % q)) as f:
        return json.load(f)


def steps_until(q, last_out):
    """the single-GPU plan's steps up to and including the one that produces `last_out`"""
    out = []
    for st in single(q)["steps"]:
        out.append(st)
        if st.get("out") == last_out:
            return out
    raise KeyError(last_out)


def S(op, **kw):
    d = {"op": op}
    d.update(kw)
    return d


def mat(inp, cols, out):
    return S("materialize", **{"in": inp, "cols": cols, "out": out})


def gather(inp, out):
    return S("allgather", **{"in": inp, "out": out})


def sums(names, typ=None):
    aggs = []
    for n in names:
        a = {"fn": "sum", "expr": n, "as": n}
        if typ:
            a["type"] = typ
        aggs.append(a)
    return aggs


PLANS = {}


def plan(q, doc, inputs, steps, result="result", replicated=None):
    p = {"name": "tpch_q%d_dist" % q, "ref": "resources/sql/tpch/%d.sql" % q, "doc": doc, "inputs": inputs}
    if replicated:
        p["replicated_inputs"] = replicated
        p["inputs"] = inputs + sorted(replicated)
    p["steps"] = steps
    p["result"] = result
    PLANS[q] = p


# ------------------------------------------------------------------ Q1, Q6: partial aggregates, all-gather, merge
q1 = single(1)["steps"][0]
plan(1, "shard-local partial aggregation (sums and the row count; AVG needs both), all-gather of the <= 6-row partial tables, merge = the reference's combine step "
        "(add sums, add counts, AVG = merged sum / merged count)", ["lineitem"], [
    S("groupby", **{"in": "lineitem", "keys": q1["keys"], "preds": q1["preds"], "est_groups": 6, "out": "partial", "aggs": [
        {"fn": "sum", "expr": "l_quantity", "as": "sum_qty"}, {"fn": "sum", "expr": "l_extendedprice", "as": "sum_base_price"},
        {"fn": "sum", "expr": REV, "as": "sum_disc_price"}, {"fn": "sum", "expr": {"mul": ["l_extendedprice", {"sub": [1, "l_discount"]}, {"add": [1, "l_tax"]}]}, "as": "sum_charge"},
        {"fn": "sum", "expr": "l_discount", "as": "sum_disc"}, {"fn": "count_star", "as": "cnt"}]}),
    gather("partial", "partials"),
    S("groupby", **{"in": "partials", "keys": q1["keys"], "est_groups": 6, "out": "g", "aggs": sums(["sum_qty", "sum_base_price", "sum_disc_price", "sum_charge"]) + [
        {"fn": "avg", "expr": "sum_qty", "count": "cnt", "as": "avg_qty"}, {"fn": "avg", "expr": "sum_base_price", "count": "cnt", "as": "avg_price"},
        {"fn": "avg", "expr": "sum_disc", "count": "cnt", "as": "avg_disc"}, {"fn": "sum", "expr": "cnt", "as": "count_order"}]}),
] + single(1)["steps"][1:])

q6 = single(6)["steps"][0]
plan(6, "shard-local key-less SUM (NULL where nothing passes), all-gather, SUM of the partials (NULLs ignored)", ["lineitem"], [
    dict(q6, out="part"),
    gather("part", "parts"),
    S("groupby", **{"in": "parts", "keys": [], "aggs": sums(["revenue"]), "est_groups": 1, "out": "revenue"}),
], result="revenue")

# ------------------------------------------------------------------ Q3
s3 = single(3)["steps"]
plan(3, "the filtered customer keys are replicated (small build side); orders and lineitem are co-located, so both joins and the group-by are shard-local "
        "(order keys are disjoint across shards); the shard top-10s are all-gathered and cut to the global top-10", ["customer", "orders", "lineitem"], [
    s3[0], mat("c1", ["c_custkey"], "ck"), gather("ck", "ck_all"), s3[1], s3[2],
    S("join_build", **{"in": "ck_all", "keys": ["c_custkey"], "unique": True, "out": "hc"}),
    s3[4], s3[5], s3[6], s3[7], s3[8],
    mat("top", ["l_orderkey", "revenue", "o_orderdate", "o_shippriority"], "tl"),
    gather("tl", "tops"),
    S("topk", **{"in": "tops", "by": [{"col": "revenue", "desc": True}, "o_orderdate"], "k": 10, "out": "t10"}),
    mat("t10", ["l_orderkey", "revenue", "o_orderdate", "o_shippriority"], "result"),
])

# ------------------------------------------------------------------ Q4, Q12: co-located, merge the per-shard counts
plan(4, "orders and their lineitems are co-located: the single-GPU plan per shard, then the per-shard counts are added", ["orders", "lineitem"],
     [dict(st, out="partial") if st.get("out") == "g" else st for st in steps_until(4, "g")] + [
    gather("partial", "partials"),
    S("groupby", **{"in": "partials", "keys": ["o_orderpriority"], "aggs": sums(["order_count"]), "est_groups": 5, "out": "g"}),
] + single(4)["steps"][-2:])
plan(12, "co-located like Q4; the two conditional sums stay int32 (SUM keeps the type)", ["orders", "lineitem"],
     [dict(st, out="partial") if st.get("out") == "g" else st for st in steps_until(12, "g")] + [
    gather("partial", "partials"),
    S("groupby", **{"in": "partials", "keys": ["l_shipmode"], "aggs": sums(["high_line_count", "low_line_count"], "int32"), "est_groups": 2, "out": "g"}),
] + single(12)["steps"][-2:])


def with_gathers(steps, names):
    """after the step that produces table `n` insert an all-gather to `n`_all and make later steps read that"""
    out = []
    renamed = {}
    for st in steps:
        st = dict(st)
        for f in ("in", "ht", "table"):
            if st.get(f) in renamed:
                st[f] = renamed[st[f]]
        out.append(st)
        if st.get("out") in names:
            out.append(gather(st["out"], st["out"] + "_all"))
            renamed[st["out"]] = st["out"] + "_all"
    return out


# ------------------------------------------------------------------ Q5, Q7, Q8: replicate the reduced dimension tables
s5 = single(5)["steps"]
i5 = [i for i, st in enumerate(s5) if st.get("out") == "partial"][0]
plan(5, "the region's customers and suppliers, reduced to (key, nationkey), are all-gathered; the joins and the partial SUM per nation key are shard-local; "
        "the <= 25-row partials are all-gathered and merged", ["customer", "orders", "lineitem", "supplier", "nation", "region"],
     with_gathers(s5[:i5 + 1], {"custs", "supps", "partial"}) + [dict(st, **({"in": "partial_all"} if st.get("in") == "partial" else {})) for st in s5[i5 + 1:]])
s7 = single(7)["steps"]
i7 = [i for i, st in enumerate(s7) if st.get("out") == "partial"][0]
plan(7, "like Q5: customers and suppliers of the two nations are all-gathered as (key, nationkey); the nation-pair condition, extract(year) and the partial sums are shard-local",
     ["customer", "orders", "lineitem", "supplier", "nation"],
     with_gathers(s7[:i7 + 1], {"custs", "supps", "partial"}) + [dict(st, **({"in": "partial_all"} if st.get("in") == "partial" else {})) for st in s7[i7 + 1:]])
s8 = [dict(st) for st in single(8)["steps"]]
for st in s8:
    if st.get("in") == "supplier":
        st["in"] = "supplier_all"
i8 = [i for i, st in enumerate(s8) if st.get("out") == "g"][0]
s8[i8]["out"] = "partial"
plan(8, "the part keys of the type (1/150 of part) and the region's customers are all-gathered, supplier is replicated once; lineitem ⋈ part keys ⋈ supplier, the orders of the two "
        "years, the customer semi join and the two sums per year are shard-local; the <= 2-row partials are added before the ratio is formed",
     ["part", "lineitem", "orders", "customer", "nation", "region"],
     with_gathers(s8[:i8 + 1], {"parts", "custs", "partial"}) + [
    S("groupby", **{"in": "partial_all", "keys": ["o_year"], "aggs": sums(["brazil", "total"]), "est_groups": 8, "out": "g"})] + s8[i8 + 1:],
     replicated={"supplier_all": {"table": "supplier"}})

# ------------------------------------------------------------------ Q9 (BASELINE configs[4]): one shuffle per repartitioned side
AMOUNT = {"sub": [REV, {"mul": ["ps_supplycost", "l_quantity"]}]}
plan(9, "SURVEY §8(e): the keys of the green parts are all-gathered (small); every rank reduces its lineitem and partsupp shards with them; the surviving lineitems meet their "
        "co-located orders (→ o_year) and are hash-radix partitioned on l_partkey, the green partsupp rows on ps_partkey: ONE all-to-all per side co-partitions them; the "
        "(partkey, suppkey) join, the supplier join (replicated once) and the partial aggregation are local; the <= 175-row partials are all-gathered and summed",
     ["part", "lineitem", "partsupp", "orders", "nation"], [
    S("filter", **{"in": "part", "out": "p1", "preds": [{"col": "p_name", "op": "LIKE", "value": "%green%"}]}),
    mat("p1", ["p_partkey"], "green"), gather("green", "green_all"),
    S("join_build", **{"in": "green_all", "keys": ["p_partkey"], "unique": True, "out": "hp"}),
    S("join_probe", ht="hp", **{"in": "lineitem", "keys": ["l_partkey"], "kind": "semi", "out": "lp"}),
    S("join_build", **{"in": "orders", "keys": ["o_orderkey"], "unique": True, "out": "ho"}),
    S("join_probe", ht="ho", **{"in": "lp", "keys": ["l_orderkey"], "kind": "inner", "out": "lpo"}),
    S("map", **{"in": "lpo", "fn": "extract_year", "col": "o_orderdate", "as": "o_year", "out": "lpy"}),
    S("shuffle", **{"in": "lpy", "keys": ["l_partkey"], "cols": ["l_partkey", "l_suppkey", "l_quantity", "l_extendedprice", "l_discount", "o_year"], "out": "lrows"}),
    S("join_probe", ht="hp", **{"in": "partsupp", "keys": ["ps_partkey"], "kind": "semi", "out": "ps1"}),
    S("shuffle", **{"in": "ps1", "keys": ["ps_partkey"], "cols": ["ps_partkey", "ps_suppkey", "ps_supplycost"], "out": "psrows"}),
    S("join_build", **{"in": "psrows", "keys": ["ps_partkey", "ps_suppkey"], "unique": True, "out": "hps"}),
    S("join_probe", ht="hps", **{"in": "lrows", "keys": ["l_partkey", "l_suppkey"], "kind": "inner", "out": "lps"}),
    S("join_build", **{"in": "supplier_all", "keys": ["s_suppkey"], "unique": True, "out": "hs"}),
    S("join_probe", ht="hs", **{"in": "lps", "keys": ["l_suppkey"], "kind": "inner", "out": "lpss"}),
    S("groupby", **{"in": "lpss", "keys": ["s_nationkey", "o_year"], "aggs": [{"fn": "sum", "expr": AMOUNT, "as": "amount"}], "est_groups": 200, "out": "partial"}),
    gather("partial", "partials"),
    S("join_build", **{"in": "nation", "keys": ["n_nationkey"], "unique": True, "out": "hn"}),
    S("join_probe", ht="hn", **{"in": "partials", "keys": ["s_nationkey"], "kind": "inner", "out": "pn"}),
    S("groupby", **{"in": "pn", "keys": ["n_name", "o_year"], "aggs": [{"fn": "sum", "expr": "amount", "as": "sum_profit"}], "est_groups": 200, "out": "g"}),
    S("sort", **{"in": "g", "by": ["n_name", {"col": "o_year", "desc": True}], "out": "gs"}),
    mat("gs", ["n_name", "o_year", "sum_profit"], "result"),
], replicated={"supplier_all": {"table": "supplier"}})

# ------------------------------------------------------------------ Q10, Q15, Q11: the group-by exchange
s10 = single(10)["steps"]
plan(10, "orders and their lineitems are co-located but a customer's orders are spread over the shards: the shard-local (o_custkey, revenue) groups are shuffled on the key and "
         "merged; every rank's 20 best merged groups are all-gathered, the global 20 looked up in the row-sharded customer table, and the gathered rows ordered",
     ["customer", "orders", "lineitem", "nation"], s10[:4] + [
    dict(s10[4], out="gl"),
    S("shuffle", **{"in": "gl", "keys": ["o_custkey"], "cols": ["o_custkey", "revenue"], "out": "grows"}),
    S("groupby", **{"in": "grows", "keys": ["o_custkey"], "aggs": sums(["revenue"]), "est_groups_from": "rows", "out": "groups"}),
    s10[5], mat("t20", ["o_custkey", "revenue"], "tl"), gather("tl", "tops"),
    S("topk", **{"in": "tops", "by": [{"col": "revenue", "desc": True}], "k": 20, "out": "t20g"}),
    mat("t20g", ["o_custkey", "revenue"], "top"),
] + s10[7:11] + [
    mat("ctn", ["c_custkey", "c_name", "revenue", "c_acctbal", "n_name"], "nl"), gather("nl", "named"),
] + s10[12:])
s15 = single(15)["steps"]
plan(15, "a supplier's lineitems are spread over the shards: exchange + merge of the (l_suppkey, revenue) groups; the maximum of the ranks' best groups is the view's maximum; "
         "every rank keeps its groups that reach it; the gathered winners meet the row-sharded supplier table", ["supplier", "lineitem"], [
    s15[0], dict(s15[1], out="gl"),
    S("shuffle", **{"in": "gl", "keys": ["l_suppkey"], "cols": ["l_suppkey", "total_revenue"], "out": "grows"}),
    S("groupby", **{"in": "grows", "keys": ["l_suppkey"], "aggs": sums(["total_revenue"]), "est_groups_from": "rows", "out": "groups"}),
    s15[2], mat("b1", ["l_suppkey", "total_revenue"], "bl"), gather("bl", "bests"),
    S("topk", **{"in": "bests", "by": [{"col": "total_revenue", "desc": True}], "k": 1, "out": "b1g"}),
    mat("b1g", ["l_suppkey", "total_revenue"], "best"),
    s15[4], mat("w1", ["l_suppkey", "total_revenue"], "wl"), gather("wl", "winners"),
    s15[6], s15[7], mat("sw", ["s_suppkey", "total_revenue"], "jl"), gather("jl", "joined"),
] + s15[9:])
s11 = single(11)["steps"]
plan(11, "partsupp is sharded by rows, a part's four rows may straddle two shards: the shard-local (ps_partkey, value) groups are shuffled on the key and merged; the scalar "
         "subquery's total is the sum of the ranks' totals; the survivors of the HAVING filter are all-gathered and sorted", ["partsupp", "supplier", "nation"],
     s11[:4] + [gather("supps", "supps_all"), dict(s11[4], **{"in": "supps_all"}), s11[5], dict(s11[6], out="gl", est_groups_from="rows"),
    S("shuffle", **{"in": "gl", "keys": ["ps_partkey"], "cols": ["ps_partkey", "value"], "out": "grows"}),
    S("groupby", **{"in": "grows", "keys": ["ps_partkey"], "aggs": sums(["value"]), "est_groups_from": "rows", "out": "groups"}),
    dict(s11[7], out="tl"), gather("tl", "tls"),
    S("groupby", **{"in": "tls", "keys": [], "aggs": sums(["total"]), "est_groups": 1, "out": "total"}),
    s11[8], mat("kept", ["ps_partkey", "value"], "kl"), gather("kl", "kall"),
    S("sort", **{"in": "kall", "by": [{"col": "value", "desc": True}], "out": "ks"}),
    s11[10]])
for st in PLANS[11]["steps"]:
    st.pop("est_groups", None) if st.get("out") == "gl" else None

# ------------------------------------------------------------------ Q13: both sides shuffled on the customer key
s13 = single(13)["steps"]
plan(13, "orders are sharded by order ranges, customers by rows, o_custkey is random: the shard-local counts per o_custkey are shuffled on the key and merged, the customer keys are "
         "shuffled by the same hash (db.hash of an int32 key is the same on both sides), the outer join and the count distribution are local; the <= 64-row distributions are added",
     ["customer", "orders"], [
    s13[0], dict(s13[1], out="ocl", est_groups_from="rows"),
    S("shuffle", **{"in": "ocl", "keys": ["o_custkey"], "cols": ["o_custkey", "c_cnt"], "out": "ocr"}),
    S("groupby", **{"in": "ocr", "keys": ["o_custkey"], "aggs": sums(["c_cnt"]), "est_groups_from": "rows", "out": "oc"}),
    S("shuffle", **{"in": "customer", "keys": ["c_custkey"], "cols": ["c_custkey"], "out": "cpart"}),
    s13[2], dict(s13[3], **{"in": "cpart"}), s13[4], dict(s13[5], out="gl"),
    gather("gl", "gall"),
    S("groupby", **{"in": "gall", "keys": ["c_count"], "aggs": sums(["custdist"]), "est_groups": 64, "out": "g"}),
] + s13[6:])
for st in PLANS[13]["steps"]:
    if st.get("out") == "ocl":
        st.pop("est_groups", None)

# ------------------------------------------------------------------ Q14
s14 = [dict(st) for st in single(14)["steps"]]
for st in s14:
    if st["op"] == "join_build" and st["in"] == "part":
        st["in"] = "part_keys_all"
i14 = [i for i, st in enumerate(s14) if st.get("out") == "sums"][0]
s14[i14]["out"] = "sl"
plan(14, "lineitem is sharded by orders, part by rows: the PROMO part keys (1/6 of part) are all-gathered per query, the part key column (a static dimension) once; the two partial "
         "sums are added before the ratio is formed", ["part", "lineitem"],
     with_gathers(s14[:i14 + 1], {"promo", "sl"}) + [
    S("groupby", **{"in": "sl_all", "keys": [], "aggs": sums(["promo_rev", "total_rev"]), "est_groups": 1, "out": "sums"})] + s14[i14 + 1:],
     replicated={"part_keys_all": {"table": "part", "cols": ["p_partkey"]}})

# ------------------------------------------------------------------ Q16: two shuffles (part key, then the group key)
s16 = single(16)["steps"]
plan(16, "the complaint suppliers' keys and the selected parts' keys are all-gathered (key columns only); partsupp is reduced with both on its shard, then its survivors and the "
         "selected part rows are co-partitioned on the part key; COUNT(DISTINCT ps_suppkey) needs every (brand, type, size) group on one rank: the locally de-duplicated "
         "(brand, type, size, suppkey) rows are shuffled on the three group keys, de-duplicated again and counted", ["part", "partsupp", "supplier"], [
    s16[0], mat("sc", ["s_suppkey"], "sck"), gather("sck", "sck_all"),
    s16[1], mat("p1", ["p_partkey"], "pk"), gather("pk", "pk_all"),
    S("join_build", **{"in": "pk_all", "keys": ["p_partkey"], "unique": True, "out": "hpk"}),
    S("join_probe", ht="hpk", **{"in": "partsupp", "keys": ["ps_partkey"], "kind": "semi", "out": "ps1"}),
    S("join_build", **{"in": "sck_all", "keys": ["s_suppkey"], "unique": True, "out": "hsc"}),
    S("join_probe", ht="hsc", **{"in": "ps1", "keys": ["ps_suppkey"], "kind": "anti", "out": "ps2"}),
    S("shuffle", **{"in": "ps2", "keys": ["ps_partkey"], "cols": ["ps_partkey", "ps_suppkey"], "out": "psr"}),
    S("shuffle", **{"in": "p1", "keys": ["p_partkey"], "cols": ["p_partkey", "p_brand", "p_type", "p_size"], "out": "p1r"}),
    S("join_build", **{"in": "p1r", "keys": ["p_partkey"], "unique": True, "out": "hp"}),
    S("join_probe", ht="hp", **{"in": "psr", "keys": ["ps_partkey"], "kind": "inner", "out": "pp"}),
    S("groupby", **{"in": "pp", "keys": ["p_brand", "p_type", "p_size", "ps_suppkey"], "aggs": [{"fn": "count_star", "as": "n"}], "est_groups_from": "rows", "out": "dl"}),
    S("shuffle", **{"in": "dl", "keys": ["p_brand", "p_type", "p_size"], "cols": ["p_brand", "p_type", "p_size", "ps_suppkey"], "out": "dr"}),
    S("groupby", **{"in": "dr", "keys": ["p_brand", "p_type", "p_size", "ps_suppkey"], "aggs": [{"fn": "count_star", "as": "n"}], "est_groups_from": "rows", "out": "d"}),
    S("groupby", **{"in": "d", "keys": ["p_brand", "p_type", "p_size"], "aggs": [{"fn": "count_star", "as": "supplier_cnt"}], "est_groups": 20000, "out": "gl"}),
    mat("gl", ["p_brand", "p_type", "p_size", "supplier_cnt"], "glm"), gather("glm", "g"),
] + s16[-2:])

# ------------------------------------------------------------------ Q17
s17 = single(17)["steps"]
i17 = [i for i, st in enumerate(s17) if st.get("out") == "t"][0]
plan(17, "the selected part keys (1/1000 of part) are all-gathered; the few lineitems of those parts are shuffled on l_partkey (the average is per part over ALL its lineitems), "
         "the correlated average, the join back and the partial SUM are local; the partial sums are added before the division", ["lineitem", "part"], [
    s17[0], mat("p1", ["p_partkey"], "pk"), gather("pk", "pk_all"),
    S("join_build", **{"in": "pk_all", "keys": ["p_partkey"], "unique": True, "out": "hp"}),
    dict(s17[2], out="l0"),
    S("shuffle", **{"in": "l0", "keys": ["l_partkey"], "cols": ["l_partkey", "l_quantity", "l_extendedprice"], "out": "l1"}),
] + s17[3:i17] + [dict(s17[i17], out="tl"), gather("tl", "tls"),
    S("groupby", **{"in": "tls", "keys": [], "aggs": sums(["s"]), "est_groups": 1, "out": "t"})] + s17[i17 + 1:])

# ------------------------------------------------------------------ Q18
s18 = single(18)["steps"]
plan(18, "the 1.5 M x SF-group aggregation, the HAVING and the order / lineitem joins are shard-local (co-located); the shard top-100s are all-gathered and cut to the global "
         "top-100; every rank looks up c_name for the winners whose customer row it owns; the gathered rows are ordered", ["customer", "orders", "lineitem"], [
    s18[0], s18[1], s18[2], s18[3],
    S("join_build", **{"in": "o1", "keys": ["o_orderkey"], "unique": True, "out": "ho1"}),
    S("join_probe", ht="ho1", **{"in": "lineitem", "keys": ["l_orderkey"], "kind": "inner", "out": "lo"}),
    S("groupby", **{"in": "lo", "keys": ["o_custkey", "o_orderkey", "o_orderdate", "o_totalprice"], "aggs": [{"fn": "sum", "expr": "l_quantity", "as": "sum_quantity"}],
                    "est_groups": {"rows_of": "o1"}, "out": "gl"}),
    S("topk", **{"in": "gl", "by": [{"col": "o_totalprice", "desc": True}, "o_orderdate"], "k": 100, "out": "tl"}),
    mat("tl", ["o_custkey", "o_orderkey", "o_orderdate", "o_totalprice", "sum_quantity"], "tlm"), gather("tlm", "tops"),
    S("topk", **{"in": "tops", "by": [{"col": "o_totalprice", "desc": True}, "o_orderdate"], "k": 100, "out": "t100"}),
    mat("t100", ["o_custkey", "o_orderkey", "o_orderdate", "o_totalprice", "sum_quantity"], "top"),
    S("join_build", **{"in": "top", "keys": ["o_custkey"], "unique": False, "out": "ht"}),
    S("join_probe", ht="ht", **{"in": "customer", "keys": ["c_custkey"], "kind": "inner", "out": "ct"}),
    mat("ct", ["c_name", "c_custkey", "o_orderkey", "o_orderdate", "o_totalprice", "sum_quantity"], "nl"), gather("nl", "named"),
    S("topk", **{"in": "named", "by": [{"col": "o_totalprice", "desc": True}, "o_orderdate"], "k": 100, "out": "fin"}),
    mat("fin", ["c_name", "c_custkey", "o_orderkey", "o_orderdate", "o_totalprice", "sum_quantity"], "result"),
])

# ------------------------------------------------------------------ Q19
s19 = single(19)["steps"]
plan(19, "the selected parts (key and brand; 48 k of 20 M at SF100) are all-gathered, lineitem probes them on its shard, the partial sums are added", ["lineitem", "part"], [
    s19[0], mat("p1", ["p_partkey", "p_brand"], "pl"), gather("pl", "p_all"),
    dict(s19[1], **{"in": "p_all"}), s19[2], s19[3], s19[4], dict(s19[5], out="rl"),
    gather("rl", "rls"),
    S("groupby", **{"in": "rls", "keys": [], "aggs": sums(["revenue"]), "est_groups": 1, "out": "result"}),
])

# ------------------------------------------------------------------ Q20
s20 = single(20)["steps"]
plan(20, "the forest part keys (1 %) are all-gathered; partsupp and the year's lineitems are reduced with them on their shards; the (partkey, suppkey) quantity sums are "
         "merged after a shuffle on the part key, the reduced partsupp rows are shuffled the same way; the qualifying supplier keys are all-gathered and meet the "
         "row-sharded CANADA suppliers; the gathered names are sorted", ["lineitem", "part", "partsupp", "supplier", "nation"], [
    s20[0], mat("p1", ["p_partkey"], "pk"), gather("pk", "pk_all"),
    dict(s20[1], **{"in": "pk_all"}), dict(s20[2], out="ps0"), s20[3], s20[4],
    dict(s20[5], out="lql", key_names=["l_partkey", "l_suppkey"]),
    S("shuffle", **{"in": "lql", "keys": ["l_partkey"], "cols": ["l_partkey", "l_suppkey", "sum_qty"], "out": "lqr"}),
    S("groupby", **{"in": "lqr", "keys": ["l_partkey", "l_suppkey"], "aggs": sums(["sum_qty"]), "est_groups_from": "rows", "key_names": ["q_partkey", "q_suppkey"], "out": "lq"}),
    S("shuffle", **{"in": "ps0", "keys": ["ps_partkey"], "cols": ["ps_partkey", "ps_suppkey", "ps_availqty"], "out": "ps1"}),
    s20[6], s20[7], s20[8], s20[9],
    mat("pq3", ["ps_suppkey"], "qk"), gather("qk", "qk_all"),
    s20[10], s20[11], s20[12], s20[13],
    dict(s20[14], **{"in": "qk_all"}),
    mat("s2", ["s_name", "s_address"], "sl"), gather("sl", "sall"),
    S("sort", **{"in": "sall", "by": ["s_name"], "out": "s3"}),
    s20[16],
])

# ------------------------------------------------------------------ Q21
s21 = single(21)["steps"]
i21 = [i for i, st in enumerate(s21) if st.get("out") == "g"][0]
plan(21, "the nation's suppliers (key, name) are all-gathered; every EXISTS / NOT EXISTS is over the lineitems of ONE order, co-located with it: all joins are shard-local; the "
         "per-supplier counts are added and the top 100 taken", ["lineitem", "orders", "supplier", "nation"], [
    s21[0], s21[1], s21[2], mat("s1", ["s_suppkey", "s_name"], "sl"), gather("sl", "s_all"),
    dict(s21[3], **{"in": "s_all"}),
] + s21[4:i21] + [dict(s21[i21], out="gl"), gather("gl", "gall"),
    S("groupby", **{"in": "gall", "keys": ["s_name"], "aggs": sums(["numwait"]), "est_groups_from": "rows", "out": "g"})] + s21[i21 + 1:])

# ------------------------------------------------------------------ Q22
s22 = single(22)["steps"]
plan(22, "customers are sharded by rows, orders by order ranges: the average balance is merged from (sum, count) partials; the candidate customers' keys are all-gathered, every "
         "rank reports which of them have an order on its shard, the gathered reports remove them (NOT EXISTS); the per-code partials are added", ["customer", "orders"], [
    s22[0], s22[1], s22[2],
    S("groupby", **{"in": "c3", "keys": [], "aggs": [{"fn": "sum", "expr": "c_acctbal", "as": "s"}, {"fn": "count_star", "as": "n"}], "est_groups": 1, "out": "al"}),
    gather("al", "als"),
    S("groupby", **{"in": "als", "keys": [], "aggs": [{"fn": "avg", "expr": "s", "count": "n", "as": "avg_bal"}], "est_groups": 1, "out": "t_avg"}),
    s22[4],
    mat("c4", ["c_custkey"], "ck"), gather("ck", "ck_all"),
    S("join_build", **{"in": "ck_all", "keys": ["c_custkey"], "unique": True, "out": "hca"}),
    S("join_probe", ht="hca", **{"in": "orders", "keys": ["o_custkey"], "kind": "semi_build", "out": "withord"}),
    mat("withord", [{"col": "c_custkey", "as": "w_custkey"}], "wl"), gather("wl", "wall"),
    S("join_build", **{"in": "wall", "keys": ["w_custkey"], "unique": False, "out": "hw"}),
    S("join_probe", ht="hw", **{"in": "c4", "keys": ["c_custkey"], "kind": "anti", "out": "c5"}),
    dict(s22[7], out="gl"), gather("gl", "gall"),
    S("groupby", **{"in": "gall", "keys": ["cntrycode"], "aggs": sums(["numcust", "totacctbal"]), "est_groups": 8, "out": "g"}),
] + s22[8:])

# ------------------------------------------------------------------ Q2
s2 = single(2)["steps"]
plan(2, "the selected parts (0.4 %) and the region's suppliers are all-gathered; partsupp meets both on its shard; the few joined offers are shuffled on the part key (a part's "
        "four partsupp rows may straddle two shards), the correlated MIN and the join back are local; the shard top-100s are all-gathered and cut",
     ["part", "supplier", "partsupp", "nation", "region"], s2[:5] + [
    mat("sn", ["s_suppkey", "s_acctbal", "s_name", "n_name", "s_address", "s_phone", "s_comment"], "snl"), gather("snl", "sn_all"),
    s2[5], mat("p1", ["p_partkey", "p_mfgr"], "pl"), gather("pl", "p_all"),
    dict(s2[6], **{"in": "p_all"}), s2[7], dict(s2[8], **{"in": "sn_all"}), dict(s2[9], out="all0"),
    S("shuffle", **{"in": "all0", "keys": ["ps_partkey"], "cols": ["ps_partkey", "ps_supplycost", "s_acctbal", "s_name", "n_name", "p_partkey", "p_mfgr", "s_address", "s_phone", "s_comment"],
                    "out": "all"}),
    s2[10], s2[11], s2[12], s2[13],
    mat("top", ["s_acctbal", "s_name", "n_name", "p_partkey", "p_mfgr", "s_address", "s_phone", "s_comment"], "tl"), gather("tl", "tops"),
    S("topk", **{"in": "tops", "by": [{"col": "s_acctbal", "desc": True}, "n_name", "s_name", "p_partkey"], "k": 100, "out": "t100"}),
    mat("t100", ["s_acctbal", "s_name", "n_name", "p_partkey", "p_mfgr", "s_address", "s_phone", "s_comment"], "result"),
])


def main():
    os.makedirs(OUT, exist_ok=True)
    for q, p in sorted(PLANS.items()):
        with open(os.path.join(OUT, "q%d.json" % q), "w") as f:
            f.write('{"name": %s, "ref": %s, "doc": %s, "inputs": %s,%s\n "steps": [\n' % (
                json.dumps(p["name"]), json.dumps(p["ref"]), json.dumps(p["doc"], ensure_ascii=False), json.dumps(p["inputs"]),
                (' "replicated_inputs": %s,' % json.dumps(p["replicated_inputs"])) if "replicated_inputs" in p else ""))
            f.write(",\n".join("  " + json.dumps(st, ensure_ascii=False) for st in p["steps"]))
            f.write('\n ], "result": %s}\n' % json.dumps(p["result"]))
    print("wrote %d sharded plans to %s" % (len(PLANS), OUT))


if __name__ == "__main__":
    main()
