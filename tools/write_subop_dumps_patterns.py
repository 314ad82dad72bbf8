#!/usr/bin/env python3
"""Small dumps for the lowering patterns TPC-H does not exercise (tools/subop_lower.py, the schema of the reference's
`tools/ct/mlir-subop-to-json.cpp`), each with a numpy-checkable answer over the generated tables:
  pat_mark         MarkJoinLowering (:1376-1408): suppliers whose nation is in region 1 OR whose balance exceeds 9000 — the mark is a value
  pat_right_outer  OuterJoinLowering with reverseSides (:1511-1525): every nation with the number of its rich suppliers (0 when none)
  pat_full_outer   FullOuterJoinLowering (:1446-1484): the richest suppliers FULL OUTER JOIN the nations of region 2, counted
  pat_groupjoin[_outer] GroupJoinLowering (:2682-2950), inner (outer: every nation, 0 / NULL without a supplier above 9990): per nation with a rich supplier its name, their number and total balance
  pat_window[_part] WindowLowering (:2193-2553): rank + SUM + COUNT(*) over the suppliers by key, over a 3-row frame / per nation from the partition start
  pat_<set op>     UnionAll / UnionDistinct / CountingSetOperation lowerings (:622-915) over the nation keys of rich customers (balance > 9000) / the richest suppliers (> 9990)
  pat_between      a HALF-OPEN db.between that survives as a residual selection over joined rows (`x >= a and x < b` canonicalised by DBOps.cpp:475-483:
                   lowerInclusive = true, upperInclusive = false — emitter extension E10) next to a db.sub (E1)
Writes tests/golden/subop_pat_*.json."""
from subop_lower import I64_MAX, I64_MIN, Aggregate, C, Cx, GroupJoin, Join, Map, Select, SetOp, Sort, Table, Window, between, dec, eq, gt, or_, result, const, sub

RICH = dec("9000.00", 12, 2)


def mark():
    cx = Cx("pat_mark")
    s, n = Table("supplier"), Table("nation", filters=[("n_regionkey", "EQ", 1)])
    m = C("markjoin0::mark", "int1")
    j = Join("mark", s, n, [(s["s_nationkey"], n["n_nationkey"])], mark=m)
    sel = Select(j, or_(m.j, gt(s["s_acctbal"].j, RICH)))
    return result(cx, Sort(sel, [(s["s_suppkey"], "asc")]), [("s_suppkey", s["s_suppkey"])])


def right_outer():
    cx = Cx("pat_right_outer")
    n, s = Table("nation"), Table("supplier", filters=[("s_acctbal", "GT", "9000.00")])
    oj = C("oj0::s_suppkey", "nullable(int32)")
    j = Join("outer", s, n, [(s["s_nationkey"], n["n_nationkey"])], reverse=True, mapping=[(oj, s["s_suppkey"])])
    cnt = C("aggr0::rich", "int64")
    g = Aggregate(j, [n["n_nationkey"]], [("count", oj, cnt)], nullable_args=[oj])
    return result(cx, Sort(g, [(n["n_nationkey"], "asc")]), [("n_nationkey", n["n_nationkey"]), ("rich", cnt)])


def set_op(kind):
    cx = Cx("pat_" + kind)
    c, s = Table("customer", filters=[("c_acctbal", "GT", "9000.00")]), Table("supplier", filters=[("s_acctbal", "GT", "9990.00")])
    k = C("setop0::nationkey", "nullable(int32)")
    u = SetOp(kind, c, s, [(k, c["c_nationkey"], s["s_nationkey"])])
    return result(cx, Sort(u, [(k, "asc")]), [("nationkey", k)])


def full_outer():
    """FullOuterJoinLowering (:1446-1484): the richest suppliers FULL OUTER JOIN the nations of region 2 — matches, suppliers of other nations (nation NULL)
    and nations of the region without such a supplier (supplier NULL); one row of counts and sums over the nullable columns"""
    cx = Cx("pat_full_outer")
    n, s = Table("nation", filters=[("n_regionkey", "EQ", 2)]), Table("supplier", filters=[("s_acctbal", "GT", "9990.00")])
    fs, fn = C("foj0::s_suppkey", "nullable(int32)"), C("foj0::n_nationkey", "nullable(int32)")
    j = Join("full", s, n, [(s["s_nationkey"], n["n_nationkey"])], mapping=[(fs, s["s_suppkey"]), (fn, n["n_nationkey"])])
    rows, cs, cn, ss, sn = C("aggr0::rows", "int64"), C("aggr0::suppliers", "int64"), C("aggr0::nations", "int64"), C("aggr0::sum_s", "nullable(int64)"), C("aggr0::sum_n", "nullable(int64)")
    g = Aggregate(j, [], [("count_star", None, rows), ("count", fs, cs), ("count", fn, cn), ("sum", fs, ss), ("sum", fn, sn)], nullable_args=[fs, fn])
    return result(cx, g, [("rows", rows), ("suppliers", cs), ("nations", cn), ("sum_s", ss), ("sum_n", sn)])


def groupjoin(behavior="inner"):
    cx = Cx("pat_groupjoin" if behavior == "inner" else "pat_groupjoin_outer")
    n, s = Table("nation"), Table("supplier")
    cnt, tot = C("aggr0::suppliers", "int64"), C("aggr0::balance", "nullable(decimal(38,2))")
    gj = GroupJoin(n, s, [(n["n_nationkey"], s["s_nationkey"])], [("count_star", None, cnt), ("sum", s["s_acctbal"], tot)], stored=[n["n_name"]], predicate=[gt(s["s_acctbal"].j, RICH if behavior == "inner" else dec("9990.00", 12, 2))], behavior=behavior)
    return result(cx, Sort(gj, [(s["s_nationkey"], "asc")]), [("s_nationkey", s["s_nationkey"]), ("n_name", n["n_name"]), ("suppliers", cnt), ("balance", tot)])


def window_static(partitioned):
    """a frame unbounded on both sides: the share of a supplier's balance in its nation's (or everybody's) total needs SUM and COUNT(*) of the whole partition"""
    cx = Cx("pat_window_total_part" if partitioned else "pat_window_total")
    s = Table("supplier")
    total, cnt = C("win0::total", "nullable(decimal(38,2))"), C("win0::rows", "int64")
    w = Window(s, [s["s_nationkey"]] if partitioned else [], [], (I64_MIN, I64_MAX), [("sum", s["s_acctbal"], total), ("count_star", None, cnt)])
    return result(cx, Sort(w, [(s["s_suppkey"], "asc")]), [("s_suppkey", s["s_suppkey"]), ("total", total), ("rows", cnt)])


def window(partitioned):
    """rank and running / moving aggregates over the suppliers ordered by their key: per nation with an unbounded-preceding frame, or over all of them
    with ROWS BETWEEN 2 PRECEDING AND CURRENT ROW"""
    cx = Cx("pat_window_part" if partitioned else "pat_window")
    s = Table("supplier")
    rank, total, cnt = C("win0::rank", "int64"), C("win0::balance", "nullable(decimal(38,2))"), C("win0::rows", "int64")
    w = Window(s, [s["s_nationkey"]] if partitioned else [], [(s["s_suppkey"], "asc")], (I64_MIN, 0) if partitioned else (-2, 0),
               [("rank", None, rank), ("sum", s["s_acctbal"], total), ("count_star", None, cnt)])
    return result(cx, Sort(w, [(s["s_suppkey"], "asc")]), [("s_suppkey", s["s_suppkey"]), ("rank", rank), ("balance", total), ("rows", cnt)])


def between_half_open():
    """suppliers with 10 <= s_suppkey < 20 joined with their nation: the range is a residual db.between over the joined rows, the result also shows
    s_suppkey - n_regionkey (db.sub)"""
    cx = Cx("pat_between")
    s, n = Table("supplier"), Table("nation")
    j = Join("inner", s, n, [(s["s_nationkey"], n["n_nationkey"])])
    sel = Select(j, between(s["s_suppkey"].j, const(10, "int32"), const(20, "int32"), lower_inclusive=True, upper_inclusive=False))
    diff = C("map0::diff", "int32")
    m = Map(sel, [(diff, sub(s["s_suppkey"].j, n["n_regionkey"].j))])
    return result(cx, Sort(m, [(s["s_suppkey"], "asc")]), [("s_suppkey", s["s_suppkey"]), ("diff", diff)])


SET_KINDS = ("union_all", "union", "intersect", "except", "intersect_all", "except_all")

if __name__ == "__main__":
    print(mark())
    print(between_half_open())
    print(right_outer())
    print(full_outer())
    print(groupjoin())
    print(groupjoin("outer"))
    print(window(False))
    print(window(True))
    print(window_static(False))
    print(window_static(True))
    for kind in SET_KINDS:
        print(set_op(kind))
