"""TPC-H Q1..Q22 over the ORACLE (oracle/ldb_oracle.c = the CPU restatement of the reference's
sub-operator path) — TEST INFRASTRUCTURE: the checker of the GPU plans at sizes the C oracle
finishes in seconds (tests/test_gpu_sf1_oracle.py), and the reported `cpu_baseline` of bench.py.
Never imported by the product.

Every scan + pushed-down filter, hash join and hash aggregation over base-table-sized inputs runs in
the C oracle on `threads` workers (morsels of 20 000 rows, the reference's ScanBatchesTask /
HashIndexedView / PreAggregationHashtable restated); what follows the first aggregation — a few
thousand rows of decimal arithmetic, HAVING, ORDER BY … LIMIT — is numpy / Python integers with the
reference's decimal typing (sql_analyzer.cpp:3058-3159) spelled out where it matters.

`Legs(n_orders, threads).run(q)` returns the rows of resources/sql/tpch/<q>.sql in the column order
of the GPU plan's result (decimals as unscaled integers, dates as day numbers, char(1) as the int32
of its 4 bytes, strings as str)."""
import os
import sys

import numpy as np
import pyarrow as pa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lingo-db_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import ctypes as C  # noqa: E402

import oracle_bind  # noqa: E402
import tpch_data as T  # noqa: E402
from lingodb_amd import api, capi  # noqa: E402
from lingodb_amd.capi import AggSpec  # noqa: E402

D = capi.T_DECIMAL128
OPS = {"EQ": capi.F_EQ, "NEQ": capi.F_NEQ, "LT": capi.F_LT, "LTE": capi.F_LTE, "GT": capi.F_GT, "GTE": capi.F_GTE, "IN": capi.F_IN, "LIKE": capi.F_LIKE,
       "NOT LIKE": capi.F_NOT_LIKE}
KINDS = {"inner": capi.JOIN_INNER, "semi": capi.JOIN_SEMI, "anti": capi.JOIN_ANTI, "left_outer": capi.JOIN_LEFT_OUTER, "semi_build": capi.JOIN_SEMI_BUILD,
         "anti_build": capi.JOIN_ANTI_BUILD}
EPOCH_YEAR_DAYS = None


def days(s):
    import datetime

    return (datetime.date.fromisoformat(s) - datetime.date(1970, 1, 1)).days


def year_of(day_numbers):
    """civil year of date32 day numbers (DateRuntime::extractYear)"""
    return (np.asarray(day_numbers, dtype="int64").astype("datetime64[D]").astype("datetime64[Y]").astype(np.int64) + 1970)


class Frame:
    """late-materialised relation over host tables: sides = [(HostTable, rowids | None)]"""

    def __init__(self, legs, sides, n_rows=None):
        self.legs = legs
        self.rel = oracle_bind.HostRel(sides, n_rows)
        self.n = self.rel.n_rows

    def c(self, name):
        for s, (t, _) in enumerate(self.rel.sides):
            if name in t.arrow.column_names:
                return (s, t.arrow.column_names.index(name))
        raise KeyError(name)

    def pred(self, col, op, val):
        ref = self.c(col)
        if isinstance(val, tuple) and val[0] == "col":
            return api.pred(ref, OPS[op], rhs_col=self.c(val[1]))
        if op == "IN":
            return api.pred(ref, capi.F_IN, values=val)
        return api.pred(ref, OPS[op], val)

    def where(self, *preds):
        plist = [self.pred(*p) for p in preds]
        idx = self.legs.O.scan_filter(self.rel, plist, self.legs.threads)
        return self.take(idx)

    def take(self, idx):
        idx = np.asarray(idx, dtype=np.int64)
        sides = [(t, idx.astype(np.uint32) if r is None else r[idx]) for t, r in self.rel.sides]
        return Frame(self.legs, sides, len(idx))

    def join(self, build, on, kind="inner"):
        """self probes a hash table built on `build`; on = [(probe column, build column)]"""
        pk = [self.c(a) for a, _ in on]
        bk = [build.c(b) for _, b in on]
        op, ob, _ = self.legs.O.join(build.rel, bk, self.rel, pk, KINDS[kind], self.legs.threads)
        if kind in ("semi", "anti"):
            return self.take(op)
        if kind in ("semi_build", "anti_build"):
            return build.take(op)
        sides = [(t, self.rel.phys(s)[op]) for s, (t, _) in enumerate(self.rel.sides)]
        if kind == "left_outer":
            safe = np.where(ob == 0xFFFFFFFF, 0, ob)
            self.legs.last_unmatched = ob == 0xFFFFFFFF
            sides += [(t, build.rel.phys(s)[safe]) for s, (t, _) in enumerate(build.rel.sides)]
        else:
            sides += [(t, build.rel.phys(s)[ob]) for s, (t, _) in enumerate(build.rel.sides)]
        return Frame(self.legs, sides, len(op))

    # ---- values
    def np(self, name):
        """numeric column (ints, dates, decimals with p < 19 as unscaled int64, char(1) as int32) at the current rows"""
        s, ci = self.c(name)
        t, _ = self.rel.sides[s]
        return self.legs.numeric(t, ci)[self.rel.phys(s)]

    def strs(self, name):
        s, ci = self.c(name)
        t, _ = self.rel.sides[s]
        col = t.arrow.column(ci).combine_chunks()
        return np.array(col.take(pa.array(self.rel.phys(s))).to_pylist(), dtype=object)

    # ---- aggregation through the oracle's PreAggregationHashtable restatement
    def factor(self, spec):
        """'col' | int | ('1-', col) | ('1+', col)"""
        if isinstance(spec, int):
            return api.factor(spec, 0)
        if isinstance(spec, tuple):
            ref = self.c(spec[1])
            scale = self.legs.scale_of(self, spec[1])
            return api.factor(10**scale, -1 if spec[0] == "1-" else 1, ref)
        return api.factor(0, 1, self.c(spec))

    def expr(self, *terms):
        """terms: lists of factor specs; a leading '-' marks a subtracted term"""
        out = []
        for t in terms:
            neg = t[0] == "-"
            fs = t[1:] if neg else t
            out.append({"factors": [self.factor(f) for f in fs], "negate": neg})
        return api.expr(out)

    def groupby(self, keys, aggs, preds=()):
        """aggs: [(fn, expr | None, wide)] → (Frame of representative rows, [int64 arrays or python-int lists])"""
        specs = []
        for fn, e, wide in aggs:
            if fn == "count_star":
                specs.append(api.agg(capi.AGG_COUNT_STAR))
            else:
                specs.append(api.agg({"sum": capi.AGG_SUM, "min": capi.AGG_MIN, "max": capi.AGG_MAX}[fn], e, wide=wide, out_type=D, p=38 if wide else 18, s=0))
        plist = [self.pred(*p) for p in preds]
        krefs = [self.c(k) for k in keys]
        cap = max(self.n, 1)
        na = len(specs)
        rep = np.empty(cap, dtype=np.uint32)
        vals = np.empty((cap, max(na, 1), 2), dtype=np.int64)
        valid = np.empty((cap, max(na, 1)), dtype=np.uint8)
        aarr = (AggSpec * max(na, 1))()
        for i, a in enumerate(specs):
            aarr[i] = a
        O = self.legs.O
        g = O.lib.ora_groupby(C.byref(self.rel.struct), oracle_bind._preds(plist), len(plist), oracle_bind._refs(krefs), len(krefs), aarr, na, self.legs.threads,
                              rep.ctypes.data, vals.ctypes.data, valid.ctypes.data, cap)
        out = []
        for a in range(na):
            lo, hi = vals[:g, a, 0], vals[:g, a, 1]
            if np.array_equal(hi, lo >> 63):
                out.append(lo.copy())
            else:
                out.append(np.array([(int(h) << 64) | (int(l) & 0xFFFFFFFFFFFFFFFF) for l, h in zip(lo, hi)], dtype=object))
        return self.take(rep[:g]), out, valid[:g, :na]


class Legs:
    COLS = {  # columns each table is loaded with (all queries)
        T.LINEITEM: list(range(15)), T.ORDERS: list(range(8)), T.CUSTOMER: list(range(6)), T.PART: list(range(8)), T.SUPPLIER: list(range(7)),
        T.PARTSUPP: list(range(4)), T.NATION: [0, 1, 2], T.REGION: [0, 1]}

    def __init__(self, n_orders, threads=None, queries=None):
        self.n_orders = n_orders
        self.O = oracle_bind.load()
        self.threads = threads or self.O.num_cores()
        self._tables, self._numeric = {}, {}
        self.queries = queries

    NEED = {  # table → {query: columns}; only what the selected queries touch is generated
        T.LINEITEM: {1: [4, 5, 6, 7, 8, 9, 10], 3: [0, 5, 6, 10], 4: [0, 11, 12], 5: [0, 2, 5, 6], 6: [4, 5, 6, 10], 7: [0, 2, 5, 6, 10], 8: [0, 1, 2, 5, 6], 9: [0, 1, 2, 4, 5, 6],
                     10: [0, 5, 6, 8], 12: [0, 10, 11, 12, 14], 14: [1, 5, 6, 10], 15: [2, 5, 6, 10], 17: [1, 4, 5], 18: [0, 4], 19: [1, 4, 5, 6, 13, 14], 20: [1, 2, 4, 10],
                     21: [0, 2, 11, 12]},
        T.ORDERS: {3: [0, 1, 4, 6], 4: [0, 4, 5], 5: [0, 1, 4], 7: [0, 1], 8: [0, 1, 4], 9: [0, 4], 10: [0, 1, 4], 12: [0, 5], 13: [1, 7], 18: [0, 1, 3, 4], 21: [0, 2], 22: [1]},
        T.CUSTOMER: {3: [0, 3], 5: [0, 1], 7: [0, 1], 8: [0, 1], 10: [0, 1, 2, 4], 13: [0], 18: [0, 4], 22: [0, 2, 5]},
        T.PART: {2: [0, 1, 4, 7], 8: [0, 4], 9: [0, 3], 14: [0, 4], 16: [0, 1, 4, 5], 17: [0, 5, 6], 19: [0, 1, 5, 6], 20: [0, 3]},
        T.SUPPLIER: {2: [0, 1, 2, 3, 4, 5, 6], 5: [0, 1], 7: [0, 1], 8: [0, 1], 9: [0, 1], 11: [0, 1], 15: [0, 1], 16: [0, 6], 20: [0, 1, 3, 4], 21: [0, 1, 3]},
        T.PARTSUPP: {2: [0, 1, 3], 9: [0, 1, 3], 11: [0, 1, 2, 3], 16: [0, 1], 20: [0, 1, 2]},
        T.NATION: {q: [0, 1, 2] for q in (2, 5, 7, 8, 9, 10, 11, 20, 21)},
        T.REGION: {2: [0, 1], 5: [0, 1], 8: [0, 1]},
    }

    def table(self, tid):
        if tid not in self._tables:
            qs = self.queries or list(range(1, 23))
            cols = sorted({c for q in qs for c in self.NEED[tid].get(q, [])}) or [0]
            self._tables[tid] = oracle_bind.HostTable(T.host_table(tid, self.n_orders, cols=cols))
        return self._tables[tid]

    def frame(self, tid):
        return Frame(self, [(self.table(tid), None)])

    def numeric(self, t, ci):
        key = (id(t), ci)
        if key not in self._numeric:
            col = t.arrow.column(ci).combine_chunks()
            ty = col.type
            n = len(col)
            if pa.types.is_decimal(ty):
                v = np.frombuffer(col.buffers()[1], dtype=np.int64)[::2][col.offset : col.offset + n].copy()
            elif pa.types.is_fixed_size_binary(ty) or pa.types.is_int32(ty) or pa.types.is_date32(ty):
                v = np.frombuffer(col.buffers()[1], dtype=np.int32)[col.offset : col.offset + n].astype(np.int64)
            elif pa.types.is_int64(ty):
                v = np.frombuffer(col.buffers()[1], dtype=np.int64)[col.offset : col.offset + n].copy()
            else:
                raise TypeError(f"numeric view of {ty}")
            self._numeric[key] = v
        return self._numeric[key]

    def scale_of(self, frame, name):
        s, ci = frame.c(name)
        ty = frame.rel.sides[s][0].arrow.schema.field(ci).type
        return ty.scale if pa.types.is_decimal(ty) else 0

    def run(self, q):
        return getattr(self, "q%d" % q)()

    # ------------------------------------------------------------------ the queries
    def revenue(self, f, ext="l_extendedprice", disc="l_discount"):
        return ("sum", f.expr([ext, ("1-", disc)]), True)  # decimal(12,2) x decimal(21,2) → decimal(33,4), 128-bit accumulator

    def q1_partials(self):
        """{(returnflag, linestatus): [Σ qty, Σ price, Σ disc_price, Σ charge, Σ discount, count]} — sums and counts add up over any
        partition of lineitem (bench.py's sliced oracle at the bench's own scale adds the slices' partials before the averages)"""
        li = self.frame(T.LINEITEM)
        aggs = [("sum", li.expr(["l_quantity"]), False), ("sum", li.expr(["l_extendedprice"]), False), self.revenue(li),
                ("sum", li.expr(["l_extendedprice", ("1-", "l_discount"), ("1+", "l_tax")]), True), ("sum", li.expr(["l_discount"]), False), ("count_star", None, False)]
        g, cols, _ = li.groupby(["l_returnflag", "l_linestatus"], aggs, [("l_shipdate", "LTE", days("1998-09-02"))])
        rf, ls = g.np("l_returnflag"), g.np("l_linestatus")
        return {(int(rf[i]), int(ls[i])): [int(c[i]) for c in cols] for i in range(g.n)}

    @staticmethod
    def q1_finish(partials):
        rows = []
        for (rf, ls), (sq, sb, sd, sc, sdisc, c) in partials.items():
            avg = lambda s: (int(s) * 10**19) // c  # AVG = (SUM * 10^19) sdiv COUNT → decimal(31,21)
            rows.append((rf, ls, sq, sb, sd, sc, avg(sq), avg(sb), avg(sdisc), c))
        return sorted(rows)

    def q1(self):
        return self.q1_finish(self.q1_partials())

    def q6(self):
        li = self.frame(T.LINEITEM)
        p = [("l_shipdate", "GTE", days("1994-01-01")), ("l_shipdate", "LT", days("1995-01-01")), ("l_discount", "GTE", 5), ("l_discount", "LTE", 7), ("l_quantity", "LT", 2400)]
        _, (s,), valid = li.groupby([], [("sum", li.expr(["l_extendedprice", "l_discount"]), True)], p)
        return [(int(s[0]) if valid[0][0] else None,)]

    def q3(self):
        cu = self.frame(T.CUSTOMER).where(("c_mktsegment", "EQ", "BUILDING"))
        od = self.frame(T.ORDERS).where(("o_orderdate", "LT", days("1995-03-15")))
        li = self.frame(T.LINEITEM).where(("l_shipdate", "GT", days("1995-03-15")))
        co = od.join(cu, [("o_custkey", "c_custkey")])
        lco = li.join(co, [("l_orderkey", "o_orderkey")])
        g, (rev,), _ = lco.groupby(["l_orderkey", "o_orderdate", "o_shippriority"], [self.revenue(lco)])
        rows = list(zip(g.np("l_orderkey").tolist(), [int(r) for r in rev], g.np("o_orderdate").tolist(), g.np("o_shippriority").tolist()))
        return sorted(rows, key=lambda r: (-r[1], r[2]))  # the caller applies LIMIT 10 (ties beyond the ORDER BY keys are unspecified)

    def q4(self):
        od = self.frame(T.ORDERS).where(("o_orderdate", "GTE", days("1993-07-01")), ("o_orderdate", "LT", days("1993-10-01")))
        li = self.frame(T.LINEITEM).where(("l_commitdate", "LT", ("col", "l_receiptdate")))
        keep = li.join(od, [("l_orderkey", "o_orderkey")], "semi_build")
        g, (cnt,), _ = keep.groupby(["o_orderpriority"], [("count_star", None, False)])
        return sorted(zip(g.strs("o_orderpriority").tolist(), [int(c) for c in cnt]))

    def region_nations(self, region):
        re_ = self.frame(T.REGION).where(("r_name", "EQ", region))
        return self.frame(T.NATION).join(re_, [("n_regionkey", "r_regionkey")], "semi")

    def q5(self):
        na = self.region_nations("ASIA")
        cu = self.frame(T.CUSTOMER).join(na, [("c_nationkey", "n_nationkey")])
        su = self.frame(T.SUPPLIER).join(na, [("s_nationkey", "n_nationkey")], "semi")
        od = self.frame(T.ORDERS).where(("o_orderdate", "GTE", days("1994-01-01")), ("o_orderdate", "LT", days("1995-01-01")))
        oc = od.join(cu, [("o_custkey", "c_custkey")])
        lo = self.frame(T.LINEITEM).join(oc, [("l_orderkey", "o_orderkey")])
        ls = lo.join(su, [("l_suppkey", "s_suppkey"), ("c_nationkey", "s_nationkey")])
        g, (rev,), _ = ls.groupby(["n_name"], [self.revenue(ls)])
        return sorted(zip(g.strs("n_name").tolist(), [int(r) for r in rev]), key=lambda r: -r[1])

    def q7(self):
        na = self.frame(T.NATION).where(("n_name", "IN", ["FRANCE", "GERMANY"]))
        nname = dict(zip(na.np("n_nationkey").tolist(), na.strs("n_name").tolist()))
        cu = self.frame(T.CUSTOMER).join(na, [("c_nationkey", "n_nationkey")], "semi")
        su = self.frame(T.SUPPLIER).join(na, [("s_nationkey", "n_nationkey")], "semi")
        li = self.frame(T.LINEITEM).where(("l_shipdate", "GTE", days("1995-01-01")), ("l_shipdate", "LTE", days("1996-12-31")))
        ls = li.join(su, [("l_suppkey", "s_suppkey")])
        lso = ls.join(self.frame(T.ORDERS), [("l_orderkey", "o_orderkey")])
        all_ = lso.join(cu, [("o_custkey", "c_custkey")])
        sn, cn = all_.np("s_nationkey"), all_.np("c_nationkey")
        d = all_.take(np.nonzero(sn != cn)[0])
        key = d.np("s_nationkey") * 100000 + d.np("c_nationkey") * 10000 + year_of(d.np("l_shipdate"))
        vol = d.np("l_extendedprice") * (100 - d.np("l_discount"))
        uk, inv = np.unique(key, return_inverse=True)
        sums = np.zeros(len(uk), dtype=np.int64)
        np.add.at(sums, inv, vol)
        return sorted((nname[int(k) // 100000], nname[int(k) // 10000 % 10], int(k) % 10000, int(s)) for k, s in zip(uk, sums))

    def q8(self):
        pa_ = self.frame(T.PART).where(("p_type", "EQ", "ECONOMY ANODIZED STEEL"))
        lp = self.frame(T.LINEITEM).join(pa_, [("l_partkey", "p_partkey")], "semi")
        na = self.region_nations("AMERICA")
        cu = self.frame(T.CUSTOMER).join(na, [("c_nationkey", "n_nationkey")], "semi")
        od = self.frame(T.ORDERS).where(("o_orderdate", "GTE", days("1995-01-01")), ("o_orderdate", "LTE", days("1996-12-31"))).join(cu, [("o_custkey", "c_custkey")], "semi")
        lo = lp.join(od, [("l_orderkey", "o_orderkey")])
        ls = lo.join(self.frame(T.SUPPLIER), [("l_suppkey", "s_suppkey")])
        brazil = int(self.frame(T.NATION).where(("n_name", "EQ", "BRAZIL")).np("n_nationkey")[0])
        yr = year_of(ls.np("o_orderdate"))
        vol = ls.np("l_extendedprice") * (100 - ls.np("l_discount"))
        out = []
        for y in sorted(set(yr.tolist())):
            m = yr == y
            num, den = int(vol[m & (ls.np("s_nationkey") == brazil)].sum()), int(vol[m].sum())
            out.append((int(y), num * 10**6 // den))  # decimal(38,6): (num * 10^6) sdiv den
        return out

    def q9_partials(self):
        """{(s_nationkey, year): Σ amount} — the sums add up over any partition of lineitem that keeps an order's lines with its order row (bench.py's
        sliced oracle at the bench's own scale; tpch_plans.oracle_q9_at_scale hands in part / partsupp already reduced to the '%green%' parts — the
        semi-join reduction leaves every join below unchanged)"""
        pt = self.frame(T.PART).where(("p_name", "LIKE", "%green%"))
        lp = self.frame(T.LINEITEM).join(pt, [("l_partkey", "p_partkey")], "semi")
        lps = lp.join(self.frame(T.PARTSUPP), [("l_partkey", "ps_partkey"), ("l_suppkey", "ps_suppkey")])
        ls = lps.join(self.frame(T.SUPPLIER), [("l_suppkey", "s_suppkey")])
        lo = ls.join(self.frame(T.ORDERS), [("l_orderkey", "o_orderkey")])
        amount = lo.np("l_extendedprice") * (100 - lo.np("l_discount")) - lo.np("ps_supplycost") * lo.np("l_quantity")
        key = lo.np("s_nationkey") * 10000 + year_of(lo.np("o_orderdate"))
        uk, inv = np.unique(key, return_inverse=True)
        sums = np.zeros(len(uk), dtype=np.int64)
        np.add.at(sums, inv, amount)
        return {(int(k) // 10000, int(k) % 10000): int(s) for k, s in zip(uk, sums)}

    def q9_finish(self, partials):
        na = self.frame(T.NATION)
        nname = dict(zip(na.np("n_nationkey").tolist(), na.strs("n_name").tolist()))
        return sorted(((nname[nk], yr, int(s)) for (nk, yr), s in partials.items()), key=lambda r: (r[0], -r[1]))

    def q9(self):
        return self.q9_finish(self.q9_partials())

    def q10(self):
        od = self.frame(T.ORDERS).where(("o_orderdate", "GTE", days("1993-10-01")), ("o_orderdate", "LT", days("1994-01-01")))
        li = self.frame(T.LINEITEM).where(("l_returnflag", "EQ", int.from_bytes(b"R\0\0\0", "little")))
        lo = li.join(od, [("l_orderkey", "o_orderkey")])
        g, (rev,), _ = lo.groupby(["o_custkey"], [self.revenue(lo)])
        ck = g.np("o_custkey")
        order = np.lexsort((ck, -rev.astype(np.int64)))[:64]
        cu, na = self.frame(T.CUSTOMER), self.frame(T.NATION)
        nname = dict(zip(na.np("n_nationkey").tolist(), na.strs("n_name").tolist()))
        ckeys = cu.np("c_custkey")
        rows = []
        for i in order:
            r = int(np.searchsorted(ckeys, ck[i]))
            one = cu.take([r])
            rows.append((int(ck[i]), one.strs("c_name")[0], int(rev[i]), int(one.np("c_acctbal")[0]), nname[int(one.np("c_nationkey")[0])]))
        return rows  # ORDER BY revenue DESC; the caller applies LIMIT 20

    def q11(self):
        na = self.frame(T.NATION).where(("n_name", "EQ", "GERMANY"))
        su = self.frame(T.SUPPLIER).join(na, [("s_nationkey", "n_nationkey")], "semi")
        ps = self.frame(T.PARTSUPP).join(su, [("ps_suppkey", "s_suppkey")], "semi")
        g, (val,), _ = ps.groupby(["ps_partkey"], [("sum", ps.expr(["ps_supplycost", "ps_availqty"]), True)])
        total = int(val.sum()) if val.dtype != object else sum(int(v) for v in val)
        rows = [(int(k), int(v)) for k, v in zip(g.np("ps_partkey"), val) if int(v) * 10_000 > total]  # value > total * 0.0001 at the common scale 6
        return sorted(rows, key=lambda r: -r[1])

    def q12(self):
        li = self.frame(T.LINEITEM).where(("l_receiptdate", "GTE", days("1994-01-01")), ("l_receiptdate", "LT", days("1995-01-01")), ("l_commitdate", "LT", ("col", "l_receiptdate")),
                                          ("l_shipdate", "LT", ("col", "l_commitdate")), ("l_shipmode", "IN", ["MAIL", "SHIP"]))
        lo = li.join(self.frame(T.ORDERS), [("l_orderkey", "o_orderkey")])
        mode, prio = lo.strs("l_shipmode"), lo.strs("o_orderpriority")
        high = np.isin(prio, ["1-URGENT", "2-HIGH"])
        return [(m, int((high & (mode == m)).sum()), int((~high & (mode == m)).sum())) for m in sorted(set(mode.tolist()))]

    def q13(self):
        od = self.frame(T.ORDERS).where(("o_comment", "NOT LIKE", "%special%requests%"))
        g, (cnt,), _ = od.groupby(["o_custkey"], [("count_star", None, False)])
        n_cust = self.frame(T.CUSTOMER).n
        dist = np.bincount(cnt.astype(np.int64))
        dist[0] += n_cust - g.n  # customers without a (qualifying) order: the NULL-extended rows of the outer join count 0
        return sorted(((int(c), int(n)) for c, n in enumerate(dist) if n), key=lambda r: (-r[1], -r[0]))

    def q14(self):
        li = self.frame(T.LINEITEM).where(("l_shipdate", "GTE", days("1995-09-01")), ("l_shipdate", "LT", days("1995-10-01")))
        promo = self.frame(T.PART).where(("p_type", "LIKE", "PROMO%"))
        lp = li.join(promo, [("l_partkey", "p_partkey")], "semi")
        rev = lambda f: int((f.np("l_extendedprice") * (100 - f.np("l_discount"))).sum())
        a, b = rev(lp), rev(li)
        return [((a * 10000) * 10**4 // b,)]  # 100.00 * a / b: decimal(38,6), truncating

    def q15(self):
        li = self.frame(T.LINEITEM)
        g, (rev,), _ = li.groupby(["l_suppkey"], [self.revenue(li)], [("l_shipdate", "GTE", days("1996-01-01")), ("l_shipdate", "LT", days("1996-04-01"))])
        if g.n == 0:
            return []
        best = max(int(r) for r in rev)
        return sorted((int(k), int(r)) for k, r in zip(g.np("l_suppkey"), rev) if int(r) == best)

    def q16(self):
        bad = self.frame(T.SUPPLIER).where(("s_comment", "LIKE", "%Customer%Complaints%"))
        pt = self.frame(T.PART).where(("p_brand", "NEQ", "Brand#45"), ("p_type", "NOT LIKE", "MEDIUM POLISHED%"), ("p_size", "IN", [49, 14, 23, 45, 19, 3, 36, 9]))
        pp = self.frame(T.PARTSUPP).join(pt, [("ps_partkey", "p_partkey")]).join(bad, [("ps_suppkey", "s_suppkey")], "anti")
        d, _, _ = pp.groupby(["p_brand", "p_type", "p_size", "ps_suppkey"], [("count_star", None, False)])
        g, (cnt,), _ = d.groupby(["p_brand", "p_type", "p_size"], [("count_star", None, False)])
        rows = zip(g.strs("p_brand").tolist(), g.strs("p_type").tolist(), g.np("p_size").tolist(), [int(c) for c in cnt])
        return sorted(rows, key=lambda r: (-r[3], r[0], r[1], r[2]))

    def q17(self):
        pt = self.frame(T.PART).where(("p_brand", "EQ", "Brand#23"), ("p_container", "EQ", "MED BOX"))
        l1 = self.frame(T.LINEITEM).join(pt, [("l_partkey", "p_partkey")], "semi")
        g, (sq, cnt), _ = l1.groupby(["l_partkey"], [("sum", l1.expr(["l_quantity"]), False), ("count_star", None, False)])
        avg21 = {int(k): (int(s) * 10**19) // int(c) for k, s, c in zip(g.np("l_partkey"), sq, cnt)}
        keep = [int(q) * 10**20 < 2 * avg21[int(k)] for q, k in zip(l1.np("l_quantity"), l1.np("l_partkey"))]  # l_quantity < 0.2 * avg at decimal(33,22)
        s = int(l1.np("l_extendedprice")[np.array(keep, dtype=bool)].sum()) if any(keep) else None
        return [(None if s is None else (s * 10**5) // 70,)]  # sum / 7.0 → decimal(17,6)

    def q18_big(self):
        """[(o_custkey, o_orderkey, o_orderdate, o_totalprice, Σ l_quantity)] of the orders with Σ l_quantity > 300 — an order's lines never leave its
        order-range slice, so the lists of any such partition of (orders, lineitem) concatenate (tpch_plans.oracle_q18_at_scale)"""
        li = self.frame(T.LINEITEM)
        g, (sq,), _ = li.groupby(["l_orderkey"], [("sum", li.expr(["l_quantity"]), False)])
        big = g.take(np.nonzero(sq > 300 * 100)[0])
        bigq = dict(zip(big.np("l_orderkey").tolist(), sq[sq > 300 * 100].tolist()))
        od = self.frame(T.ORDERS).join(big, [("o_orderkey", "l_orderkey")], "semi")
        return [(int(ck), int(ok), int(d), int(tp), int(bigq[int(ok)])) for ck, ok, d, tp in
                zip(od.np("o_custkey"), od.np("o_orderkey"), od.np("o_orderdate"), od.np("o_totalprice"))]

    def q18_finish(self, big_rows):
        """the customer join of the big orders (hash table on customer, probed by the few surviving orders) + ORDER BY"""
        if not big_rows:
            return []
        oc_t = oracle_bind.HostTable(pa.table({"b_custkey": pa.array([r[0] for r in big_rows], type=pa.int32())}))
        oc = Frame(self, [(oc_t, None)]).join(self.frame(T.CUSTOMER), [("b_custkey", "c_custkey")])
        src = oc.rel.phys(0)  # the row of big_rows each joined row comes from
        rows = [(nm, int(ck)) + tuple(big_rows[int(i)][1:]) for nm, ck, i in zip(oc.strs("c_name"), oc.np("c_custkey"), src)]
        return sorted(rows, key=lambda r: (-r[4], r[3]))  # the caller applies LIMIT 100

    def q18(self):
        return self.q18_finish(self.q18_big())

    def q19(self):
        li = self.frame(T.LINEITEM).where(("l_shipmode", "IN", ["AIR", "AIR REG"]), ("l_shipinstruct", "EQ", "DELIVER IN PERSON"))
        pt = self.frame(T.PART).where(("p_brand", "IN", ["Brand#12", "Brand#23", "Brand#34"]), ("p_size", "GTE", 1), ("p_size", "LTE", 15))
        lp = li.join(pt, [("l_partkey", "p_partkey")])
        brand, cont, size, qty = lp.strs("p_brand"), lp.strs("p_container"), lp.np("p_size"), lp.np("l_quantity")
        alt = lambda b, cs, lo, hi, smax: (brand == b) & np.isin(cont, cs) & (qty >= lo * 100) & (qty <= hi * 100) & (size <= smax)
        m = alt("Brand#12", ["SM CASE", "SM BOX", "SM PACK", "SM PKG"], 1, 11, 5) | alt("Brand#23", ["MED BAG", "MED BOX", "MED PKG", "MED PACK"], 10, 20, 10) | \
            alt("Brand#34", ["LG CASE", "LG BOX", "LG PACK", "LG PKG"], 20, 30, 15)
        if not m.any():
            return [(None,)]
        return [(int((lp.np("l_extendedprice")[m] * (100 - lp.np("l_discount")[m])).sum()),)]

    def q20(self):
        pt = self.frame(T.PART).where(("p_name", "LIKE", "forest%"))
        ps = self.frame(T.PARTSUPP).join(pt, [("ps_partkey", "p_partkey")], "semi")
        li = self.frame(T.LINEITEM).where(("l_shipdate", "GTE", days("1994-01-01")), ("l_shipdate", "LT", days("1995-01-01"))).join(pt, [("l_partkey", "p_partkey")], "semi")
        g, (sq,), _ = li.groupby(["l_partkey", "l_suppkey"], [("sum", li.expr(["l_quantity"]), False)])
        qty = dict(zip(zip(g.np("l_partkey").tolist(), g.np("l_suppkey").tolist()), sq.tolist()))
        ok = {int(sk) for pk, sk, av in zip(ps.np("ps_partkey"), ps.np("ps_suppkey"), ps.np("ps_availqty")) if (int(pk), int(sk)) in qty and int(av) * 1000 > 5 * qty[(int(pk), int(sk))]}
        na = self.frame(T.NATION).where(("n_name", "EQ", "CANADA"))
        su = self.frame(T.SUPPLIER).join(na, [("s_nationkey", "n_nationkey")], "semi")
        keep = su.take(np.nonzero(np.isin(su.np("s_suppkey"), list(ok)))[0])
        return sorted(zip(keep.strs("s_name").tolist(), keep.strs("s_address").tolist()))

    def q21(self):
        na = self.frame(T.NATION).where(("n_name", "EQ", "SAUDI ARABIA"))
        su = self.frame(T.SUPPLIER).join(na, [("s_nationkey", "n_nationkey")], "semi")
        late = self.frame(T.LINEITEM).where(("l_receiptdate", "GT", ("col", "l_commitdate")))
        l1 = late.join(su, [("l_suppkey", "s_suppkey")])
        of = self.frame(T.ORDERS).where(("o_orderstatus", "EQ", int.from_bytes(b"F\0\0\0", "little")))
        l1 = l1.join(of, [("l_orderkey", "o_orderkey")], "semi")
        # EXISTS / NOT EXISTS with l_suppkey <> l1.l_suppkey: join the candidates with the lineitems of their orders, residual in numpy
        cand = Frame(self, [(l1.rel.sides[0][0], l1.rel.phys(0))], l1.n)  # the l1 rows as a lineitem-only relation
        def others(probe):
            pairs = probe.join(cand, [("l_orderkey", "l_orderkey")])  # sides: probe lineitem, candidate lineitem
            t = pairs.rel.sides[0][0]
            sk = self.numeric(t, t.arrow.column_names.index("l_suppkey"))
            differs = sk[pairs.rel.phys(0)] != sk[pairs.rel.phys(1)]
            return set(pairs.rel.phys(1)[differs].tolist())  # physical rows of candidates that have a partner with another supplier
        has_other = others(self.frame(T.LINEITEM))
        has_other_late = others(late)
        phys = l1.rel.phys(0)
        keep = np.array([int(p) in has_other and int(p) not in has_other_late for p in phys], dtype=bool)
        names = l1.strs("s_name")[keep]
        uk, cnt = np.unique(names, return_counts=True)
        return sorted(zip(uk.tolist(), cnt.tolist()), key=lambda r: (-r[1], r[0]))[:100]

    def q22(self):
        cu = self.frame(T.CUSTOMER)
        code = np.array([p[:2] for p in cu.strs("c_phone")], dtype=object)
        bal = cu.np("c_acctbal")
        in_codes = np.isin(code, ["13", "31", "23", "29", "30", "18", "17"])
        pos = in_codes & (bal > 0)
        avg = (int(bal[pos].sum()) * 10**19) // int(pos.sum())
        cand = cu.take(np.nonzero(in_codes & np.array([int(b) * 10**19 > avg for b in bal], dtype=bool))[0])
        keep = self.frame(T.ORDERS).join(cand, [("o_custkey", "c_custkey")], "anti_build")
        kc = np.array([p[:2] for p in keep.strs("c_phone")], dtype=object)
        return [(c, int((kc == c).sum()), int(keep.np("c_acctbal")[kc == c].sum())) for c in sorted(set(kc.tolist()))]

    def q2(self):
        na = self.region_nations("EUROPE")
        sn = self.frame(T.SUPPLIER).join(na, [("s_nationkey", "n_nationkey")])
        pt = self.frame(T.PART).where(("p_size", "EQ", 15), ("p_type", "LIKE", "%BRASS"))
        all_ = self.frame(T.PARTSUPP).join(pt, [("ps_partkey", "p_partkey")]).join(sn, [("ps_suppkey", "s_suppkey")])
        g, (mn,), _ = all_.groupby(["ps_partkey"], [("min", all_.expr(["ps_supplycost"]), False)])
        mins = dict(zip(g.np("ps_partkey").tolist(), mn.tolist()))
        best = all_.take(np.nonzero(all_.np("ps_supplycost") == np.array([mins[int(k)] for k in all_.np("ps_partkey")]))[0])
        rows = list(zip(best.np("s_acctbal").tolist(), best.strs("s_name").tolist(), best.strs("n_name").tolist(), best.np("p_partkey").tolist(), best.strs("p_mfgr").tolist(),
                        best.strs("s_address").tolist(), best.strs("s_phone").tolist(), best.strs("s_comment").tolist()))
        return sorted(rows, key=lambda r: (-r[0], r[2], r[1], r[3]))[:100]
