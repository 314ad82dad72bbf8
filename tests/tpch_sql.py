"""Independent evaluations of the reference's SQL text (resources/sql/tpch/*.sql) with pandas / Python
integers over the generated tables — what both the GPU plans (tests/test_gpu_tpch_new.py) and the
oracle legs (tests/test_oracle_legs.py) are held to.  Decimals are unscaled integers."""
import datetime
import re

import numpy as np
import pandas as pd
import pyarrow as pa

import tpch_data as T

EPOCH = datetime.date(1970, 1, 1)


def days(s):
    return (datetime.date.fromisoformat(s) - EPOCH).days


def frame(table):
    """pyarrow table → pandas with decimals as unscaled int64, dates as day numbers, char(1) as str"""
    cols = {}
    for name in table.column_names:
        col = table.column(name).combine_chunks()
        t = col.type
        if pa.types.is_decimal(t):
            cols[name] = np.frombuffer(col.buffers()[1], dtype=np.int64)[::2][col.offset : col.offset + len(col)].copy()
        elif pa.types.is_date32(t) or pa.types.is_int32(t):
            cols[name] = np.frombuffer(col.buffers()[1], dtype=np.int32)[col.offset : col.offset + len(col)].astype(np.int64)
        elif pa.types.is_fixed_size_binary(t):
            cols[name] = [v.as_py().rstrip(b"\0").decode() for v in col]
        else:
            cols[name] = col.to_pylist()
    return pd.DataFrame(cols)


def like(series, *fragments):
    """SQL LIKE '%f1%f2%…'"""
    rx = re.compile("^.*" + ".*".join(re.escape(f) for f in fragments) + ".*$", re.S)
    return series.map(lambda s: rx.match(s) is not None)


class Tables:
    """generated host tables as data frames, cached per (table, columns)"""

    def __init__(self, n_orders):
        self.n, self.host, self.df = n_orders, {}, {}

    def arrow(self, tid, cols):
        key = (tid, tuple(cols))
        if key not in self.host:
            self.host[key] = T.host_table(tid, self.n, cols=cols)
        return self.host[key]

    def get(self, tid, cols):
        key = (tid, tuple(cols))
        if key not in self.df:
            self.df[key] = frame(self.arrow(tid, cols))
        return self.df[key]


# columns each evaluation reads: {query: {table name: (table id, columns)}}
INPUTS = {
    19: {"lineitem": (T.LINEITEM, [1, 4, 5, 6, 13, 14]), "part": (T.PART, [0, 1, 5, 6])},
    22: {"customer": (T.CUSTOMER, [0, 2, 5]), "orders": (T.ORDERS, [0, 1])},
    13: {"customer": (T.CUSTOMER, [0]), "orders": (T.ORDERS, [0, 1, 7])},
    16: {"part": (T.PART, [0, 1, 4, 5]), "partsupp": (T.PARTSUPP, [0, 1]), "supplier": (T.SUPPLIER, [0, 6])},
    17: {"lineitem": (T.LINEITEM, [1, 4, 5, 6, 13, 14]), "part": (T.PART, [0, 1, 5, 6])},
    20: {"lineitem": (T.LINEITEM, [1, 2, 4, 10]), "part": (T.PART, [0, 3]), "partsupp": (T.PARTSUPP, [0, 1, 2]), "supplier": (T.SUPPLIER, [0, 1, 3, 4]), "nation": (T.NATION, [0, 1, 2])},
    21: {"lineitem": (T.LINEITEM, [0, 2, 11, 12]), "orders": (T.ORDERS, [0, 2]), "supplier": (T.SUPPLIER, [0, 1, 3, 4]), "nation": (T.NATION, [0, 1, 2])},
    2: {"part": (T.PART, [0, 1, 4, 7]), "partsupp": (T.PARTSUPP, [0, 1, 3]), "supplier": (T.SUPPLIER, [0, 1, 2, 3, 4, 5, 6]), "nation": (T.NATION, [0, 1, 2]), "region": (T.REGION, [0, 1])},
}


def q19(t):
    li, pa_ = t["lineitem"], t["part"]
    j = li.merge(pa_, left_on="l_partkey", right_on="p_partkey")
    common = j.l_shipmode.isin(["AIR", "AIR REG"]) & (j.l_shipinstruct == "DELIVER IN PERSON")
    c1 = (j.p_brand == "Brand#12") & j.p_container.isin(["SM CASE", "SM BOX", "SM PACK", "SM PKG"]) & (j.l_quantity >= 100) & (j.l_quantity <= 1100) & j.p_size.between(1, 5)
    c2 = (j.p_brand == "Brand#23") & j.p_container.isin(["MED BAG", "MED BOX", "MED PKG", "MED PACK"]) & (j.l_quantity >= 1000) & (j.l_quantity <= 2000) & j.p_size.between(1, 10)
    c3 = (j.p_brand == "Brand#34") & j.p_container.isin(["LG CASE", "LG BOX", "LG PACK", "LG PKG"]) & (j.l_quantity >= 2000) & (j.l_quantity <= 3000) & j.p_size.between(1, 15)
    sel = j[common & (c1 | c2 | c3)]
    return [(int((sel.l_extendedprice * (100 - sel.l_discount)).sum()) if len(sel) else None,)]


def q22(t):
    cu, od = t["customer"], t["orders"]
    cu = cu.assign(cntrycode=cu.c_phone.str[:2])
    c2 = cu[cu.cntrycode.isin(["13", "31", "23", "29", "30", "18", "17"])]
    pos = c2[c2.c_acctbal > 0]
    avg = (int(pos.c_acctbal.sum()) * 10**19) // len(pos)  # decimal(31,21): (sum * 10^19) sdiv count
    c4 = c2[c2.c_acctbal.map(lambda v: int(v) * 10**19 > avg)]
    c5 = c4[~c4.c_custkey.isin(set(od.o_custkey.tolist()))]
    g = c5.groupby("cntrycode").agg(numcust=("c_custkey", "size"), tot=("c_acctbal", "sum")).reset_index().sort_values("cntrycode")
    return [(r.cntrycode, int(r.numcust), int(r.tot)) for r in g.itertuples()]


def q13(t):
    cu, od = t["customer"], t["orders"]
    keep = od[~like(od.o_comment, "special", "requests")]
    cnt = keep.groupby("o_custkey").size()
    c_count = cu.c_custkey.map(cnt).fillna(0).astype(np.int64)
    return sorted(((int(c), int(n)) for c, n in c_count.value_counts().items()), key=lambda r: (-r[1], -r[0]))


def q16(t):
    pa_, ps, su = t["part"], t["partsupp"], t["supplier"]
    bad = set(su[like(su.s_comment, "Customer", "Complaints")].s_suppkey.tolist())
    p1 = pa_[(pa_.p_brand != "Brand#45") & ~pa_.p_type.str.startswith("MEDIUM POLISHED") & pa_.p_size.isin([49, 14, 23, 45, 19, 3, 36, 9])]
    j = ps.merge(p1, left_on="ps_partkey", right_on="p_partkey")
    j = j[~j.ps_suppkey.isin(bad)]
    g = j.groupby(["p_brand", "p_type", "p_size"]).ps_suppkey.nunique().reset_index(name="cnt")
    return sorted(((r.p_brand, r.p_type, int(r.p_size), int(r.cnt)) for r in g.itertuples()), key=lambda r: (-r[3], r[0], r[1], r[2]))


def q17(t):
    li, pa_ = t["lineitem"], t["part"]
    keys = set(pa_[(pa_.p_brand == "Brand#23") & (pa_.p_container == "MED BOX")].p_partkey.tolist())
    l1 = li[li.l_partkey.isin(keys)]
    stats = l1.groupby("l_partkey").l_quantity.agg(["sum", "size"])
    avg21 = {k: (int(r["sum"]) * 10**19) // int(r["size"]) for k, r in stats.iterrows()}  # avg(l_quantity): decimal(31,21)
    # l_quantity < 0.2 * avg: 0.2 is decimal(2,1), the product decimal(33,22); l_quantity is cast to it (x 10^20)
    small = l1[[int(q) * 10**20 < 2 * avg21[k] for q, k in zip(l1.l_quantity, l1.l_partkey)]]
    return [((int(small.l_extendedprice.sum()) * 10**5) // 70 if len(small) else None,)]  # sum / 7.0 → decimal(17,6)


def q20(t):
    li, pa_, ps, su, na = t["lineitem"], t["part"], t["partsupp"], t["supplier"], t["nation"]
    forest = set(pa_[pa_.p_name.str.startswith("forest")].p_partkey.tolist())
    l1 = li[(li.l_shipdate >= days("1994-01-01")) & (li.l_shipdate < days("1995-01-01")) & li.l_partkey.isin(forest)]
    qty = l1.groupby(["l_partkey", "l_suppkey"]).l_quantity.sum().to_dict()
    ps1 = ps[ps.ps_partkey.isin(forest)]
    ok = [sk for pk, sk, av in zip(ps1.ps_partkey, ps1.ps_suppkey, ps1.ps_availqty) if (pk, sk) in qty and int(av) * 1000 > 5 * int(qty[(pk, sk)])]
    canada = set(na[na.n_name == "CANADA"].n_nationkey.tolist())
    s = su[su.s_nationkey.isin(canada) & su.s_suppkey.isin(set(ok))].sort_values("s_name")
    return [(r.s_name, r.s_address) for r in s.itertuples()]


def q21(t):
    li, od, su, na = t["lineitem"], t["orders"], t["supplier"], t["nation"]
    saudi = set(na[na.n_name == "SAUDI ARABIA"].n_nationkey.tolist())
    sname = {k: n for k, n, nk in zip(su.s_suppkey, su.s_name, su.s_nationkey) if nk in saudi}
    forders = set(od[od.o_orderstatus == "F"].o_orderkey.tolist())
    by_order_all, by_order_late = {}, {}
    late = li.l_receiptdate > li.l_commitdate
    for ok, sk, lt in zip(li.l_orderkey.tolist(), li.l_suppkey.tolist(), late.tolist()):
        by_order_all.setdefault(ok, set()).add(sk)
        if lt:
            by_order_late.setdefault(ok, set()).add(sk)
    cnt = {}
    for ok, sk, lt in zip(li.l_orderkey.tolist(), li.l_suppkey.tolist(), late.tolist()):
        if lt and sk in sname and ok in forders and len(by_order_all[ok] - {sk}) > 0 and len(by_order_late[ok] - {sk}) == 0:
            cnt[sname[sk]] = cnt.get(sname[sk], 0) + 1
    return sorted(cnt.items(), key=lambda r: (-r[1], r[0]))[:100]


def q2(t):
    pa_, ps, su, na, re_ = t["part"], t["partsupp"], t["supplier"], t["nation"], t["region"]
    eur = set(re_[re_.r_name == "EUROPE"].r_regionkey.tolist())
    n1 = na[na.n_regionkey.isin(eur)]
    sn = su.merge(n1, left_on="s_nationkey", right_on="n_nationkey")
    p1 = pa_[(pa_.p_size == 15) & pa_.p_type.str.endswith("BRASS")]
    allj = ps.merge(p1, left_on="ps_partkey", right_on="p_partkey").merge(sn, left_on="ps_suppkey", right_on="s_suppkey")
    mins = allj.groupby("ps_partkey").ps_supplycost.transform("min")
    best = allj[allj.ps_supplycost == mins]
    rows = [(int(r.s_acctbal), r.s_name, r.n_name, int(r.p_partkey), r.p_mfgr, r.s_address, r.s_phone, r.s_comment) for r in best.itertuples()]
    return sorted(rows, key=lambda r: (-r[0], r[2], r[1], r[3]))[:100]


SQL = {2: q2, 13: q13, 16: q16, 17: q17, 19: q19, 20: q20, 21: q21, 22: q22}


def evaluate(q, tables: Tables):
    return SQL[q]({name: tables.get(tid, cols) for name, (tid, cols) in INPUTS[q].items()})
