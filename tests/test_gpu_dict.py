"""utf8 dictionary encoding at registration (csrc/ldb_dict.hip): every result must be what the string path
gives — the tests run each operator twice, on a table registered with dictionaries (the default) and on the same table
registered with option dict_encode = 0, and on small inputs also against Python's own string semantics (bytewise order =
std::string_view order, StringRuntime.cpp:242-256)."""
import collections

import numpy as np
import pyarrow as pa
import pytest

from lingodb_amd import api, capi

pytestmark = pytest.mark.gpu
WORDS = ["", "AIR", "AIR REG", "FOB", "MAIL", "RAIL", "REG AIR", "SHIP", "TRUCK", "Zürich", "a", "ab", "abc", "b", "ünïcode", "中文"]


def make(n, seed=1, nulls=True):
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, len(WORDS), n)
    s = [None if nulls and i % 17 == 3 else WORDS[k] for i, k in enumerate(idx.tolist())]
    return pa.table({"s": pa.array(s, pa.string()), "v": pa.array(rng.integers(0, 1000, n).astype(np.int64)), "k": pa.array(rng.integers(0, 50, n).astype(np.int32))})


@pytest.fixture(scope="module")
def tables(ctx):
    t = make(50_000)
    lib = capi.gpu_lib()
    enc = ctx.register("dict_on", t)
    lib.ldb_gpu_set_option(b"dict_encode", 0)
    try:
        plain = ctx.register("dict_off", t)
    finally:
        lib.ldb_gpu_set_option(b"dict_encode", 1)
    assert enc.dict_size(0) == len(WORDS) and plain.dict_size(0) == -1
    return t, enc, plain


PREDS = [(capi.F_EQ, "MAIL", None), (capi.F_NEQ, "MAIL", None), (capi.F_LT, "FOB", None), (capi.F_LTE, "FOB", None), (capi.F_GT, "a", None), (capi.F_GTE, "ab", None),
         (capi.F_EQ, "not there", None), (capi.F_NEQ, "not there", None), (capi.F_LT, "", None), (capi.F_GTE, "", None),
         (capi.F_IN, None, ["MAIL", "SHIP", "nope"]), (capi.F_IN, None, ["nope"]), (capi.F_LIKE, "%AIR%", None), (capi.F_NOT_LIKE, "%AIR%", None),
         (capi.F_LIKE, "a%", None), (capi.F_LIKE, "_", None), (capi.F_LIKE, "%", None), (capi.F_LIKE, "R_IL", None), (capi.F_LIKE, "Z%ch", None)]


@pytest.mark.parametrize("op,value,values", PREDS)
def test_predicates_equal_the_string_path(tables, op, value, values):
    t, enc, plain = tables
    mk = lambda: [api.pred((0, 0), op, value, values=values)]  # noqa: E731
    a, b = enc.rel().scan_filter(mk()).rowids(0), plain.rel().scan_filter(mk()).rowids(0)
    assert np.array_equal(a, b), (op, value, values)
    s = t.column(0).to_pylist()
    import re

    def like(x, pat):
        rx = "".join(".*" if ch == "%" else "." if ch == "_" else re.escape(ch) for ch in pat)
        return re.fullmatch(rx, x, re.S) is not None

    cmp_ = {capi.F_EQ: lambda x: x == value, capi.F_NEQ: lambda x: x != value, capi.F_LT: lambda x: x.encode() < value.encode(), capi.F_LTE: lambda x: x.encode() <= value.encode(),
            capi.F_GT: lambda x: x.encode() > value.encode(), capi.F_GTE: lambda x: x.encode() >= value.encode(), capi.F_IN: lambda x: x in values,
            capi.F_LIKE: lambda x: like(x, value), capi.F_NOT_LIKE: lambda x: not like(x, value)}[op]
    assert a.tolist() == [i for i, x in enumerate(s) if x is not None and cmp_(x)]
    # the same conjunct fused into a count (the batched evaluator) and next to an integer conjunct
    both = lambda: [api.pred((0, 2), capi.F_LT, 25), api.pred((0, 0), op, value, values=values)]  # noqa: E731
    assert enc.rel().scan_count(both()) == plain.rel().scan_count(both())


def test_group_by_and_sort_on_codes(tables):
    t, enc, plain = tables
    aggs = lambda: [api.agg(capi.AGG_SUM, api.col_expr((0, 1))), api.agg(capi.AGG_COUNT_STAR)]  # noqa: E731
    for keys in ([(0, 0)], [(0, 0), (0, 2)], [(0, 2), (0, 0)]):
        ga, gb = enc.rel().groupby(keys, aggs(), est_groups=2000).to_arrow(), plain.rel().groupby(keys, aggs(), est_groups=2000).to_arrow()
        rows = lambda g: collections.Counter(zip(*[c.to_pylist() for c in g.columns]))  # noqa: E731
        assert rows(ga) == rows(gb) and ga.schema == gb.schema, keys  # key columns come out as utf8 either way, NULL group included
    nn = [api.pred((0, 0), capi.F_NOTNULL)]
    specs = [api.sort_spec((0, 0), True), api.sort_spec((0, 2))]
    fa, fb = enc.rel().scan_filter(nn), plain.rel().scan_filter([api.pred((0, 0), capi.F_NOTNULL)])
    sa, sb = fa.sort(specs).materialize([(0, 0), (0, 2)]).to_arrow(), fb.sort(specs).materialize([(0, 0), (0, 2)]).to_arrow()
    assert sa.equals(sb)
    s = sa.column(0).to_pylist()
    assert all(s[i].encode() >= s[i + 1].encode() for i in range(len(s) - 1))  # bytewise descending: the dictionary preserves string order
    ta = fa.topk([api.sort_spec((0, 0)), api.sort_spec((0, 1), True)], 37).materialize([(0, 0), (0, 1)]).to_arrow()
    tb = fb.topk([api.sort_spec((0, 0)), api.sort_spec((0, 1), True)], 37).materialize([(0, 0), (0, 1)]).to_arrow()
    assert ta.equals(tb)


def test_joins_and_hashes_still_see_strings(tables, ctx):
    """joins, db.hash and the shuffle's partitioning must not depend on a rank-local dictionary"""
    t, enc, plain = tables
    assert np.array_equal(enc.rel().hash_keys([(0, 0)]), plain.rel().hash_keys([(0, 0)]))
    dim = ctx.register("dict_dim", pa.table({"w": pa.array(["MAIL", "SHIP", "abc", "zzz"]), "n": pa.array([1, 2, 3, 4], pa.int32())}))
    ja = dim.rel().join_build([(0, 0)], unique=True).probe(enc.rel(), [(0, 0)])
    jb = dim.rel().join_build([(0, 0)], unique=True).probe(plain.rel(), [(0, 0)])
    assert np.array_equal(ja.rowids(0), jb.rowids(0)) and np.array_equal(ja.rowids(1), jb.rowids(1)) and ja.rows > 0
    pa_, ca = enc.rel().partition([(0, 0)], 3, [(0, 0), (0, 1)])
    pb_, cb = plain.rel().partition([(0, 0)], 3, [(0, 0), (0, 1)])
    assert ca == cb and pa_.to_arrow().equals(pb_.to_arrow())


def test_high_cardinality_columns_are_left_alone(ctx):
    n = 100_000
    t = pa.table({"c": pa.array(["comment %d" % (i * 7919 % n) for i in range(n)]), "few": pa.array(["x%d" % (i % 3) for i in range(n)])})
    dev = ctx.register("dict_hi", t)
    assert dev.dict_size(0) == -1 and dev.dict_size(1) == 3
    small = ctx.register("dict_small", t.slice(0, 100))  # below dict_min_rows: not encoded automatically …
    assert small.dict_size(1) == -1
    assert small.dict_encode(1) == 3 and small.dict_size(1) == 3  # … but on request
    got = small.rel().scan_filter([api.pred((0, 1), capi.F_EQ, "x1")]).rowids(0)
    assert got.tolist() == [i for i in range(100) if i % 3 == 1]


def test_tpch_q12_q16_q19_unchanged_by_the_dictionary(ctx):
    """the three plans whose string predicates / group keys go through codes now: same rows with dict_encode = 0"""
    import tpch_plans

    lib = capi.gpu_lib()
    queries = [12, 16, 19, 4, 3]
    on = tpch_plans.Runner(ctx, tpch_plans.Database(ctx, 60_000, 0, 1, queries, False), 1, None, None)
    assert on.db.lineitem.dict_size(on.db.lineitem.col("l_shipmode")) == 7
    lib.ldb_gpu_set_option(b"dict_encode", 0)
    try:
        off = tpch_plans.Runner(ctx, tpch_plans.Database(ctx, 60_000, 0, 1, queries, False), 1, None, None)
        assert off.db.lineitem.dict_size(off.db.lineitem.col("l_shipmode")) == -1
        want = {q: off.run(q).to_arrow() for q in queries}
    finally:
        lib.ldb_gpu_set_option(b"dict_encode", 1)
    for q in queries:
        assert on.run(q).to_arrow().equals(want[q]), q


def test_gathered_columns_inherit_the_dictionary(tables):
    """a group-by result's key column and a materialised column keep the source's dictionary (codes gathered alongside):
    the second-level GROUP BY / ORDER BY over them (TPC-H Q16's shape) still equals the string path"""
    t, enc, plain = tables
    aggs = lambda: [api.agg(capi.AGG_COUNT_STAR)]  # noqa: E731
    g1a, g1b = enc.rel().groupby([(0, 0), (0, 2)], aggs(), est_groups=2000), plain.rel().groupby([(0, 0), (0, 2)], aggs(), est_groups=2000)
    assert g1a.dict_size(0) == len(WORDS) and g1b.dict_size(0) == -1
    g2a, g2b = g1a.rel().groupby([(0, 0)], aggs(), est_groups=64).to_arrow(), g1b.rel().groupby([(0, 0)], aggs(), est_groups=64).to_arrow()
    rows = lambda g: collections.Counter(zip(*[c.to_pylist() for c in g.columns]))  # noqa: E731
    assert rows(g2a) == rows(g2b) and g2a.num_rows == len(WORDS) + 1  # + the NULL group
    m = enc.rel().scan_filter([api.pred((0, 2), capi.F_LT, 5), api.pred((0, 0), capi.F_NOTNULL)]).materialize([(0, 0), (0, 1)])
    assert m.dict_size(0) == len(WORDS)
    srt = m.rel().sort([api.sort_spec((0, 0)), api.sort_spec((0, 1))]).materialize([(0, 0), (0, 1)]).to_arrow()
    s = srt.column(0).to_pylist()
    assert all(s[i].encode() <= s[i + 1].encode() for i in range(len(s) - 1)) and len(s) == m.rows
    assert m.rel().scan_filter([api.pred((0, 0), capi.F_LIKE, "%AIR%")]).rows == sum(1 for x in m.to_arrow().column(0).to_pylist() if "AIR" in x)


# ------------------------------------------------------------------ lazy strings (round 4): gathered dictionary columns keep codes only
@pytest.mark.parametrize("lazy", [1, 0])
def test_lazy_dictionary_strings_are_written_out_only_where_bytes_are_needed(ctx, tables, lazy):
    """a column gathered from a dictionary-encoded one carries codes + the shared dictionary and no bytes (>= 4 096 rows) until
    export, a byte-wise join / LIKE, a concatenation or a second gather below the threshold needs them: every consumer must
    see exactly the strings — with lazy_strings = 0 (bytes written at every gather) as the control"""
    t, enc, plain = tables
    lib = capi.gpu_lib()
    lib.ldb_gpu_set_option(b"lazy_strings", lazy)
    try:
        s, v, k = t.column(0).to_pylist(), t.column(1).to_pylist(), t.column(2).to_pylist()
        keep = [i for i in range(len(s)) if v[i] < 700]
        m = enc.rel().scan_filter([api.pred((0, 1), capi.F_LT, 700)]).materialize([(0, 0), (0, 2), (0, 1)])  # lazy when on: 35 k rows
        assert m.dict_size(0) == len(WORDS)
        # (1) export writes the strings (NULLs stay NULL)
        assert m.to_arrow().column(0).to_pylist() == [s[i] for i in keep]
        # (2) a second gather of few rows from the (possibly still lazy) column
        few = m.rel().scan_filter([api.pred((0, 2), capi.F_LT, 1)]).materialize([(0, 0), (0, 2)])
        assert few.to_arrow().column(0).to_pylist() == [s[i] for i in keep if v[i] < 1]
        # (3) group-by on the string key (codes) of a fresh lazy column, keys come out as strings
        m2 = enc.rel().scan_filter([api.pred((0, 1), capi.F_LT, 700)]).materialize([(0, 0), (0, 1)])
        g = m2.rel().groupby([(0, 0)], [api.agg(capi.AGG_COUNT_STAR), api.agg(capi.AGG_SUM, api.col_expr((0, 1)))], est_groups=32).to_arrow()
        want = collections.Counter(), collections.Counter()
        for i in keep:
            want[0][s[i]] += 1
            want[1][s[i]] += v[i]
        got = {a: (b, c) for a, b, c in zip(g.column(0).to_pylist(), g.column(1).to_pylist(), g.column(2).to_pylist())}
        assert got == {key: (want[0][key], want[1][key]) for key in want[0]}
        # (4) byte-wise consumers of a fresh lazy column: LIKE through the dictionary, an equi-join on the strings
        m3 = enc.rel().scan_filter([api.pred((0, 1), capi.F_LT, 700)]).materialize([(0, 0), (0, 1)])
        hit = m3.rel().scan_filter([api.pred((0, 0), capi.F_LIKE, "%AIR%")]).rowids(0)
        assert hit.tolist() == [j for j, i in enumerate(keep) if s[i] is not None and "AIR" in s[i]]
        words = ctx.register("lazy_words", pa.table({"w": pa.array(["MAIL", "Zürich", "nope", "中文"], pa.string()), "id": pa.array([0, 1, 2, 3], pa.int32())}))
        j = words.rel().join_build([(0, 0)], unique=True).probe(m3.rel(), [(0, 0)], capi.JOIN_INNER)
        pr, br = j.rowids(0), j.rowids(1)
        wl = ["MAIL", "Zürich", "nope", "中文"]
        assert sorted(zip(pr.tolist(), br.tolist())) == sorted((jj, wl.index(s[i])) for jj, i in enumerate(keep) if s[i] in wl)
        # (5) concatenation (set operation) of two lazy columns
        a1 = enc.rel().scan_filter([api.pred((0, 1), capi.F_LT, 100)]).materialize([(0, 0)])
        a2 = enc.rel().scan_filter([api.pred((0, 1), capi.F_GTE, 900)]).materialize([(0, 0)])
        u = a1.rel().set_op(a2.rel(), capi.SET_UNION_ALL).to_arrow().column(0).to_pylist()
        assert sorted(x or "\0" for x in u) == sorted((s[i] or "\0") for i in range(len(s)) if v[i] < 100 or v[i] >= 900)
    finally:
        lib.ldb_gpu_set_option(b"lazy_strings", 1)
