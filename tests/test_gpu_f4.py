"""SURVEY §8(f).4 on the device, through the C-ABI: window functions over the device segment tree (ldb_gpu_window),
set operations (ldb_gpu_set_op), outer joins that keep the build side's unmatched rows (LDB_JOIN_RIGHT_OUTER /
LDB_JOIN_FULL_OUTER).  Checked against (a) the answers of the reference's REAL SegmentTreeView
(tests/golden/ref_segtree.npz, written by tests/golden/make_ref_window.py from src/runtime/SegmentTreeView.cpp compiled in
place), (b) the oracle restatements (oracle/ldb_oracle.c: ora_window, ora_setop_multiplicity), (c) the expected rows of
the reference's own SQL-level cases where they exist (test/sqlite-small/join.test outer joins are in test_gpu_z_golden).
Bit-exact: everything here is integer / decimal / string work."""
import collections
import decimal
import os

import numpy as np
import pyarrow as pa
import pytest

import golden_io
import oracle_bind
from lingodb_amd import api, capi

pytestmark = pytest.mark.gpu
I64_MIN, I64_MAX = -(2 ** 63), 2 ** 63 - 1
FNS = [(capi.WIN_SUM, (0, 1)), (capi.WIN_MIN, (0, 1)), (capi.WIN_MAX, (0, 1)), (capi.WIN_COUNT, (0, 1)), (capi.WIN_RANK, None), (capi.WIN_COUNT_STAR, None)]


def unscaled(col):
    t = col.type
    if pa.types.is_decimal(t):
        return [None if v.as_py() is None else int(v.as_py().scaleb(t.scale)) for v in col]
    return col.to_pylist()


# ---------------------------------------------------------------- window functions
def test_window_frames_match_the_reference_segment_tree(ctx):
    """one partition, ten frames, SUM / MIN / MAX / COUNT from the real SegmentTreeView; RANK and COUNT(*) from the frame bounds"""
    z = np.load(os.path.join(golden_io.GOLDEN, "ref_segtree.npz"))
    vals, valid = z["vals"], z["valid"]
    n = len(vals)
    rng = np.random.default_rng(5)
    order = rng.permutation(n)  # the rows arrive shuffled; ORDER BY pos restores the golden order
    t = pa.table({"pos": pa.array(np.arange(n, dtype=np.int32)[order]), "v": pa.array([None if not valid[i] else int(vals[i]) for i in order], pa.int64())})
    dev = ctx.register("win_golden", t)
    for fi, (frm, to) in enumerate(z["frames"].tolist()):
        rel, cols = dev.rel().window([], [api.sort_spec((0, 0))], FNS, frame=(frm, to))
        got = cols.to_arrow()
        assert rel.rowids(0).tolist() == np.argsort(order).tolist()  # window order = ORDER BY pos
        for k, fn in enumerate((1, 2, 3, 4)):
            want = [int(v) if (ok or fn == 4) else None for v, ok in zip(z["f%d_fn%d_val" % (fi, fn)].tolist(), z["f%d_fn%d_ok" % (fi, fn)].tolist())]
            assert got.column(k).to_pylist() == want, (fi, fn)
        cur = np.arange(n)
        lo = np.zeros(n, np.int64) if frm == I64_MIN else np.clip(cur + frm, 0, n - 1)
        hi = np.full(n, n - 1) if to == I64_MAX else np.clip(cur + to, 0, n - 1)
        assert got.column(4).to_pylist() == (cur - lo + 1).tolist() and got.column(5).to_pylist() == (hi - lo + 1).tolist()


@pytest.mark.parametrize("frame", [(I64_MIN, 0), (I64_MIN, I64_MAX), (-2, 1), (0, 3), (-5, -1)])
def test_window_partitions_against_the_oracle(ctx, oracle, frame):
    """PARTITION BY p ORDER BY o DESC over 128-bit decimals with NULLs: partitions of very different sizes (one of a
    single row), every frame clamped into its partition; expected values from ora_window over the same order"""
    rng = np.random.default_rng(11)
    sizes = [1, 2, 700, 64, 65, 3000, 5]
    p = np.concatenate([np.full(s, 10 * k - 30, np.int32) for k, s in enumerate(sizes)])
    n = len(p)
    o = np.concatenate([rng.permutation(s).astype(np.int32) for s in sizes])  # unique inside a partition: the order is total
    big = [int(x) * 10 ** 9 + int(y) for x, y in zip(rng.integers(-10 ** 15, 10 ** 15, n), rng.integers(0, 10 ** 9, n))]
    valid = rng.integers(0, 6, n) > 0
    shuffle = rng.permutation(n)
    t = pa.table({"p": pa.array(p[shuffle]), "o": pa.array(o[shuffle]),
                  "d": pa.array([decimal.Decimal(big[i]).scaleb(-2) if valid[i] else None for i in shuffle], pa.decimal128(30, 2))})
    dev = ctx.register("win_parts", t)
    fns = [(capi.WIN_SUM, (0, 2)), (capi.WIN_MIN, (0, 2)), (capi.WIN_MAX, (0, 2)), (capi.WIN_COUNT, (0, 2)), (capi.WIN_RANK, None), (capi.WIN_COUNT_STAR, None)]
    rel, cols = dev.rel().window([(0, 0)], [api.sort_spec((0, 1), True)], fns, frame=frame)
    got = cols.to_arrow()
    # expected window order: p ascending, o descending
    ps, os_ = p[shuffle], o[shuffle]
    want_order = np.lexsort((-os_, ps))
    assert rel.rowids(0).tolist() == want_order.tolist()
    sp = ps[want_order]
    starts = np.concatenate([[0], np.nonzero(np.diff(sp))[0] + 1])
    ends = np.concatenate([starts[1:], [n]])
    part_start = np.repeat(starts, ends - starts)
    part_end = np.repeat(ends, ends - starts)
    svals = [big[shuffle[i]] for i in want_order]
    sok = [1 if valid[shuffle[i]] else 0 for i in want_order]
    for k, (fn, _) in enumerate(fns):
        want = oracle_bind.window(oracle, svals, sok, part_start, part_end, fn, frame[0], frame[1])
        assert unscaled(got.column(k).combine_chunks()) == want, (frame, fn)
    assert got.schema.field(0).type == pa.decimal128(30, 2) and got.schema.field(3).type == pa.int64()


def test_window_edge_cases(ctx):
    """empty input; no PARTITION BY / ORDER BY (one partition in input order); narrow columns and dates"""
    empty = ctx.register("win_empty", pa.table({"a": pa.array([], pa.int32()), "b": pa.array([], pa.int64())}))
    rel, cols = empty.rel().window([(0, 0)], [api.sort_spec((0, 1))], [(capi.WIN_RANK, None), (capi.WIN_SUM, (0, 1))])
    assert rel.rows == 0 and cols.rows == 0
    import datetime

    t = pa.table({"a": pa.array([5, 3, 9, 1], pa.int32()), "dt": pa.array([datetime.date(2020, 1, d) for d in (4, 2, 9, 7)], pa.date32())})
    dev = ctx.register("win_small", t)
    rel, cols = dev.rel().window([], [], [(capi.WIN_SUM, (0, 0)), (capi.WIN_MAX, (0, 1)), (capi.WIN_RANK, None)], frame=(I64_MIN, 0))
    got = cols.to_arrow()
    assert rel.rowids(0).tolist() == [0, 1, 2, 3]
    assert got.column(0).to_pylist() == [5, 8, 17, 18] and got.column(2).to_pylist() == [1, 2, 3, 4]
    assert got.column(1).to_pylist() == [datetime.date(2020, 1, 4), datetime.date(2020, 1, 4), datetime.date(2020, 1, 9), datetime.date(2020, 1, 9)]


# ---------------------------------------------------------------- set operations
def _set_tables():
    left = pa.table({"k": pa.array([1, 1, 1, 2, 2, None, None, 4, 7, 7], pa.int32()), "s": pa.array(["a", "a", "a", "b", "b", None, None, "dd", "x", "y"]),
                     "d": pa.array([decimal.Decimal(v) / 100 if v is not None else None for v in (100, 100, 100, 250, 250, None, None, 4, 7, 7)], pa.decimal128(12, 2))})
    right = pa.table({"k": pa.array([1, 1, 2, 2, 2, None, 5, 7, 7], pa.int32()), "s": pa.array(["a", "a", "b", "b", "b", None, "e", "x", "z"]),
                      "d": pa.array([decimal.Decimal(v) / 100 if v is not None else None for v in (100, 100, 250, 250, 250, None, 5, 7, 7)], pa.decimal128(12, 2))})
    return left, right


@pytest.mark.parametrize("op", [capi.SET_UNION_ALL, capi.SET_UNION, capi.SET_INTERSECT, capi.SET_INTERSECT_ALL, capi.SET_EXCEPT, capi.SET_EXCEPT_ALL])
def test_set_operations(ctx, oracle, op):
    """duplicates on both sides, NULL rows (NULL = NULL in set operations), a string and a decimal column: the result as
    a multiset against the two counters per distinct row and ora_setop_multiplicity"""
    left, right = _set_tables()
    a, b = ctx.register("set_l", left), ctx.register("set_r", right)
    got = collections.Counter(map(tuple, zip(*[c.to_pylist() for c in a.rel().set_op(b.rel(), op).to_arrow().columns])))
    cl = collections.Counter(map(tuple, zip(*[c.to_pylist() for c in left.columns])))
    cr = collections.Counter(map(tuple, zip(*[c.to_pylist() for c in right.columns])))
    want = collections.Counter()
    for row in set(cl) | set(cr):
        m = oracle.lib.ora_setop_multiplicity(op, cl.get(row, 0), cr.get(row, 0))
        if m:
            want[row] = m
    assert got == want, op
    if op == capi.SET_UNION_ALL:
        assert sum(got.values()) == left.num_rows + right.num_rows


def test_set_operations_at_size_and_empty_sides(ctx):
    """200 k x 150 k rows of two integer columns (the group-by path above the LDS limits) + an empty side"""
    rng = np.random.default_rng(2)
    la, lb = rng.integers(0, 300, 200_000).astype(np.int32), rng.integers(0, 200, 200_000).astype(np.int64)
    ra, rb = rng.integers(100, 400, 150_000).astype(np.int32), rng.integers(0, 200, 150_000).astype(np.int64)
    a, b = ctx.register("set_big_l", pa.table({"x": la, "y": lb})), ctx.register("set_big_r", pa.table({"x": ra, "y": rb}))
    cl, cr = collections.Counter(zip(la.tolist(), lb.tolist())), collections.Counter(zip(ra.tolist(), rb.tolist()))
    for op, fn in ((capi.SET_INTERSECT_ALL, lambda l, r: min(l, r)), (capi.SET_EXCEPT_ALL, lambda l, r: max(l - r, 0)), (capi.SET_EXCEPT, lambda l, r: 1 if l > 0 and r == 0 else 0),
                   (capi.SET_UNION, lambda l, r: 1)):
        res = a.rel().set_op(b.rel(), op).to_arrow()
        got = collections.Counter(zip(res.column(0).to_pylist(), res.column(1).to_pylist()))
        want = collections.Counter({k: fn(cl.get(k, 0), cr.get(k, 0)) for k in set(cl) | set(cr) if fn(cl.get(k, 0), cr.get(k, 0))})
        assert got == want, op
    none = ctx.register("set_none", pa.table({"x": pa.array([], pa.int32()), "y": pa.array([], pa.int64())}))
    assert a.rel().set_op(none.rel(), capi.SET_INTERSECT).rows == 0
    assert none.rel().set_op(b.rel(), capi.SET_EXCEPT_ALL).rows == 0
    assert a.rel().set_op(none.rel(), capi.SET_EXCEPT).rows == len(cl)


# ---------------------------------------------------------------- outer joins keeping the build side
@pytest.mark.parametrize("unique_hint", [True, False])
def test_right_and_full_outer_joins(ctx, unique_hint):
    """build keys with duplicates, NULLs and keys no probe row has; probe keys with NULLs and misses: pairs as multisets
    against a nested-loop evaluation (a NULL key matches nothing on either side but its row is kept by the outer side)"""
    build = pa.table({"bk": pa.array([1, 2, 2, None, 9, 10, 11], pa.int32()), "bv": pa.array([10, 20, 21, 30, 90, 100, 110], pa.int64())})
    probe = pa.table({"pk": pa.array([2, 2, 3, None, 1, 11, 11, 4], pa.int32()), "pv": pa.array(list("abcdefgh"))})
    if unique_hint:  # a unique build side: the rank-bitmap table
        build = pa.table({"bk": pa.array([1, 2, None, 9, 10, 11], pa.int32()), "bv": pa.array([10, 20, 30, 90, 100, 110], pa.int64())})
    b, p = ctx.register("oj_b_%d" % unique_hint, build), ctx.register("oj_p_%d" % unique_hint, probe)
    ht = b.rel().join_build([(0, 0)], unique=unique_hint)
    bk, bv, pk, pv = build.column(0).to_pylist(), build.column(1).to_pylist(), probe.column(0).to_pylist(), probe.column(1).to_pylist()
    inner = [(pi, bi) for pi in range(len(pk)) for bi in range(len(bk)) if pk[pi] is not None and pk[pi] == bk[bi]]
    matched_b, matched_p = {bi for _, bi in inner}, {pi for pi, _ in inner}
    for kind in (capi.JOIN_RIGHT_OUTER, capi.JOIN_FULL_OUTER):
        want = [(pv[pi], bv[bi]) for pi, bi in inner] + [(None, bv[bi]) for bi in range(len(bk)) if bi not in matched_b]
        if kind == capi.JOIN_FULL_OUTER:
            want += [(pv[pi], None) for pi in range(len(pk)) if pi not in matched_p]
        out = ht.probe(p.rel(), [(0, 0)], kind).materialize([(0, 1), (1, 1)]).to_arrow()
        got = list(zip(out.column(0).to_pylist(), out.column(1).to_pylist()))
        assert collections.Counter(got) == collections.Counter(want), kind


def test_full_outer_join_in_a_plan(ctx):
    """the plan interpreter's "full_outer" / "right_outer" kinds: COUNT of the padded rows per side"""
    import json

    l = pa.table({"a": pa.array(np.arange(0, 3000, 2, dtype=np.int32))})
    r = pa.table({"b": pa.array(np.arange(0, 3000, 3, dtype=np.int32))})
    plan = {"steps": [{"op": "join_build", "in": "r", "keys": ["b"], "unique": True, "out": "h"},
                      {"op": "join_probe", "ht": "h", "in": "l", "keys": ["a"], "kind": "full_outer", "out": "j"},
                      {"op": "groupby", "in": "j", "keys": [], "aggs": [{"fn": "count_star", "as": "rows"}, {"fn": "count", "expr": "a", "as": "with_a"}, {"fn": "count", "expr": "b", "as": "with_b"}],
                       "est_groups": 1, "out": "result"}], "result": "result"}
    got = ctx.run_plan(json.dumps(plan), {"l": ctx.register("fo_l", l), "r": ctx.register("fo_r", r)}).to_arrow()
    both = len(set(range(0, 3000, 2)) & set(range(0, 3000, 3)))
    assert got.column(1).to_pylist() == [1500] and got.column(2).to_pylist() == [1000] and got.column(0).to_pylist() == [1500 + 1000 - both]


def test_nested_loop_join_equals_brute_force(ctx):
    """translateNLJ (RelAlgToSubOp.cpp:948-1033): a join whose predicate has no equality — a band join
    `lo <= x AND x < hi` against a small dimension table, and the cross product — for the pair-producing and the
    existence kinds, against a numpy brute force; through the plan step as well."""
    import json

    import numpy as np

    rng = np.random.default_rng(5)
    n, m = 20000, 37
    x = rng.integers(0, 1000, n).astype(np.int64)
    lo = np.sort(rng.integers(0, 1000, m)).astype(np.int64)
    hi = lo + rng.integers(0, 60, m)
    probe = ctx.register("nl_probe", pa.table({"x": pa.array(x), "tag": pa.array(np.arange(n, dtype=np.int32))}))
    build = ctx.register("nl_build", pa.table({"lo": pa.array(lo), "hi": pa.array(hi), "b": pa.array(np.arange(m, dtype=np.int32))}))
    match = (x[:, None] >= lo[None, :]) & (x[:, None] < hi[None, :])
    resid = [((0, 0), capi.F_GTE, (0, 0)), ((0, 0), capi.F_LT, (0, 1))]
    pairs = probe.rel().join_nl(build.rel(), resid)
    got = sorted(pairs.materialize([(0, 1), (1, 2)]).to_arrow().to_pylist(), key=lambda r: (r["tag"], r["b"]))
    want = [{"tag": int(i), "b": int(j)} for i, j in zip(*np.nonzero(match))]
    assert got == want and len(want) > 1000
    semi = probe.rel().join_nl(build.rel(), resid, capi.JOIN_SEMI)
    anti = probe.rel().join_nl(build.rel(), resid, capi.JOIN_ANTI)
    has = match.any(axis=1)
    assert sorted(r["tag"] for r in semi.materialize([(0, 1)]).to_arrow().to_pylist()) == list(np.nonzero(has)[0])
    assert sorted(r["tag"] for r in anti.materialize([(0, 1)]).to_arrow().to_pylist()) == list(np.nonzero(~has)[0])
    sb = probe.rel().join_nl(build.rel(), resid, capi.JOIN_SEMI_BUILD)
    assert sorted(r["b"] for r in sb.materialize([(0, 2)]).to_arrow().to_pylist()) == list(np.nonzero(match.any(axis=0))[0])
    lo_ = probe.rel().join_nl(build.rel(), resid, capi.JOIN_LEFT_OUTER)
    assert lo_.rows == int(match.sum() + (~has).sum())
    cross = build.rel().join_nl(build.rel(), ())
    assert cross.rows == m * m
    plan = json.dumps({"name": "band", "inputs": ["p", "b"], "steps": [
        {"op": "join_nl", "in": "p", "build": "b", "residual": [{"probe": "x", "op": "GTE", "build": "lo"}, {"probe": "x", "op": "LT", "build": "hi"}], "out": "j"},
        {"op": "groupby", "in": "j", "keys": ["b"], "aggs": [{"fn": "count_star", "as": "n"}], "out": "g"},
        {"op": "sort", "in": "g", "by": ["b"], "out": "s"},
        {"op": "materialize", "in": "s", "cols": ["b", "n"], "out": "result"}], "result": "result"})
    rows = ctx.run_plan(plan, {"p": probe, "b": build}).to_arrow().to_pylist()
    cnt = match.sum(axis=0)
    assert rows == [{"b": int(j), "n": int(cnt[j])} for j in range(m) if cnt[j]]


# ------------------------------------------------------------------ subop.loop / nested_map (f4's fourth item)
def _plan(name):
    import os

    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lingo-db_amd", "plans", "subop", name)) as f:
        return f.read()


def test_loop_counter_of_the_reference_lit_test(ctx):
    """test/lit/SubOp/loop.mlir: a counter state carried through subop.loop, continued while ctr < 5 → CHECK: 6"""
    t = ctx.register("ctr0", pa.table({"ctr": pa.array([0], pa.int32())}))
    got = ctx.run_plan(_plan("loop_counter.json"), {"ctr0": t}).to_arrow().to_pylist()
    assert got == [{"ctr": 6}]
    # a loop that never reaches its fixpoint is an error, not a hang
    import json

    p = json.loads(_plan("loop_counter.json"))
    p["steps"][0]["body"][2]["expr"] = {"cmp": ["GTE", "ctr", 0]}
    p["steps"][0]["max_iterations"] = 7
    with pytest.raises(capi.LdbError) as e:
        ctx.run_plan(json.dumps(p), {"ctr0": t})
    assert "no fixpoint after 7 iterations" in str(e.value)


KMEANS_POINTS = [(1, 1), (1, 2), (2, 1), (2, 4), (2, 5), (3, 2), (3, 5), (6, 3), (6, 5), (8, 4)]  # kmeans.mlir's ten points


def _kmeans_model(points, cents):
    """the plan's arithmetic in Python integers: nearest centroid by squared distance (ties → lowest id), truncating means"""
    iters = 0
    while True:
        iters += 1
        acc = {}
        for px, py in points:
            best = min(((cx - px) ** 2 + (cy - py) ** 2) * 1024 + cid for cid, (cx, cy) in cents.items())
            a = acc.setdefault(best % 1024, [0, 0, 0])
            a[0] += px
            a[1] += py
            a[2] += 1
        nxt = {cid: (sx // n, sy // n) for cid, (sx, sy, n) in acc.items()}
        moved = any(nxt[c] != cents[c] for c in nxt if c in cents)
        cents = nxt
        if not moved:
            return cents, iters


@pytest.mark.parametrize("prepared", [False, True])
def test_kmeans_to_a_fixpoint_after_the_reference_lit_test(ctx, prepared):
    """test/lit/SubOp/kmeans.mlir in fixed point (x 1000): nested_map (per point: scan the centroids, distance, arg-min state),
    hash aggregation per centroid, next centroids = means, loop while a centroid moved.  The fixpoint is the lit test's CHECK
    (1.75, 1.5) / (2.3333333, 4.6666665) / (6.6666665, 4) truncated to three digits — and equals a Python model of the same integer
    arithmetic, also on 3 000 random points with 12 centroids"""
    pts = [(x * 1000, y * 1000) for x, y in KMEANS_POINTS]
    cases = [(pts, {i: pts[i] for i in range(3)}, [{"id": 0, "x": 1750, "y": 1500}, {"id": 1, "x": 2333, "y": 4666}, {"id": 2, "x": 6666, "y": 4000}])]
    rng = np.random.default_rng(8)
    big = [(int(x), int(y)) for x, y in zip(rng.integers(0, 100_000, 3000), rng.integers(0, 100_000, 3000))]
    cases.append((big, {i: big[i * 37] for i in range(12)}, None))
    for points, cents, literal in cases:
        model, iters = _kmeans_model(points, dict(cents))
        want = [{"id": c, "x": model[c][0], "y": model[c][1]} for c in sorted(model)]
        if literal is not None:
            assert want == literal
        assert iters >= 3  # (a real iteration, not a one-shot)
        tp = ctx.register("points", pa.table({"px": pa.array([p[0] for p in points], pa.int64()), "py": pa.array([p[1] for p in points], pa.int64())}))
        tc = ctx.register("initial", pa.table({"cx": pa.array([cents[c][0] for c in sorted(cents)], pa.int64()), "cy": pa.array([cents[c][1] for c in sorted(cents)], pa.int64()),
                                               "cid": pa.array(sorted(cents), pa.int64())}))
        if prepared:
            plan = ctx.prepare_plan(_plan("kmeans.json"))
            for _ in range(3):
                got = plan.execute({"points": tp, "initial": tc}).to_arrow().to_pylist()
                assert got == want
            assert plan.stats()["misses"] == 0 and plan.stats()["replays"] >= 1
            plan.release()
        else:
            assert ctx.run_plan(_plan("kmeans.json"), {"points": tp, "initial": tc}).to_arrow().to_pylist() == want


def test_nested_map_without_a_reduce_returns_the_nested_stream(ctx):
    """the general form: the nested pipeline's tuples themselves (outer x scanned state, filtered by the residual, mapped)"""
    import json

    a = ctx.register("nm_a", pa.table({"x": pa.array([1, 5, 9, 12], pa.int64())}))
    b = ctx.register("nm_b", pa.table({"lo": pa.array([0, 4, 10], pa.int64()), "hi": pa.array([6, 9, 20], pa.int64())}))
    plan = {"steps": [{"op": "nested_map", "in": "a", "scan": "b", "residual": [{"probe": "x", "build": "lo", "op": "GTE"}, {"probe": "x", "build": "hi", "op": "LTE"}],
                       "map": [{"as": "w", "expr": {"sub": ["hi", "x"]}}], "out": "s"},
                      {"op": "materialize", "in": "s", "cols": ["x", "lo", "w"], "out": "result"}], "result": "result"}
    got = sorted(tuple(r.values()) for r in ctx.run_plan(json.dumps(plan), {"a": a, "b": b}).to_arrow().to_pylist())
    want = sorted((x, lo, hi - x) for x in (1, 5, 9, 12) for lo, hi in ((0, 6), (4, 9), (10, 20)) if lo <= x <= hi)
    assert got == want
