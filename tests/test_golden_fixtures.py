"""The C oracle against tests/golden/ — answers of the reference's own runtime objects
(Hash.cpp, Restrictions.cpp, LazyJoinHashtable.cpp, PreAggregationHashtable.cpp,
StringRuntime.cpp, DateRuntime.cpp) recorded by tests/golden/make_ref_golden.py.  Unlike
test_oracle_vs_ref.py this needs neither /root/reference nor oracle/_ref, so the pin travels."""
import numpy as np
import pyarrow as pa

from lingodb_amd import api, capi
from oracle_bind import HostTable
import os

import golden_io
from golden_io import GOLDEN
import tpch_data


def _h64(x):
    m = (x * 11400714819323198549) & (2 ** 64 - 1)
    return m ^ int.from_bytes(m.to_bytes(8, "little"), "big")


def test_oracle_hash_vs_golden(oracle):
    """db.hash (LowerToStd.cpp:1065-1152, what generated code computes and what the oracle and
    the kernels follow) against the reference runtime's dbHashApplyColumn (Hash.cpp:58-247).
    The two agree on every type and on NULLs of every fixed-width type (NULL folds nothing), with
    one divergence INSIDE the reference: a NULL *string* is skipped by the lowering
    (LowerToStd.cpp:1118-1132) but folded as the default VarLen32 image {len 0, first4 0xffffffff}
    by the runtime (Hash.cpp:228-230, helpers.h:193).  The hot path follows the lowering; the
    fixture rows with a NULL string are checked against the runtime's rule instead."""
    t = golden_io.types_table()
    h = HostTable(t).rel()
    cases = golden_io.hash_cases()
    assert len(cases) == 17
    s_null = np.array([v is None for v in t.column(10).to_pylist()])
    assert 10 < s_null.sum() < 40
    for keys, want in cases:
        got = oracle.hash_keys(h, [(0, k) for k in keys])
        rows = ~s_null if 10 in keys else np.ones(len(want), bool)
        assert np.array_equal(got[rows], want[rows]), keys
    single = dict((tuple(k), w) for k, w in cases)[(10,)]
    assert set(single[s_null].tolist()) == {_h64(0xFFFFFFFF << 32) ^ 0}  # h64(first64) ^ bswap(h64(last64 = 0))
    assert not oracle.hash_keys(h, [(0, 10)])[s_null].any()  # db.hash: NULL contributes nothing


def test_oracle_filters_vs_golden(oracle):
    meta, cases = golden_io.filter_cases()
    li = tpch_data.host_table(tpch_data.LINEITEM, meta["orders"])
    assert li.num_rows == meta["lineitem_rows"]
    h = HostTable(li).rel()
    nonempty = 0
    for case, want in cases:
        got = oracle.scan_filter(h, golden_io.preds_of(case, li.schema), threads=2)
        assert np.array_equal(got, want), case
        nonempty += len(want) > 0
    assert nonempty >= 6


def test_oracle_join_vs_golden(oracle):
    z = golden_io.npz("ref_join.npz")
    b, p = (HostTable(pa.table({"k": pa.array(z[k])})).rel() for k in ("build_keys", "probe_keys"))
    gp, gb, _ = oracle.join(b, [(0, 0)], p, [(0, 0)], capi.JOIN_INNER, threads=3)
    order = np.lexsort((gb, gp))
    assert np.array_equal(gp[order], z["probe_rows"]) and np.array_equal(gb[order], z["build_rows"])


def test_oracle_groupby_vs_golden(oracle):
    z = golden_io.npz("ref_groupby.npz")
    rel = HostTable(pa.table({"k": pa.array(z["keys"]), "v": pa.array(z["vals"])})).rel()
    rep, v, _ = oracle.groupby(rel, [(0, 0)], [api.agg(capi.AGG_SUM, api.col_expr((0, 1))), api.agg(capi.AGG_COUNT_STAR)], threads=3)
    got = sorted((int(z["keys"][r]), (s if s < 1 << 63 else s - (1 << 64)), c) for r, (s, c) in zip(rep, v))
    assert got == list(zip(z["group_keys"].tolist(), z["sums"].tolist(), z["counts"].tolist()))


def test_oracle_like_vs_golden(oracle):
    cases = golden_io.like_cases()
    assert len(cases) > 3000 and any(w for _, _, w in cases)
    bad = [(s, p, w) for s, p, w in cases if oracle.like(s.encode(), p.encode()) != w]
    assert not bad, bad[:10]


def test_oracle_extract_year_vs_golden(oracle):
    for d, y in golden_io.year_cases():
        assert oracle.extract_year(d) == y, d


def _nullable_i64(values):
    return pa.table({"v": pa.array(values, pa.int64())})


def test_oracle_sqlite_small_join_cases(oracle):
    """the reference's SQL-level join tests (inner / NULL keys / left, right, full outer / semi /
    anti / mark, mixed int-decimal keys) through the oracle's join"""
    def join(build, probe, kind):
        b, p = HostTable(_nullable_i64(build)).rel(), HostTable(_nullable_i64(probe)).rel()
        pr, br, mark = oracle.join(b, [(0, 0)], p, [(0, 0)], kind)
        if kind == capi.JOIN_MARK:
            return list(range(len(probe))), None, mark.tolist()
        return pr.tolist(), None if br is None else br.tolist(), None

    cases = golden_io.sqlite_join_cases()
    assert len(cases) == 22
    for case in cases:
        got, want = golden_io.run_sqlite_join_case(case, join)
        assert got == want, case["source"]


# ---- sort / top-k / key-less aggregation / generic hash map / substr: answers of the reference's own
# Sorting.cpp (GrowingBuffer::sort → parallelSort), Heap.cpp, SimpleState.cpp, Hashtable.cpp, StringRuntime::substr
def _sort_cases():
    import numpy as np

    z = np.load(os.path.join(GOLDEN, "ref_sort.npz"))
    return z, ["small", "tie_heavy", "large"]


def _keys_table(keys):
    import pyarrow as pa

    return pa.table({"k%d" % j: pa.array(keys[:, j], pa.int64()) for j in range(keys.shape[1])})


def test_oracle_sort_and_topk_match_reference_objects(oracle):
    import numpy as np

    from lingodb_amd import api
    from oracle_bind import HostTable

    z, names = _sort_cases()
    for name in names:
        keys, desc, perm = z[name + "_keys"], z[name + "_desc"], z[name + "_perm"]
        t = HostTable(_keys_table(keys))
        specs = [api.sort_spec((0, j), bool(desc[j])) for j in range(keys.shape[1])]
        assert np.array_equal(oracle.sort(t.rel(), specs), perm), name  # the oracle's sort is stable = the fixture's row-number tie-break
        for kt in (1, 10, 100):
            assert np.array_equal(oracle.topk(t.rel(), specs, kt), z[f"{name}_top{kt}"]), (name, kt)


def test_oracle_keyless_and_hashmap_match_reference_objects(oracle):
    import numpy as np
    import pyarrow as pa

    from lingodb_amd import api, capi
    from oracle_bind import HostTable

    z, _ = _sort_cases()
    t = HostTable(pa.table({"v": pa.array(z["ss_vals"], pa.int64()), "keep": pa.array(z["ss_keep"].astype(np.int32), pa.int32())}))
    aggs = [api.agg(capi.AGG_SUM, api.col_expr((0, 0)), wide=True, out_type=capi.T_DECIMAL128, p=38, s=0), api.agg(capi.AGG_COUNT_STAR)]
    _, vals, valid = oracle.groupby(t.rel(), [], aggs, [api.pred((0, 1), capi.F_EQ, 1)])
    want = (int(z["ss_sum_lohi"][1]) << 64) | (int(z["ss_sum_lohi"][0]) & 0xFFFFFFFFFFFFFFFF)
    sgn = lambda v: v - (1 << 128) if v >= 1 << 127 else v
    assert sgn(vals[0][0]) == sgn(want) and vals[0][1] == int(z["ss_count"][0])
    _, vals0, valid0 = oracle.groupby(t.rel(), [], aggs, [api.pred((0, 1), capi.F_EQ, 7)])  # nothing passes: SUM is NULL, COUNT 0 (SimpleState over no rows)
    assert not valid0[0][0] and vals0[0][1] == 0 == int(z["ss_empty_count"][0])
    g = HostTable(pa.table({"k": pa.array(z["ht_keys"], pa.int64()), "v": pa.array(z["ht_vals"], pa.int64())}))
    rep, gv, _ = oracle.groupby(g.rel(), [(0, 0)], [api.agg(capi.AGG_SUM, api.col_expr((0, 1))), api.agg(capi.AGG_COUNT_STAR)])
    got = sorted((int(z["ht_keys"][r]), sgn(v[0]), v[1]) for r, v in zip(rep, gv))
    assert got == list(zip(z["ht_out_keys"].tolist(), z["ht_out_sums"].tolist(), z["ht_out_counts"].tolist()))


def test_oracle_substr_matches_reference_string_runtime(oracle):
    import json

    with open(os.path.join(GOLDEN, "ref_substr.json")) as f:
        cases = json.load(f)
    assert len(cases) >= 1000
    for s, fr, ln, want in cases:
        assert oracle.substr(s, fr, ln).decode() == want, (s, fr, ln)


def test_window_frames_match_the_reference_segment_tree(oracle):
    """ora_window (frames clamped into the partition + the restated SegmentTreeView) against the answers of the
    reference's real SegmentTreeView for ten frames x SUM / MIN / MAX / COUNT (tests/golden/ref_segtree.npz)"""
    import oracle_bind

    z = np.load(os.path.join(golden_io.GOLDEN, "ref_segtree.npz"))
    vals, valid = [int(v) for v in z["vals"]], z["valid"]
    n = len(vals)
    for fi, (frm, to) in enumerate(z["frames"].tolist()):
        for fn in (1, 2, 3, 4):
            want = [int(v) if (k or fn == 4) else None for v, k in zip(z["f%d_fn%d_val" % (fi, fn)].tolist(), z["f%d_fn%d_ok" % (fi, fn)].tolist())]
            got = oracle_bind.window(oracle, vals, valid, [0] * n, [n] * n, fn, frm, to)
            assert got == want, (fi, fn)


def test_outer_join_fixture_of_the_reference_hash_multi_map_is_what_sql_says():
    """tests/golden/ref_hmm.npz (the reference's real HashMultiMap driven as translateHJWithMarker drives it,
    RelAlgToSubOp.cpp:1248-1287) against an independent evaluation: the pairs are the equi-join on non-NULL keys, the
    flag-less build rows are those no probe key reaches (every NULL-key row among them), the partner-less probe rows the rest"""
    import collections

    z = np.load(os.path.join(golden_io.GOLDEN, "ref_hmm.npz"))
    for c in range(int(z["n_cases"][0])):
        bk, bv, pk, pv = (z["c%d_%s" % (c, n)] for n in ("bk", "bv", "pk", "pv"))
        by_key = collections.defaultdict(list)
        for i, (k, ok) in enumerate(zip(bk.tolist(), bv.tolist())):
            if ok:
                by_key[k].append(i)
        want_pairs = sorted((j, i) for j, (k, ok) in enumerate(zip(pk.tolist(), pv.tolist())) if ok for i in by_key.get(k, ()))
        assert sorted(zip(z["c%d_pairs_p" % c].tolist(), z["c%d_pairs_b" % c].tolist())) == want_pairs, c
        probed = {k for k, ok in zip(pk.tolist(), pv.tolist()) if ok}
        assert sorted(z["c%d_unmatched_b" % c].tolist()) == [i for i, (k, ok) in enumerate(zip(bk.tolist(), bv.tolist())) if not ok or k not in probed], c
        assert z["c%d_probe_matched" % c].tolist() == [1 if ok and k in by_key else 0 for k, ok in zip(pk.tolist(), pv.tolist())], c
