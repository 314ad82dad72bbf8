"""Round-4 host-floor work under test: the single-pass (chained) scan, the fused bitmap → row-id compaction, the descriptor
cache and prepared plans that replay their read-back trace (ldb_plan_prepare / ldb_plan_execute, ldb_gpu_trace_*).

Parity bar: the chained scan / compaction paths against numpy, bit-exact; a replayed execution returns exactly the rows the
recording execution (and the plain interpreter) returned; a replay over data that changed behind the library's back is
DETECTED (the counts no longer hold) and repeated — never a wrong answer."""
import ctypes as C
import json

import numpy as np
import pyarrow as pa
import pytest

from lingodb_amd import api, capi
import tpch_plans

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ chained scan / compaction at sizes beyond one tile
@pytest.mark.parametrize("n", [2049, 300_000, 40_000_000])
def test_scan_filter_row_ids_beyond_one_scan_tile(ctx, n):
    """k_scan_bitmap → (chained) scan of the block counts → k_scan_expand; 40 M rows = 2 442 count blocks > one tile"""
    rng = np.random.default_rng(n)
    x = rng.integers(0, 1000, n, dtype=np.int32)
    t = ctx.register("chain_scan", pa.table({"x": pa.array(x, pa.int32())}))
    for cut in (1, 37, 500, 999):
        got = t.rel().scan_filter([api.pred((0, 0), capi.F_LT, cut)]).rowids(0)
        assert np.array_equal(got, np.nonzero(x < cut)[0].astype(np.uint32)), (n, cut)
    t.release()


@pytest.mark.parametrize("single_pass", [1, 0])
def test_semi_anti_inner_over_many_compaction_tiles(ctx, single_pass):
    """probe of 9 M rows = 140 625 bitmap words = 69 compaction tiles: dense tiles, sparse tiles and empty tiles;
    the three-level scan (scan_single_pass = 0) must give the same row ids"""
    lib = capi.gpu_lib()
    lib.ldb_gpu_set_option(b"scan_single_pass", single_pass)
    try:
        rng = np.random.default_rng(11)
        n = 9_000_000
        pk = rng.integers(0, 1_000_000, n).astype(np.int32)
        pk[: n // 3] = rng.integers(2_000_000, 3_000_000, n // 3)  # no partner at all: empty tiles
        pk[n // 3 : n // 2] = rng.integers(0, 5_000, n // 2 - n // 3)  # every row matches: dense tiles
        bk = np.unique(np.concatenate([np.arange(0, 5_000), rng.integers(5_000, 1_000_000, 30_000)])).astype(np.int32)
        p, b = ctx.register("cp_p", pa.table({"k": pa.array(pk, pa.int32())})), ctx.register("cp_b", pa.table({"k": pa.array(bk, pa.int32())}))
        ht = b.rel().join_build([(0, 0)], unique=True)
        hit = np.isin(pk, bk)
        assert np.array_equal(ht.probe(p.rel(), [(0, 0)], capi.JOIN_SEMI).rowids(0), np.nonzero(hit)[0].astype(np.uint32))
        assert np.array_equal(ht.probe(p.rel(), [(0, 0)], capi.JOIN_ANTI).rowids(0), np.nonzero(~hit)[0].astype(np.uint32))
        out = ht.probe(p.rel(), [(0, 0)], capi.JOIN_INNER)
        rp, rb = out.rowids(0), out.rowids(1)
        assert np.array_equal(rp, np.nonzero(hit)[0].astype(np.uint32))
        assert np.array_equal(bk[rb], pk[rp])
        # build-side semi join (flags → bitmap → compaction over the BUILD rows)
        sb = ht.probe(p.rel(), [(0, 0)], capi.JOIN_SEMI_BUILD).rowids(0)
        assert np.array_equal(sb, np.nonzero(np.isin(bk, pk))[0].astype(np.uint32))
        ht.release(), p.release(), b.release()
    finally:
        lib.ldb_gpu_set_option(b"scan_single_pass", 1)


def test_string_gather_offsets_over_many_scan_tiles(ctx):
    """materialising 700 000 strings: the int64 offset scan runs as a chain of 342 tiles"""
    rng = np.random.default_rng(5)
    n = 700_000
    words = ["", "a", "green", "special requests", "x" * 40, "Ünïcödé"]
    s = [words[i] for i in rng.integers(0, len(words), n)]
    t = ctx.register("chain_str", pa.table({"s": pa.array(s, pa.string()), "i": pa.array(np.arange(n, dtype=np.int32))}))
    rel = t.rel().scan_filter([api.pred((0, 1), capi.F_GTE, 7)])
    got = rel.materialize([(0, 0)]).to_arrow().column(0).to_pylist()
    assert got == s[7:]
    t.release()


# ------------------------------------------------------------------ prepared plans
N_ORDERS = 150_000  # SF 0.1


@pytest.fixture(scope="module")
def world(ctx):
    db = tpch_plans.Database(ctx, N_ORDERS, 0, 1, list(range(1, 23)), False)
    return db, tpch_plans.Runner(ctx, db, 1, None, None)


def rows_of(table):
    return table.to_arrow().to_pylist()


@pytest.mark.parametrize("q", list(range(1, 23)))
def test_replayed_executions_equal_the_interpreter(ctx, world, q):
    """execution 1 records, 2 re-records what changed (statistics cached by now), 3 and 4 replay: same rows every time"""
    db, runner = world
    want = rows_of(ctx.run_plan(runner.plan_text(q), runner.plan_inputs(q)))
    plan = ctx.prepare_plan(runner.plan_text(q))
    for _ in range(4):
        assert rows_of(plan.execute(runner.plan_inputs(q))) == want
    st = plan.stats()
    assert st["misses"] == 0, st
    assert st["replays"] >= 2, st
    assert st["readbacks"] >= 1, st
    plan.release()


def test_replay_is_used_and_descriptors_are_cached(ctx, world):
    db, runner = world
    plan = ctx.prepare_plan(runner.plan_text(3))
    for _ in range(3):
        plan.execute(runner.plan_inputs(3)).release()
    before = ctx.desc_cache_stats()
    for _ in range(3):
        plan.execute(runner.plan_inputs(3)).release()
    after = ctx.desc_cache_stats()
    assert after["hits"] > before["hits"], (before, after)
    assert after["misses"] == before["misses"], "a repeated execution uploaded a descriptor again: its buffers moved"
    assert plan.stats()["replays"] >= 4
    plan.release()


def test_no_operator_keeps_a_descriptor_reference(ctx, world):
    """every operator gives its reference on a cached descriptor back when its call ends (LdbDesc, error returns included): between plans nothing
    is held, so every entry can be evicted (`desc_cache_mb` is a bound) — and no reference was ever given back twice.  The open-addressing
    builds' run-length flag (ldb_join.hip) never mis-speculated in this process either: that kind of replay miss is counted apart"""
    db, runner = world
    for q in (3, 9, 13, 18, 21):
        plan = ctx.prepare_plan(runner.plan_text(q))
        for _ in range(3):
            plan.execute(runner.plan_inputs(q)).release()
        plan.release()
    st = ctx.desc_cache_stats()
    assert st["held"] == 0 and st["underflows"] == 0, st
    t = ctx.register("t_err", pa.table({"x": pa.array([1, 2, 3], pa.int32())}))
    with pytest.raises(Exception):  # an operator that fails: still nothing held afterwards
        ctx.run_plan(json.dumps({"name": "bad", "inputs": ["t"], "steps": [{"op": "scan", "table": "t", "out": "s"},
                                 {"op": "filter", "in": "s", "preds": [{"col": "nope", "op": "LT", "value": 1}], "out": "result"}], "result": "result"}), {"t": t})
    st = ctx.desc_cache_stats()
    assert st["held"] == 0 and st["underflows"] == 0, st
    assert ctx.lib.ldb_gpu_order_dependent_misses() == 0


PLAN_COUNT = json.dumps({"name": "count_below", "inputs": ["t"], "steps": [
    {"op": "scan", "table": "t", "out": "s"}, {"op": "filter", "in": "s", "preds": [{"col": "x", "op": "LT", "value": 500}], "out": "f"},
    {"op": "materialize", "in": "f", "cols": ["i"], "out": "m"},
    {"op": "groupby", "in": "m", "aggs": [{"fn": "count_star", "as": "n"}, {"fn": "sum", "expr": "i", "as": "s"}], "out": "result"}], "result": "result"})


def _table(ctx, x):
    return ctx.register("t_replay", pa.table({"x": pa.array(x, pa.int32()), "i": pa.array(np.arange(len(x), dtype=np.int64))}))


def _expect(x):
    sel = np.nonzero(x < 500)[0]
    return [{"n": int(len(sel)), "s": int(sel.sum())}]


@pytest.mark.parametrize("grow", [False, True])
def test_data_changed_behind_the_librarys_back_is_detected(ctx, grow):
    """the trace key covers table identity, not bytes written through raw device pointers (ldb_gpu_memcpy_d2d): the replayed
    count is then wrong — with more AND with fewer passing rows than recorded — and the execution must be repeated"""
    rng = np.random.default_rng(2)
    n = 300_000
    x = rng.integers(0, 1000, n).astype(np.int32)
    y = np.where(rng.random(n) < 0.5, x, 0 if grow else 999).astype(np.int32)  # grow: more rows pass; else fewer
    t, other = _table(ctx, x), _table(ctx, y)
    plan = ctx.prepare_plan(PLAN_COUNT)
    for _ in range(3):
        assert rows_of(plan.execute({"t": t})) == _expect(x)
    assert plan.stats()["replays"] >= 1
    dst, _, _, nbytes = t.col_ptrs(0)
    src, _, _, _ = other.col_ptrs(0)
    api.check(ctx.lib.ldb_gpu_memcpy_d2d(ctx.h, dst, src, nbytes))
    assert rows_of(plan.execute({"t": t})) == _expect(y)
    st = plan.stats()
    assert st["misses"] == 1, st
    for _ in range(2):
        assert rows_of(plan.execute({"t": t})) == _expect(y)
    assert plan.stats()["misses"] == 1
    plan.release(), t.release(), other.release()


def test_rewritten_table_or_changed_option_is_not_replayed(ctx):
    rng = np.random.default_rng(4)
    n = 200_000
    x = rng.integers(0, 1000, n).astype(np.int32)
    t = _table(ctx, x)
    plan = ctx.prepare_plan(PLAN_COUNT)
    for _ in range(3):
        assert rows_of(plan.execute({"t": t})) == _expect(x)
    r0 = plan.stats()["replays"]
    assert r0 >= 1
    # the sanctioned way to change a table re-stamps it: the next execution records, no miss
    x2 = ((x.astype(np.int64) * 7) % 1000).astype(np.int32)
    api.check(ctx.lib.ldb_gpu_table_write_fixed(ctx.h, t.h, 0, x2.ctypes.data_as(C.c_void_p), x2.nbytes))
    assert rows_of(plan.execute({"t": t})) == _expect(x2)
    st = plan.stats()
    assert st["misses"] == 0 and st["replays"] == r0, st
    assert rows_of(plan.execute({"t": t})) == _expect(x2)
    assert plan.stats()["replays"] == r0 + 1
    # an option change (anything that can alter the operators' control flow) does the same
    capi.gpu_lib().ldb_gpu_set_option(b"lazy_min_rows", 1 << 20)
    assert rows_of(plan.execute({"t": t})) == _expect(x2)
    assert plan.stats()["replays"] == r0 + 1 and plan.stats()["misses"] == 0
    # another table of the same shape under the same name: its own stamp
    t2 = _table(ctx, x)
    assert rows_of(plan.execute({"t": t2})) == _expect(x)
    assert plan.stats()["misses"] == 0
    plan.release(), t.release(), t2.release()


def test_plan_replay_can_be_switched_off(ctx):
    x = np.arange(100_000, dtype=np.int32) % 1000
    t = _table(ctx, x)
    capi.gpu_lib().ldb_gpu_set_option(b"plan_replay", 0)
    try:
        plan = ctx.prepare_plan(PLAN_COUNT)
        for _ in range(3):
            assert rows_of(plan.execute({"t": t})) == _expect(x)
        assert plan.stats()["replays"] == 0
        plan.release()
    finally:
        capi.gpu_lib().ldb_gpu_set_option(b"plan_replay", 1)
    t.release()


# ------------------------------------------------------------------ write-combining radix partition (ldb_wc.hip)
@pytest.mark.parametrize("part_bytes", [1 << 10, 1 << 14, 1 << 22])
def test_radix_probe_over_the_write_combining_partition(ctx, part_bytes):
    """join_radix = 1 with a DENSE probe column into a rank / direct table: the probe keys are partitioned by table position
    with the tile-sorting scatter — one pass up to 64 partitions, two passes above (4 096 at 1 KB of table per partition) —
    and probed partition by partition: counts, pairs and the semi-join rows equal the direct probe's and numpy's, with keys
    outside the build range, duplicates and unmatched keys"""
    lib = capi.gpu_lib()
    rng = np.random.default_rng(21)
    nb, npr = 400_000, 3_000_000
    bk = rng.permutation(1_600_000)[:nb].astype(np.int32) + 1000
    pk = rng.integers(0, 1_700_000, npr).astype(np.int32)
    pk[:1000] = -5  # below the range
    b, p = ctx.register("wc_b", pa.table({"k": pa.array(bk, pa.int32())})), ctx.register("wc_p", pa.table({"k": pa.array(pk, pa.int32())}))
    pos = {int(k): i for i, k in enumerate(bk.tolist())}
    hit = np.isin(pk, bk)
    want_pairs = sorted((j, pos[int(pk[j])]) for j in np.nonzero(hit)[0].tolist())
    try:
        for rank in (1, 0):  # the rank-bitmap table, then the direct word table
            lib.ldb_gpu_set_option(b"join_rank", rank)
            ht = b.rel().join_build([(0, 0)], unique=True)
            lib.ldb_gpu_set_option(b"join_radix_part_bytes", part_bytes)
            lib.ldb_gpu_set_option(b"join_radix", 1)
            ctx.prof_enable(True)
            ctx.prof_reset()
            assert ht.probe_count(p.rel(), [(0, 0)]) == int(hit.sum())
            assert ctx.prof_all().get("k_radix_scatter", (0, 0.0))[0] >= 1, "the radix path did not run"
            r = ht.probe(p.rel(), [(0, 0)])
            assert sorted(zip(r.rowids(0).tolist(), r.rowids(1).tolist())) == want_pairs
            lo = ht.probe(p.rel(), [(0, 0)], capi.JOIN_LEFT_OUTER)  # every probe row once; the build side padded where there is no partner
            lp, lb = lo.rowids(0), lo.rowids(1)
            assert len(lp) == npr and sorted(lp.tolist()) == list(range(npr))
            assert sorted((int(a), int(b_)) for a, b_ in zip(lp.tolist(), lb.tolist()) if b_ != 0xFFFFFFFF) == want_pairs
            for lds in (0,):  # the partitions probed from the cache instead of LDS
                lib.ldb_gpu_set_option(b"join_radix_lds", lds)
                assert ht.probe_count(p.rel(), [(0, 0)]) == int(hit.sum())
                r2 = ht.probe(p.rel(), [(0, 0)])
                assert sorted(zip(r2.rowids(0).tolist(), r2.rowids(1).tolist())) == want_pairs
                lib.ldb_gpu_set_option(b"join_radix_lds", 1)
            for wc in (0,):  # the one-pass cursor scatter as the control
                lib.ldb_gpu_set_option(b"join_radix_wc", wc)
                assert ht.probe_count(p.rel(), [(0, 0)]) == int(hit.sum())
                lib.ldb_gpu_set_option(b"join_radix_wc", 1)
            lib.ldb_gpu_set_option(b"join_radix", 0)
            assert ht.probe_count(p.rel(), [(0, 0)]) == int(hit.sum())
            ht.release()
    finally:
        lib.ldb_gpu_set_option(b"join_radix", -1)  # the default: auto
        lib.ldb_gpu_set_option(b"join_radix_wc", 1)
        lib.ldb_gpu_set_option(b"join_radix_lds", 1)
        lib.ldb_gpu_set_option(b"join_radix_part_bytes", 1 << 20)
        lib.ldb_gpu_set_option(b"join_rank", 1)
        ctx.prof_enable(False)
    b.release(), p.release()
