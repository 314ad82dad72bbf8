"""More join shapes against the oracle: build-side semi / anti joins (the reference's reverseSides
scheme), joins through filtered relations on both sides, multi-way join composition."""
import numpy as np
import pyarrow as pa
import pytest

from lingodb_amd import api, capi
from oracle_bind import HostRel, HostTable
import tpch_data

pytestmark = pytest.mark.gpu
N_ORDERS = 21007


@pytest.fixture(scope="module")
def data(ctx):
    li = tpch_data.host_table(tpch_data.LINEITEM, N_ORDERS)
    od = tpch_data.host_table(tpch_data.ORDERS, N_ORDERS)
    return {"li": li, "od": od, "gli": ctx.register("li2", li), "god": ctx.register("od2", od), "hli": HostTable(li), "hod": HostTable(od)}


@pytest.mark.parametrize("kind", [capi.JOIN_SEMI_BUILD, capi.JOIN_ANTI_BUILD])
def test_build_side_semi_anti_q4_shape(ctx, oracle, data, kind):
    """TPC-H Q4 shape: orders in a quarter that have (no) lineitem with l_commitdate < l_receiptdate"""
    opred = [api.pred((0, 4), capi.F_GTE, 8582), api.pred((0, 4), capi.F_LT, 8674)]
    lpred = [api.pred((0, 11), capi.F_LT, rhs_col=(0, 12))]
    ho = data["hod"].rel().select(oracle.scan_filter(data["hod"].rel(), opred))
    hl = data["hli"].rel().select(oracle.scan_filter(data["hli"].rel(), lpred))
    want, _, _ = oracle.join(ho, [(0, 0)], hl, [(0, 0)], kind, threads=2)
    go = data["god"].rel().scan_filter(opred)
    gl = data["gli"].rel().scan_filter(lpred)
    out = go.join_build([(0, 0)], unique=True).probe(gl, [(0, 0)], kind)
    assert out.sides == 1
    assert np.array_equal(out.rowids(0), ho.phys(0)[want])  # physical order rows, ascending build order


def test_build_side_semi_with_duplicate_build_keys(ctx, oracle):
    rng = np.random.default_rng(3)
    b = pa.table({"k": pa.array(rng.integers(0, 300, 5000), pa.int64()), "s": pa.array([["x", "a much longer string key"][i] for i in rng.integers(0, 2, 5000)])})
    p = pa.table({"k": pa.array(rng.integers(100, 500, 7000), pa.int64()), "s": pa.array([["x", "a much longer string key"][i] for i in rng.integers(0, 2, 7000)])})
    gb, gp, hb, hp = ctx.register("b3", b).rel(), ctx.register("p3", p).rel(), HostTable(b).rel(), HostTable(p).rel()
    for keys in ([(0, 0)], [(0, 0), (0, 1)]):
        for kind in (capi.JOIN_SEMI_BUILD, capi.JOIN_ANTI_BUILD):
            want, _, _ = oracle.join(hb, keys, hp, keys, kind)
            assert np.array_equal(gb.join_build(keys).probe(gp, keys, kind).rowids(0), want)


@pytest.mark.parametrize("layout", ["int32 keys (direct / chained)", "int64 keys (hashed)", "row-id probe side"])
def test_build_side_semi_and_anti_in_one_pass_q21_shape(ctx, layout):
    """ldb_gpu_join_probe_semi_anti_build (TPC-H Q21: EXISTS l2 AND NOT EXISTS late l3, both with l_suppkey <> l1.l_suppkey): the build rows with a
    partner among the probe rows and none among the probe rows that also pass the extra conjunct — against plain numpy, and equal to the two-step
    form (SEMI_BUILD, a table over its result, ANTI_BUILD probed by the filtered probe side); duplicate build keys, NULL-free integer keys"""
    rng = np.random.default_rng(2121)
    nb, npr = 40_000, 400_000
    kt = pa.int64() if layout.startswith("int64") else pa.int32()
    bk, bs = rng.integers(0, 30_000, nb), rng.integers(0, 6, nb)
    pk, ps = rng.integers(0, 33_000, npr), rng.integers(0, 6, npr)
    pa_, pb_ = rng.integers(0, 100, npr), rng.integers(0, 100, npr)  # "late" = a > b
    b = ctx.register("q21_build", pa.table({"k": pa.array(bk, kt), "s": pa.array(bs, pa.int32())}))
    p = ctx.register("q21_probe", pa.table({"k": pa.array(pk, kt), "s": pa.array(ps, pa.int32()), "a": pa.array(pa_, pa.int32()), "b": pa.array(pb_, pa.int32())}))
    prel = p.rel()
    keep = np.arange(npr)
    if layout.startswith("row-id"):
        prel = prel.scan_filter([api.pred((0, 3), capi.F_LT, 90)])  # materialised row ids on the probe side
        keep = np.nonzero(pb_ < 90)[0]
    # numpy: per key the set of supplier values among all / among late probe rows
    late = pa_ > pb_
    all_by_key, late_by_key = {}, {}
    for i in keep:
        all_by_key.setdefault(int(pk[i]), set()).add(int(ps[i]))
        if late[i]:
            late_by_key.setdefault(int(pk[i]), set()).add(int(ps[i]))
    want = [j for j in range(nb) if (all_by_key.get(int(bk[j]), set()) - {int(bs[j])}) and not (late_by_key.get(int(bk[j]), set()) - {int(bs[j])})]
    assert 0 < len(want) < nb
    resid = [((0, 1), capi.F_NEQ, (0, 1))]
    anti = [api.pred((0, 2), capi.F_GT, rhs_col=(0, 3))]
    ht = b.rel().join_build([(0, 0)])
    got = ht.probe_semi_anti_build(prel, [(0, 0)], anti, residual=resid)
    assert got.sides == 1 and np.array_equal(got.rowids(0), np.array(want, dtype=np.uint32))
    # the two-step form
    l2 = ht.probe(prel, [(0, 0)], capi.JOIN_SEMI_BUILD, residual=resid)
    late_rel = p.rel().scan_filter(([api.pred((0, 3), capi.F_LT, 90)] if layout.startswith("row-id") else []) + anti)
    l3 = l2.join_build([(0, 0)]).probe(late_rel, [(0, 0)], capi.JOIN_ANTI_BUILD, residual=resid)
    assert np.array_equal(l3.rowids(0), got.rowids(0))


def test_ordered_slots_fall_back_to_hashing_on_clustered_keys(ctx, oracle):
    """KEY32 tables spread their slots over [min key, max key]; one outlier key squeezes all others
    into a handful of slots → the build sees long probe runs and must rebuild hashed.  Results are
    the same either way (and equal to the oracle's), including probe keys outside the key range."""
    rng = np.random.default_rng(5)
    dense = rng.permutation(200_000)[:60_000].astype(np.int32)
    bk = np.concatenate([dense, np.array([2**31 - 1, -(2**31)], dtype=np.int32)])
    pk = np.concatenate([rng.integers(-1000, 250_000, 90_000), np.array([2**31 - 1, -(2**31), 2**31 - 2])]).astype(np.int32)
    b, p = pa.table({"k": pa.array(bk, pa.int32())}), pa.table({"k": pa.array(pk, pa.int32())})
    gb, gp, hb, hp = ctx.register("ob", b).rel(), ctx.register("op", p).rel(), HostTable(b).rel(), HostTable(p).rel()
    ht = gb.join_build([(0, 0)], unique=True)
    for kind in (capi.JOIN_INNER, capi.JOIN_SEMI, capi.JOIN_ANTI, capi.JOIN_LEFT_OUTER):
        op, ob, _ = oracle.join(hb, [(0, 0)], hp, [(0, 0)], kind)
        out = ht.probe(gp, [(0, 0)], kind)
        if kind in (capi.JOIN_SEMI, capi.JOIN_ANTI):
            assert np.array_equal(out.rowids(0), op)
        else:
            assert sorted(zip(out.rowids(0).tolist(), out.rowids(1).tolist())) == sorted(zip(op.tolist(), ob.tolist()))
    # and a well-spread key set of the same size keeps ordered slots: same answers again
    b2 = pa.table({"k": pa.array(dense * 7, pa.int32())})
    g2, h2 = ctx.register("ob2", b2).rel(), HostTable(b2).rel()
    op, ob, _ = oracle.join(h2, [(0, 0)], hp, [(0, 0)], capi.JOIN_INNER)
    out = g2.join_build([(0, 0)], unique=True).probe(gp, [(0, 0)], capi.JOIN_INNER)
    assert sorted(zip(out.rowids(0).tolist(), out.rowids(1).tolist())) == sorted(zip(op.tolist(), ob.tolist()))


def test_duplicate_heavy_build_keys_use_chains(ctx, oracle):
    """a build key that repeats tens of thousands of times: one slot per row would make the build
    quadratic, so the table is rebuilt chained (one slot per distinct key, rows linked) — every
    join kind must still agree with the oracle, for integer and for string keys"""
    rng = np.random.default_rng(13)
    nb = 120_000
    bk = rng.integers(0, 5, nb).astype(np.int32)
    names = np.array(["alpha", "beta", "a considerably longer key string", "delta", "epsilon"])
    b = pa.table({"k": pa.array(bk, pa.int32()), "s": pa.array(names[bk])})
    pk = np.array([0, 3, 7, 4, 4, -1], dtype=np.int32)
    p = pa.table({"k": pa.array(pk, pa.int32()), "s": pa.array(["alpha", "delta", "nope", "epsilon", "epsilon", "zeta"])})
    gb, gp, hb, hp = ctx.register("dupb", b).rel(), ctx.register("dupp", p).rel(), HostTable(b).rel(), HostTable(p).rel()
    for keys in ([(0, 0)], [(0, 1)], [(0, 0), (0, 1)]):
        ht = gb.join_build(keys)
        for kind in (capi.JOIN_INNER, capi.JOIN_LEFT_OUTER):
            op, ob, _ = oracle.join(hb, keys, hp, keys, kind)
            out = ht.probe(gp, keys, kind)
            got = np.stack([out.rowids(0), out.rowids(1)], axis=1)
            want = np.stack([op, ob], axis=1)
            assert got.shape == want.shape
            assert np.array_equal(got[np.lexsort((got[:, 1], got[:, 0]))], want[np.lexsort((want[:, 1], want[:, 0]))]), (keys, kind)
        for kind in (capi.JOIN_SEMI, capi.JOIN_ANTI, capi.JOIN_SEMI_BUILD, capi.JOIN_ANTI_BUILD):
            want, _, _ = oracle.join(hb, keys, hp, keys, kind)
            assert np.array_equal(ht.probe(gp, keys, kind).rowids(0), want), (keys, kind)
        assert ht.probe_count(gp, keys) == len(oracle.join(hb, keys, hp, keys, capi.JOIN_INNER)[0])


def test_three_way_join_composition(ctx, oracle, data):
    """(lineitem ⋈ orders) result used as a build side again: row ids compose through both joins"""
    cu = tpch_data.host_table(tpch_data.CUSTOMER, N_ORDERS)
    gcu, hcu = ctx.register("cu2", cu), HostTable(cu)
    # orders ⋈ customer on custkey, then lineitem ⋈ that on orderkey
    op, ob, _ = oracle.join(hcu.rel(), [(0, 0)], data["hod"].rel(), [(0, 1)], capi.JOIN_INNER)
    hoc = HostRel([(data["hod"], op), (hcu, ob)], len(op))
    lp, lb, _ = oracle.join(hoc, [(0, 0)], data["hli"].rel(), [(0, 0)], capi.JOIN_INNER)
    want = sorted(zip(lp.tolist(), hoc.phys(0)[lb].tolist(), hoc.phys(1)[lb].tolist()))
    oc = gcu.rel().join_build([(0, 0)], unique=True).probe(data["god"].rel(), [(0, 1)], capi.JOIN_INNER)
    loc = oc.join_build([(0, 0)], unique=True).probe(data["gli"].rel(), [(0, 0)], capi.JOIN_INNER)
    assert loc.sides == 3
    got = sorted(zip(loc.rowids(0).tolist(), loc.rowids(1).tolist(), loc.rowids(2).tolist()))
    assert got == want
    # and values gathered through the composed ids agree with the host tables
    t = loc.materialize([(0, 0), (1, 0), (2, 0), (1, 1)]).to_arrow()
    assert t.column(0).to_pylist() == t.column(1).to_pylist()  # l_orderkey == o_orderkey
    assert t.column(2).to_pylist() == t.column(3).to_pylist()  # c_custkey == o_custkey


def test_radix_clustered_probe_gives_the_direct_probe_results(ctx, oracle, data):
    """join_radix = 1 clusters the probe side by slot range before probing (north_star's radix-partitioned
    probe; off by default, DESIGN.md §2 Join): same pairs, rows and counts as the direct probe and as the
    oracle — unique and duplicated build keys, a filtered (row-id) probe side, outer-join padding, build-side
    semi / anti — with partitions small enough that the test tables span many of them."""
    lib = capi.gpu_lib()
    opred = [api.pred((0, 4), capi.F_GTE, 8582), api.pred((0, 4), capi.F_LT, 9300)]
    lpred = [api.pred((0, 10), capi.F_GTE, 8800)]
    ho = data["hod"].rel().select(oracle.scan_filter(data["hod"].rel(), opred))
    hl = data["hli"].rel().select(oracle.scan_filter(data["hli"].rel(), lpred))

    def run():
        out = {}
        go = data["god"].rel().scan_filter(opred)
        gl = data["gli"].rel().scan_filter(lpred)
        gl.rows  # force the filter: a row-id probe side
        hu = go.join_build([(0, 0)], unique=True)
        out["count"] = hu.probe_count(gl, [(0, 0)])
        r = hu.probe(gl, [(0, 0)])
        out["inner"] = sorted(zip(r.rowids(0).tolist(), r.rowids(1).tolist()))
        r = hu.probe(gl, [(0, 0)], capi.JOIN_LEFT_OUTER)
        out["left_outer"] = sorted(zip(r.rowids(0).tolist(), r.rowids(1).tolist()))
        out["semi_build"] = hu.probe(gl, [(0, 0)], capi.JOIN_SEMI_BUILD).rowids(0).tolist()
        out["anti_build"] = hu.probe(gl, [(0, 0)], capi.JOIN_ANTI_BUILD).rowids(0).tolist()
        hd = gl.join_build([(0, 0)])  # duplicated build keys: lineitems per order; orders probe
        r = hd.probe(go, [(0, 0)])
        out["pairs"] = sorted(zip(r.rowids(0).tolist(), r.rowids(1).tolist()))
        return out

    try:
        lib.ldb_gpu_set_option(b"join_radix_part_bytes", 4096)
        lib.ldb_gpu_set_option(b"join_radix", 0)
        direct = run()
        lib.ldb_gpu_set_option(b"join_radix", 1)
        radix = run()
    finally:
        lib.ldb_gpu_set_option(b"join_radix", -1)  # the default: auto
        lib.ldb_gpu_set_option(b"join_radix_part_bytes", 1 << 20)
    assert radix == direct
    want, wb, _ = oracle.join(ho, [(0, 0)], hl, [(0, 0)], capi.JOIN_INNER, threads=2)
    assert len(direct["inner"]) == len(want) == direct["count"] and len(want) > 1000
    assert direct["inner"] == sorted(zip(hl.phys(0)[want].tolist(), ho.phys(0)[wb].tolist()))
    assert len(direct["left_outer"]) == len(hl.phys(0))


@pytest.mark.parametrize("table", ["rank+coarse", "rank", "direct", "open"])
def test_selective_unique_build_large_probe_every_layout(ctx, table):
    """a 0.2 % build side over a 2 M key range probed by 5 M unfiltered rows: the shape that gets the rank-bitmap table with
    the LDS-resident coarse key bitmap in front of it (512-thread launches, specialised kernels) — every probe kind against
    numpy, and the same through the other table layouts (join_coarse / join_rank / join_direct switched off in turn)"""
    lib = capi.gpu_lib()
    rng = np.random.default_rng(17)
    keyspace = 2_000_000
    bkeys = np.sort(rng.choice(keyspace, 4000, replace=False)).astype(np.int32) + 1000
    if table != "rank+coarse":  # an unsorted build side: rank → row through the permutation
        bkeys = rng.permutation(bkeys)
    pkeys = rng.integers(0, keyspace + 3000, 5_000_000).astype(np.int32)
    pkeys[::1000] = bkeys[rng.integers(0, len(bkeys), len(pkeys[::1000]))]  # guaranteed hits
    b = ctx.register("sel_b_" + table.replace("+", "_"), pa.table({"k": pa.array(bkeys), "v": pa.array(np.arange(len(bkeys), dtype=np.int64))}))
    p = ctx.register("sel_p_" + table.replace("+", "_"), pa.table({"k": pa.array(pkeys)}))
    opts = {"rank+coarse": {}, "rank": {"join_coarse": 0}, "direct": {"join_rank": 0}, "open": {"join_direct": 0}}[table]
    for k, v in opts.items():
        lib.ldb_gpu_set_option(k.encode(), v)
    lib.ldb_gpu_set_option(b"debug_check", 1)
    try:
        ht = b.rel().join_build([(0, 0)], unique=True)
        pos = {int(k): i for i, k in enumerate(bkeys.tolist())}
        hit = np.isin(pkeys, bkeys)
        rows = np.nonzero(hit)[0]
        want_build = np.array([pos[int(k)] for k in pkeys[rows]], dtype=np.uint32)
        assert ht.probe_count(p.rel(), [(0, 0)]) == len(rows)
        inner = ht.probe(p.rel(), [(0, 0)])
        assert np.array_equal(inner.rowids(0), rows) and np.array_equal(inner.rowids(1), want_build)
        assert np.array_equal(ht.probe(p.rel(), [(0, 0)], capi.JOIN_SEMI).rowids(0), rows)
        assert np.array_equal(ht.probe(p.rel(), [(0, 0)], capi.JOIN_ANTI).rowids(0), np.nonzero(~hit)[0])
        assert np.array_equal(ht.probe(p.rel(), [(0, 0)], capi.JOIN_SEMI_BUILD).rowids(0), np.unique(want_build))
        lo = ht.probe(p.rel(), [(0, 0)], capi.JOIN_LEFT_OUTER)
        assert lo.rows == len(pkeys) and int((lo.rowids(1) != 0xFFFFFFFF).sum()) == len(rows)
    finally:
        for k in opts:
            lib.ldb_gpu_set_option(k.encode(), 1)
        lib.ldb_gpu_set_option(b"debug_check", 0)


@pytest.mark.parametrize("fine", [1, 0])
def test_dense_unique_build_gets_the_fine_lds_filter(ctx, fine):
    """round 6: a 5 % build side over an 8 M key range — too dense for 64-key blocks to be empty — probed by 6 M unfiltered rows: the rank table with
    one filter bit per SIXTEEN key values in LDS (62 KB here, up to 156 KB: one 1024-thread workgroup per CU, more than 64 KB of dynamic LDS per
    workgroup), every probe kind against numpy; the same with the fine filter switched off (join_coarse_fine = 0)"""
    lib = capi.gpu_lib()
    rng = np.random.default_rng(29)
    keyspace = 8_000_000
    bkeys = np.sort(rng.choice(keyspace, keyspace // 20, replace=False)).astype(np.int32) + 77
    pkeys = rng.integers(0, keyspace + 5000, 6_000_000).astype(np.int32)
    b = ctx.register("fine_b_%d" % fine, pa.table({"k": pa.array(bkeys)}))
    p = ctx.register("fine_p_%d" % fine, pa.table({"k": pa.array(pkeys)}))
    lib.ldb_gpu_set_option(b"join_coarse_fine", fine)
    lib.ldb_gpu_set_option(b"debug_check", 1)
    try:
        ht = b.rel().join_build([(0, 0)], unique=True)
        idx = np.searchsorted(bkeys, pkeys)
        hit = (idx < len(bkeys)) & (bkeys[np.minimum(idx, len(bkeys) - 1)] == pkeys)
        rows = np.nonzero(hit)[0]
        assert 200_000 < len(rows) < 400_000
        assert ht.probe_count(p.rel(), [(0, 0)]) == len(rows)
        assert np.array_equal(ht.probe(p.rel(), [(0, 0)], capi.JOIN_SEMI).rowids(0), rows)
        assert np.array_equal(ht.probe(p.rel(), [(0, 0)], capi.JOIN_ANTI).rowids(0), np.nonzero(~hit)[0])
        inner = ht.probe(p.rel(), [(0, 0)])
        assert np.array_equal(inner.rowids(0), rows) and np.array_equal(inner.rowids(1), idx[rows].astype(np.uint32))
    finally:
        lib.ldb_gpu_set_option(b"join_coarse_fine", 1)
        lib.ldb_gpu_set_option(b"debug_check", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("keyspace,share", [(1_000_000, 25), (400_000, 12), (100_000, 4)])
def test_small_key_range_gets_the_finest_lds_filter_that_fits(ctx, keyspace, share):
    """round 6: where the key range is small enough (a supplier-sized table), the LDS key filter takes the finest granularity that fits 16 KB — 8, 4 … 1
    key values per bit (1 = the presence bits themselves) — and rides along with the filtered (tile) probes too: unfiltered and filtered probes of every
    kind against numpy, with the row-id checks on; the same with the option off (16 keys per bit)"""
    lib = capi.gpu_lib()
    rng = np.random.default_rng(31 + share)
    bkeys = np.sort(rng.choice(keyspace, keyspace // share, replace=False)).astype(np.int32) + 5
    pkeys = rng.integers(0, keyspace + 300, 5_000_000).astype(np.int32)
    flag = rng.integers(0, 4, len(pkeys)).astype(np.int32)
    b = ctx.register("finest_b_%d" % keyspace, pa.table({"k": pa.array(bkeys)}))
    p = ctx.register("finest_p_%d" % keyspace, pa.table({"k": pa.array(pkeys), "f": pa.array(flag)}))
    idx = np.searchsorted(bkeys, pkeys)
    hit = (idx < len(bkeys)) & (bkeys[np.minimum(idx, len(bkeys) - 1)] == pkeys)
    lib.ldb_gpu_set_option(b"debug_check", 1)
    try:
        for finest in (1, 0):
            lib.ldb_gpu_set_option(b"join_coarse_finest", finest)
            ht = b.rel().join_build([(0, 0)], unique=True)
            rows = np.nonzero(hit)[0]
            assert ht.probe_count(p.rel(), [(0, 0)]) == len(rows)
            assert np.array_equal(ht.probe(p.rel(), [(0, 0)], capi.JOIN_SEMI).rowids(0), rows)
            assert np.array_equal(ht.probe(p.rel(), [(0, 0)], capi.JOIN_ANTI).rowids(0), np.nonzero(~hit)[0])
            inner = ht.probe(p.rel(), [(0, 0)])
            assert np.array_equal(inner.rowids(0), rows) and np.array_equal(inner.rowids(1), idx[rows].astype(np.uint32))
            lazy = p.rel().scan_filter([api.pred((0, 1), capi.F_EQ, 2)])  # >= 1 M rows: stays lazy, the probe runs the tile kernels
            frows = np.nonzero(hit & (flag == 2))[0]
            finner = ht.probe(lazy, [(0, 0)])
            assert np.array_equal(finner.rowids(0), frows) and np.array_equal(finner.rowids(1), idx[frows].astype(np.uint32))
            assert np.array_equal(ht.probe(lazy, [(0, 0)], capi.JOIN_SEMI).rowids(0), frows)
            ht.release()
    finally:
        lib.ldb_gpu_set_option(b"join_coarse_finest", 1)
        lib.ldb_gpu_set_option(b"debug_check", 0)
