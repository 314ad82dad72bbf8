"""f3: zone maps.  A sorted (or clustered) integer-like column gets min / max per 16 384 rows on its first range
predicate; rows of zones that cannot match fail without the column being loaded.  Same rows as numpy and as the run with
zone_maps = 0 for every comparison operator, through the materialising scan, the count-only scan, a fused group-by and a
fused join probe; a scattered column keeps no zones; NULLable and small columns are left alone."""
import numpy as np
import pyarrow as pa
import pytest

from lingodb_amd import api, capi

pytestmark = pytest.mark.gpu
N = 3_000_000


@pytest.fixture(scope="module")
def tables(ctx):
    rng = np.random.default_rng(77)
    sorted_k = np.sort(rng.integers(0, 50_000_000, N)).astype(np.int64)
    clustered = (np.arange(N) // 1000 * 37 + rng.integers(0, 5000, N)).astype(np.int32)  # local noise around a trend
    scattered = rng.integers(0, 1_000_000, N).astype(np.int32)
    dates = (8000 + np.arange(N) // 1500).astype(np.int32)
    v = rng.integers(0, 1000, N).astype(np.int64)
    t = pa.table({"s": pa.array(sorted_k), "c": pa.array(clustered), "r": pa.array(scattered), "d": pa.array(dates, pa.int32()).cast(pa.date32()), "v": pa.array(v),
                  "nul": pa.array([None if i % 1000 == 0 else int(x) for i, x in enumerate(sorted_k[:N])], pa.int64())})
    return ctx.register("zoned", t), {"s": sorted_k, "c": clustered, "r": scattered, "d": dates, "v": v}


def test_zones_are_kept_only_where_they_select(ctx, tables):
    g, h = tables
    lib = capi.gpu_lib()
    nz = (N + 16383) // 16384
    assert lib.ldb_gpu_table_zones(ctx.h, g.h, 0) == nz  # sorted
    assert lib.ldb_gpu_table_zones(ctx.h, g.h, 1) == nz  # clustered
    assert lib.ldb_gpu_table_zones(ctx.h, g.h, 2) == 0  # scattered: every zone spans the whole range
    assert lib.ldb_gpu_table_zones(ctx.h, g.h, 3) == nz  # date32
    assert lib.ldb_gpu_table_zones(ctx.h, g.h, 5) == 0  # has NULLs
    small = ctx.register("small", pa.table({"k": pa.array(np.arange(1000, dtype=np.int64))}))
    assert lib.ldb_gpu_table_zones(ctx.h, small.h, 0) == 0


OPS = [(capi.F_EQ, np.equal), (capi.F_NEQ, np.not_equal), (capi.F_LT, np.less), (capi.F_LTE, np.less_equal), (capi.F_GT, np.greater), (capi.F_GTE, np.greater_equal)]


@pytest.mark.parametrize("lazy", [0, 1])
def test_filters_with_zone_maps_equal_numpy_and_the_unzoned_run(ctx, tables, lazy):
    g, h = tables
    lib = capi.gpu_lib()
    lib.ldb_gpu_set_option(b"lazy_min_rows", 0 if lazy else 1 << 40)
    lib.ldb_gpu_set_option(b"zone_min_rows", 0)
    try:
        for col, name in ((0, "s"), (1, "c"), (3, "d")):
            vals = h[name]
            consts = [int(vals[N // 3]), int(vals.min()) - 1, int(vals.max()) + 1, int(vals[-1]), int(vals[0])]
            for op, fn in OPS:
                for c in consts:
                    preds = [api.pred((0, col), op, c), api.pred((0, 4), capi.F_LT, 900)]
                    want = np.nonzero(fn(vals, c) & (h["v"] < 900))[0]
                    got = {}
                    for zm in (1, 0):
                        lib.ldb_gpu_set_option(b"zone_maps", zm)
                        r = g.rel().scan_filter(preds)
                        # fused consumers (lazy) and the materialising scan (not lazy) see the same rows
                        cnt = r.groupby([], [api.agg(capi.AGG_COUNT_STAR), api.agg(capi.AGG_SUM, api.col_expr((0, 4)))], est_groups=1).to_arrow().to_pylist()[0]
                        got[zm] = (g.rel().scan_filter(preds).rows, tuple(cnt.values()))
                    assert got[1] == got[0] == (len(want), (len(want), int(h["v"][want].sum()) if len(want) else None)), (name, op, c)
        # a two-sided range on the sorted column, fused into a join probe
        lo, hi = int(h["s"][N // 2]), int(h["s"][N // 2 + 40000])
        build = ctx.register("b", pa.table({"k": pa.array(np.arange(0, 1000, dtype=np.int64))}))
        ht = build.rel().join_build([(0, 0)], unique=True)
        for zm in (1, 0):
            lib.ldb_gpu_set_option(b"zone_maps", zm)
            r = g.rel().scan_filter([api.pred((0, 0), capi.F_GTE, lo), api.pred((0, 0), capi.F_LT, hi)])
            m = ht.probe_count(r, [(0, 4)])
            sel = (h["s"] >= lo) & (h["s"] < hi)
            assert m == int(sel.sum())  # every v is in [0, 1000)
    finally:
        lib.ldb_gpu_set_option(b"zone_maps", 1)
        lib.ldb_gpu_set_option(b"zone_min_rows", 1 << 20)
        lib.ldb_gpu_set_option(b"lazy_min_rows", 1 << 20)
