"""Round 6: a scan over a short input is cut finer than one workgroup per 16 384 rows (gridDim.y = 2 … 16 workgroups share a zone's 256 bitmap words;
csrc/ldb_scan_kernel.h, scan_run_with in csrc/ldb_scan.hip) — conjunctions, a LIKE, a DNF and the expansion to row ids against numpy at sizes on both
sides of every split factor, and the same with the split switched off."""
import numpy as np
import pyarrow as pa
import pytest

from lingodb_amd import api, capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 63, 1025, 16384, 16385, 100_003, 1_000_000, 3_100_000, 20_000_001, 40_000_000])
def test_split_scans_equal_numpy(ctx, n):
    lib = capi.gpu_lib()
    rng = np.random.default_rng(n % 9973)
    a = rng.integers(0, 1000, n).astype(np.int32)
    b = rng.integers(0, 50, n).astype(np.int64)
    t = ctx.register("split_%d" % n, pa.table({"a": pa.array(a), "b": pa.array(b)}))
    want = np.nonzero((a < 300) & (b >= 10))[0]
    want_dnf = np.nonzero(((a < 20) & (b == 3)) | ((a >= 990) & (b < 25)))[0]
    try:
        for split in (1, 0):
            lib.ldb_gpu_set_option(b"scan_split", split)
            lib.ldb_gpu_set_option(b"lazy_filter", 0)  # the scan kernel itself, not a consumer's fused filter
            got = t.rel().scan_filter([api.pred((0, 0), capi.F_LT, 300), api.pred((0, 1), capi.F_GTE, 10)]).rowids(0)
            assert np.array_equal(got, want), (n, split)
            got = t.rel().scan_filter_dnf([[api.pred((0, 0), capi.F_LT, 20), api.pred((0, 1), capi.F_EQ, 3)],
                                           [api.pred((0, 0), capi.F_GTE, 990), api.pred((0, 1), capi.F_LT, 25)]]).rowids(0)
            assert np.array_equal(got, want_dnf), (n, split)
    finally:
        lib.ldb_gpu_set_option(b"scan_split", 1)
        lib.ldb_gpu_set_option(b"lazy_filter", 1)


@pytest.mark.parametrize("n", [70_000, 1_000_000])
def test_like_over_a_short_column_is_specialised_and_split(ctx, n):
    """Q16's shape: a two-segment LIKE over about a million comments — specialised from 64 K rows on (`jit_min_rows_like`), cut into 976 workgroups"""
    rng = np.random.default_rng(5)
    words = np.array(["Customer", "Complaints", "final", "deposits", "slyly", "regular", "ironic", "accounts", "requests", "special"])
    strs = [" ".join(words[rng.integers(0, len(words), rng.integers(3, 9))]) for _ in range(n // 50)]
    col = [strs[i] for i in rng.integers(0, len(strs), n)]
    t = ctx.register("split_like_%d" % n, pa.table({"s": pa.array(col, pa.string())}))
    has = np.array([("Customer" in s) and ("Complaints" in s[s.index("Customer") + 8:]) for s in strs])
    lookup = {s: h for s, h in zip(strs, has)}
    want = np.nonzero(np.array([lookup[s] for s in col]))[0]
    assert 0 < len(want) < n
    for op, w in ((capi.F_LIKE, want), (capi.F_NOT_LIKE, np.setdiff1d(np.arange(n), want))):
        got = t.rel().scan_filter([api.pred((0, 0), op, "%Customer%Complaints%")]).rowids(0)
        assert np.array_equal(got, w), (n, op)
