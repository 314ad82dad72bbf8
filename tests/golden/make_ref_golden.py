"""Generates the golden fixtures of this directory by running the REFERENCE'S OWN runtime objects
(oracle/_ref/libldb_ref.so = Hash.cpp, Restrictions.cpp, LazyJoinHashtable.cpp,
PreAggregationHashtable.cpp, StringRuntime.cpp, DateRuntime.cpp compiled in place from
/root/reference by oracle/ref_build/build_ref.sh) on seeded inputs.  The reference tree does not
exist on the GPU box, so its answers travel as these files:

  ref_types.arrow          input table, one column per physical type the scan can load (incl. NULLs)
  ref_types_hash.npz       dbHashApplyColumn folded over key lists of that table      (§8 a5)
  ref_filters.json/.npz    Restrictions::applyFilters row ids over the generator's lineitem  (a3)
  ref_join.npz             HashIndexedView build + probe, (probe row, build row) pairs  (a7, a8)
  ref_groupby.npz          PreAggregationHashtable fragment insert + merge, (key, sum, count)  (a9, a10)
  ref_like.json            StringRuntime::like answers
  ref_extract_year.json    DateRuntime::extractYear answers
  ref_substr.json          StringRuntime::substr answers
  ref_sort.npz             GrowingBuffer::sort / parallelSort permutations (multi-key, DESC), Heap top-k prefixes,
                           SimpleState SUM + COUNT, generic Hashtable group-by                (a11, a12, a13, a14)

Run from the repo root where /root/reference exists:  python tests/golden/make_ref_golden.py
Consumers: tests/test_golden_fixtures.py (oracle, CPU) and tests/test_gpu_z_golden.py (HIP path)."""
import ctypes as C
import decimal
import json
import os
import sys

import numpy as np
import pyarrow as pa

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "lingo-db_amd"), os.path.join(ROOT, "tests"), ROOT]

import oracle_bind  # noqa: E402
import test_oracle_vs_ref as tv  # noqa: E402  (binding helpers of the reference library)
import tpch_data  # noqa: E402

HASH_KEY_LISTS = [[c] for c in range(12)] + [[2, 10, 4], [6, 3], [10, 10, 0], [11, 3], [7, 11, 5]]
FILTER_ORDERS = 3000
FILTER_CASES = [
    [{"col": "l_shipdate", "op": "LTE", "v": "1998-09-02"}],
    [{"col": "l_shipdate", "op": "GTE", "v": "1994-01-01"}, {"col": "l_shipdate", "op": "LT", "v": "1995-01-01"},
     {"col": "l_discount", "op": "GTE", "v": "0.05"}, {"col": "l_discount", "op": "LTE", "v": "0.07"}, {"col": "l_quantity", "op": "LT", "v": 24}],
    [{"col": "l_returnflag", "op": "EQ", "v": "R"}, {"col": "l_shipmode", "op": "IN", "in": ["MAIL", "SHIP"]}],
    [{"col": "l_shipmode", "op": "LT", "v": "RAIL"}, {"col": "l_shipinstruct", "op": "NEQ", "v": "NONE"}, {"col": "l_linenumber", "op": "IN", "in": [1, 3, 7]}],
    [{"col": "l_quantity", "op": "GT", "v": "49.99"}, {"col": "l_commitdate", "op": "NEQ", "v": "1995-03-15"}],
    [{"col": "l_orderkey", "op": "GTE", "v": 40000}, {"col": "l_linestatus", "op": "NEQ", "v": "F"}, {"col": "l_tax", "op": "EQ", "v": "0.08"}],
    [{"col": "l_shipmode", "op": "GTE", "v": "TRUCK"}],
    [{"col": "l_orderkey", "op": "LT", "v": 0}],
]


def tv_table_hashes(ora, keys):
    """db.hash of an int64 key column through the oracle (pinned against the reference's Hash.cpp above)"""
    t = oracle_bind.HostTable(pa.table({"k": pa.array(keys, pa.int64())}))
    return ora.hash_keys(t.rel(), [(0, 0)])


def types_table():
    rng = np.random.default_rng(20260925)
    n = 400
    strs = ["", "a", "abcdefghijkl", "abcdefghijklm", "betaggamaetanetalambda", "x" * 40, "Customer#000000001", "é€ß"]
    return pa.table({
        "i8": pa.array(rng.integers(-128, 127, n), pa.int8()),
        "i16": pa.array(rng.integers(-30000, 30000, n), pa.int16()),
        "i32": pa.array([None if x % 17 == 0 else int(x) for x in rng.integers(-2 ** 31, 2 ** 31 - 1, n)], pa.int32()),
        "i64": pa.array(rng.integers(-2 ** 62, 2 ** 62, n), pa.int64()),
        "d32": pa.array(rng.integers(-1000, 20000, n).astype(np.int32), pa.int32()).cast(pa.date32()),
        "dec_narrow": pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in rng.integers(-10 ** 15, 10 ** 15, n)], pa.decimal128(18, 2)),
        "dec_wide": pa.array([decimal.Decimal(int(x) * 10 ** 12 + 7).scaleb(-4) for x in rng.integers(-10 ** 17, 10 ** 17, n)], pa.decimal128(32, 4)),
        "ch": pa.array([bytes([65 + int(x), 0, 0, 0]) for x in rng.integers(0, 26, n)], pa.binary(4)),
        "f64": pa.array(rng.normal(size=n), pa.float64()),
        "f32": pa.array(rng.normal(size=n).astype(np.float32), pa.float32()),
        "s": pa.array([None if i % 23 == 0 else strs[j] for i, j in enumerate(rng.integers(0, len(strs), n))], pa.string()),
        "dec_null": pa.array([None if x % 5 == 0 else decimal.Decimal(int(x)).scaleb(-2) for x in rng.integers(0, 10 ** 6, n)], pa.decimal128(12, 2)),
    })


def main():
    lib = C.CDLL(tv.REF_LIB)
    lib.ref_hash_column.restype = C.c_int32
    lib.ref_hash_column.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.ref_scan_filter.restype = C.c_int64
    lib.ref_scan_filter.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(tv.RefFilter), C.c_int32, C.c_void_p, C.c_int32]
    lib.ref_join_int64.restype = C.c_int64
    lib.ref_join_int64.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
    lib.ref_groupby_int64.restype = C.c_int64
    lib.ref_groupby_int64.argtypes = [C.c_void_p] * 3 + [C.c_int64] + [C.c_void_p] * 3 + [C.c_int64, C.c_int32]
    lib.ref_like.restype = C.c_int32
    lib.ref_like.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]
    lib.ref_extract_year.restype = C.c_int64
    lib.ref_extract_year.argtypes = [C.c_int64]
    ora = oracle_bind.load()

    # ---- hashes
    t = types_table()
    with pa.OSFile(os.path.join(HERE, "ref_types.arrow"), "wb") as f, pa.ipc.new_file(f, t.schema) as w:
        w.write_table(t)
    hashes = {",".join(map(str, ks)): tv.ref_hash(lib, [t.column(k).combine_chunks() for k in ks]) for ks in HASH_KEY_LISTS}
    np.savez_compressed(os.path.join(HERE, "ref_types_hash.npz"), **hashes)

    # ---- filters (inputs = the repo's host generator, a pure function of the seed)
    li = tpch_data.host_table(tpch_data.LINEITEM, FILTER_ORDERS)
    rowids = {str(i): tv.run_ref_filter(lib, li, fl) for i, fl in enumerate(FILTER_CASES)}
    np.savez_compressed(os.path.join(HERE, "ref_filters.npz"), **rowids)
    with open(os.path.join(HERE, "ref_filters.json"), "w") as f:
        json.dump({"orders": FILTER_ORDERS, "lineitem_rows": li.num_rows, "cases": FILTER_CASES, "counts": [int(len(rowids[str(i)])) for i in range(len(FILTER_CASES))]}, f, indent=1)

    # ---- join (duplicates on the build side, misses on the probe side); hashes = db.hash of the int64 key
    rng = np.random.default_rng(5)
    bk = rng.integers(0, 1500, 4000).astype(np.int64)
    pk = rng.integers(0, 2100, 12000).astype(np.int64)
    b, p = (oracle_bind.HostTable(pa.table({"k": pa.array(x)})).rel() for x in (bk, pk))
    bh, ph = ora.hash_keys(b, [(0, 0)]), ora.hash_keys(p, [(0, 0)])
    cap = 200000
    op, ob = np.empty(cap, np.uint32), np.empty(cap, np.uint32)
    n = lib.ref_join_int64(bk.ctypes.data, bh.ctypes.data, len(bk), pk.ctypes.data, ph.ctypes.data, len(pk), op.ctypes.data, ob.ctypes.data, cap, 4)
    assert 0 < n <= cap
    order = np.lexsort((ob[:n], op[:n]))
    np.savez_compressed(os.path.join(HERE, "ref_join.npz"), build_keys=bk, probe_keys=pk, probe_rows=op[:n][order], build_rows=ob[:n][order])

    # ---- group-by (more groups than the 1024-slot fragment cache)
    rng = np.random.default_rng(6)
    keys = rng.integers(0, 6000, 40000).astype(np.int64)
    vals = rng.integers(-1000, 1000, 40000).astype(np.int64)
    rel = oracle_bind.HostTable(pa.table({"k": pa.array(keys)})).rel()
    hs = ora.hash_keys(rel, [(0, 0)])
    ok, osum, ocnt = (np.empty(8000, np.int64) for _ in range(3))
    g = lib.ref_groupby_int64(keys.ctypes.data, hs.ctypes.data, vals.ctypes.data, len(keys), ok.ctypes.data, osum.ctypes.data, ocnt.ctypes.data, 8000, 4)
    assert 0 < g <= 8000
    order = np.argsort(ok[:g])
    np.savez_compressed(os.path.join(HERE, "ref_groupby.npz"), keys=keys, vals=vals, group_keys=ok[:g][order], sums=osum[:g][order], counts=ocnt[:g][order])

    # ---- LIKE
    rng = np.random.default_rng(7)
    alpha = ["a", "b", "c", "é", "è", "ß", "€", "%", "_", "\\"]
    palpha = ["a", "b", "c", "é", "è", "€", "%", "%", "_", "_", "\\"]
    cases = [("", ""), ("", "%"), ("abc", "abc"), ("abc", "a%"), ("abc", "%c"), ("abc", "%b%"), ("abc", "a_c"), ("abc", "a\\bc"), ("a%c", "a\\%c"),
             ("abc", "abc\\"), ("abc", "%\\"), ("a%b", "%\\%b"), ("é", "è"), ("forest green lace", "%green%"), ("PROMO BRUSHED", "PROMO%"),
             ("special packages requests", "%special%requests%"), ("Customer Complaints", "%Customer%Complaints%")]
    for _ in range(3000):
        cases.append(("".join(rng.choice(alpha, rng.integers(0, 9))), "".join(rng.choice(palpha, rng.integers(0, 7)))))
    out = [[s, pt, bool(lib.ref_like(s.encode(), len(s.encode()), pt.encode(), len(pt.encode())))] for s, pt in cases]
    with open(os.path.join(HERE, "ref_like.json"), "w") as f:
        json.dump(out, f, ensure_ascii=True)

    # ---- substr
    lib.ref_substr.restype = C.c_int64
    lib.ref_substr.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_char_p, C.c_int64]
    rng = np.random.default_rng(11)
    alpha_s = list("ab-12 xyz") + ["é", "€", "ß", "𝄞"]
    sub = []
    for _ in range(1500):
        st = "".join(rng.choice(alpha_s, rng.integers(0, 20)))
        fr, ln = int(rng.integers(-3, 24)), int(rng.integers(-2, 24))
        buf = C.create_string_buffer(256)
        b = st.encode()
        n = lib.ref_substr(b, len(b), fr, ln, buf, 256)
        sub.append([st, fr, ln, buf.raw[:n].decode()])
    with open(os.path.join(HERE, "ref_substr.json"), "w") as f:
        json.dump(sub, f, ensure_ascii=True)

    # ---- sort / top-k / key-less aggregation / generic hash map (Sorting.cpp, Heap.cpp, SimpleState.cpp, Hashtable.cpp)
    lib.ref_sort_rows.restype = C.c_int32
    lib.ref_sort_rows.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    lib.ref_topk_rows.restype = C.c_int64
    lib.ref_topk_rows.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]
    lib.ref_simple_state_sum.restype = C.c_int32
    lib.ref_simple_state_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    lib.ref_hashtable_groupby_int64.restype = C.c_int64
    lib.ref_hashtable_groupby_int64.argtypes = [C.c_void_p] * 3 + [C.c_int64] + [C.c_void_p] * 3 + [C.c_int64, C.c_int32]
    rng = np.random.default_rng(12)
    out = {}
    for name, n, k, desc in (("small", 300, 2, [0, 1]), ("tie_heavy", 5000, 3, [1, 0, 0]), ("large", 200000, 2, [1, 1])):  # > 512 rows → parallelSort
        keys = np.stack([rng.integers(0, 5 if j == 0 else 1000, n) for j in range(k)], axis=1).astype(np.int64)
        if name == "large":
            keys[:, 0] = rng.integers(-10**12, 10**12, n)
        d = np.array(desc, dtype=np.int32)
        perm = np.zeros(n, dtype=np.uint32)
        assert lib.ref_sort_rows(keys.ctypes.data, k, d.ctypes.data, n, 4, perm.ctypes.data) == 0
        out[f"{name}_keys"], out[f"{name}_desc"], out[f"{name}_perm"] = keys, d, perm
        for kt in (1, 10, 100):
            top = np.zeros(kt, dtype=np.uint32)
            got = lib.ref_topk_rows(keys.ctypes.data, k, d.ctypes.data, n, kt, 4, top.ctypes.data)
            out[f"{name}_top{kt}"] = top[:got]
    vals = rng.integers(-10**15, 10**15, 100000).astype(np.int64)
    keep = (rng.integers(0, 3, 100000) > 0).astype(np.uint8)
    lohi, cnt = (C.c_int64 * 2)(), C.c_int64()
    lib.ref_simple_state_sum(vals.ctypes.data, keep.ctypes.data, len(vals), 4, lohi, C.byref(cnt))
    out["ss_vals"], out["ss_keep"], out["ss_sum_lohi"], out["ss_count"] = vals, keep, np.array([lohi[0], lohi[1]], dtype=np.int64), np.array([cnt.value])
    nothing = np.zeros(len(vals), dtype=np.uint8)
    lib.ref_simple_state_sum(vals.ctypes.data, nothing.ctypes.data, len(vals), 4, lohi, C.byref(cnt))
    out["ss_empty_count"] = np.array([cnt.value])
    gk = rng.integers(0, 3000, 150000).astype(np.int64) * 7919 - 10**6
    gv = rng.integers(-10**9, 10**9, 150000).astype(np.int64)
    ht = tv_table_hashes(ora, gk)
    ok_, os_, oc_ = np.zeros(4000, np.int64), np.zeros(4000, np.int64), np.zeros(4000, np.int64)
    g = lib.ref_hashtable_groupby_int64(gk.ctypes.data, ht.ctypes.data, gv.ctypes.data, len(gk), ok_.ctypes.data, os_.ctypes.data, oc_.ctypes.data, 4000, 4)
    order = np.argsort(ok_[:g])
    out["ht_keys"], out["ht_vals"], out["ht_out_keys"], out["ht_out_sums"], out["ht_out_counts"] = gk, gv, ok_[:g][order], os_[:g][order], oc_[:g][order]
    np.savez_compressed(os.path.join(HERE, "ref_sort.npz"), **out)

    # ---- extract(year)
    rng = np.random.default_rng(8)
    days = list(range(-800, 800, 7)) + list(range(10950, 11330, 3)) + [-1, 0, 1, 58, 59, 60, 365, 366, 11016, 11017] + rng.integers(-100000, 100000, 1500).tolist()
    with open(os.path.join(HERE, "ref_extract_year.json"), "w") as f:
        json.dump([[int(d), int(lib.ref_extract_year(int(d) * 86_400_000_000_000))] for d in days], f)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
