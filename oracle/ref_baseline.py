"""TPC-H Q1 / Q6 / Q3 over the reference's REAL runtime objects (oracle/_ref/libldb_ref.so: Restrictions, PreAggregationHashtable(Fragment), GrowingBuffer,
HashIndexedView and the scheduler interface compiled from /root/reference in place; the JIT-generated per-tuple loops restated in
oracle/ref_build/ref_glue.cpp, compiled).  TEST INFRASTRUCTURE and the reported `cpu_baseline` of bench.py (kind "reference") — never imported by the
product.  Columns are numpy arrays of the raw Arrow value buffers: int32 for date32 / int32 / fixed_size_binary(4), 16-byte records for decimal128."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libldb_ref.so")
_lib = None


def available():
    return os.path.exists(LIB)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB)
        P, i64, f64 = C.c_void_p, C.c_int64, C.c_double
        L.ref_baseline_begin.argtypes, L.ref_baseline_begin.restype = [C.c_int32], C.c_int32
        L.ref_baseline_end.argtypes, L.ref_baseline_end.restype = [], None
        L.ref_q1.argtypes, L.ref_q1.restype = [P] * 7 + [i64, P, i64, C.POINTER(i64)], f64
        L.ref_q6.argtypes, L.ref_q6.restype = [P] * 4 + [i64, P, C.POINTER(i64)], f64
        L.ref_q3.argtypes, L.ref_q3.restype = [P, P, i64, P, P, P, P, i64, P, P, P, P, i64, P, i64, C.POINTER(i64)], f64
        _lib = L
    return _lib


def _p(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _i128(lo, hi):
    return (int(hi) << 64) | (int(lo) & 0xFFFFFFFFFFFFFFFF)


class Session:
    """scheduler + execution context for a series of runs (the reference keeps its workers alive between queries)"""

    def __init__(self, threads):
        self.threads = int(threads)

    def __enter__(self):
        if lib().ref_baseline_begin(self.threads) != 0:
            raise RuntimeError("ref_baseline_begin failed")
        return self

    def __exit__(self, *exc):
        lib().ref_baseline_end()

    def q1(self, c):
        """c: l_shipdate, l_returnflag, l_linestatus (int32), l_quantity, l_extendedprice, l_discount, l_tax (16-byte records) → (ms, rows as the legs' q1_finish takes them)"""
        n = len(c["l_shipdate"])
        out = np.zeros((64, 13), dtype=np.int64)
        g = C.c_int64()
        ms = lib().ref_q1(_p(c["l_shipdate"]), _p(c["l_returnflag"]), _p(c["l_linestatus"]), _p(c["l_quantity"]), _p(c["l_extendedprice"]), _p(c["l_discount"]), _p(c["l_tax"]), n, _p(out), 64,
                          C.byref(g))
        if ms < 0:
            raise RuntimeError("ref_q1 failed (%g)" % ms)
        partials = {}
        for r in out[: g.value]:
            sums = [_i128(r[3 + 2 * a], r[4 + 2 * a]) for a in range(5)]
            partials[(int(r[0]), int(r[1]))] = [sums[0], sums[1], sums[2], sums[3], sums[4], int(r[2])]
        return ms, partials

    def q6(self, c):
        n = len(c["l_shipdate"])
        out = np.zeros(2, dtype=np.int64)
        rows = C.c_int64()
        ms = lib().ref_q6(_p(c["l_shipdate"]), _p(c["l_discount"]), _p(c["l_quantity"]), _p(c["l_extendedprice"]), n, _p(out), C.byref(rows))
        if ms < 0:
            raise RuntimeError("ref_q6 failed (%g)" % ms)
        return ms, (_i128(out[0], out[1]) if rows.value else None)

    def q3(self, cu, od, li):
        """cu: c_custkey, c_segment4; od: o_orderkey, o_custkey, o_orderdate, o_shippriority; li: l_orderkey, l_extendedprice, l_discount, l_shipdate →
        (ms, rows (orderkey, revenue, orderdate, shippriority) sorted by the query's ORDER BY)"""
        cap = max(1024, len(od["o_orderkey"]) // 4)
        out = np.zeros((cap, 5), dtype=np.int64)
        g = C.c_int64()
        ms = lib().ref_q3(_p(cu["c_custkey"]), _p(cu["c_segment4"]), len(cu["c_custkey"]), _p(od["o_orderkey"]), _p(od["o_custkey"]), _p(od["o_orderdate"]), _p(od["o_shippriority"]),
                          len(od["o_orderkey"]), _p(li["l_orderkey"]), _p(li["l_extendedprice"]), _p(li["l_discount"]), _p(li["l_shipdate"]), len(li["l_orderkey"]), _p(out), cap, C.byref(g))
        if ms < 0:
            raise RuntimeError("ref_q3 failed (%g)" % ms)
        if g.value > cap:
            raise RuntimeError("ref_q3: %d groups exceed the output capacity %d" % (g.value, cap))
        r = out[: g.value]
        order = np.lexsort((r[:, 1], -r[:, 3]))  # revenue desc (fits 64 bits at every TPC-H scale: the hi word is the sign), orderdate asc
        return ms, [(int(r[i, 0]), _i128(r[i, 3], r[i, 4]), int(r[i, 1]), int(r[i, 2])) for i in order]


def columns_from_arrow(table, names):
    """raw value buffers of fixed-width Arrow columns as numpy arrays (int32, or 16-byte void records for decimal128)"""
    import pyarrow as pa

    out = {}
    for name in names:
        col = table.column(name).combine_chunks()
        buf = col.buffers()[1]
        if pa.types.is_decimal(col.type):
            a = np.frombuffer(buf, dtype=np.dtype("V16"))
        else:
            a = np.frombuffer(buf, dtype=np.int32)
        out[name] = np.ascontiguousarray(a[col.offset : col.offset + len(col)])
    return out


def segment4(strings):
    """first four bytes of every string of a pyarrow utf8 array as int32 (Q3's c_mktsegment: the five segment names differ in their first letter)"""
    import pyarrow as pa
    import pyarrow.compute as pc

    s = pc.utf8_slice_codeunits(strings.combine_chunks() if isinstance(strings, pa.ChunkedArray) else strings, 0, 4)
    data = np.frombuffer(s.buffers()[2], dtype=np.uint8)
    offs = np.frombuffer(s.buffers()[1], dtype=np.int32)[s.offset : s.offset + len(s) + 1]
    assert np.all(np.diff(offs) == 4), "a segment name shorter than four bytes"
    return np.ascontiguousarray(np.frombuffer(data[offs[0] : offs[-1]].tobytes(), dtype=np.int32))
