"""ctypes binding of oracle/libldb_oracle.so — TEST INFRASTRUCTURE (the checker), never the product.

Converts pyarrow tables into the oracle's plain host column structs and wraps the ora_* entry
points.  Descriptor structs are the C-ABI ones (lingodb_amd.capi) so one test input drives the
GPU path and the oracle alike.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pyarrow as pa

from lingodb_amd import capi
from lingodb_amd.capi import AggSpec, ColRef, FilterDesc, SortSpec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libldb_oracle.so")


class OraCol(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("precision", C.c_int32),
        ("scale", C.c_int32),
        ("width", C.c_int32),
        ("values", C.c_void_p),
        ("offsets", C.c_void_p),
        ("validity", C.c_void_p),
    ]


class OraTable(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_cols", C.c_int32), ("cols", C.POINTER(OraCol))]


class OraRel(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int64),
        ("n_sides", C.c_int32),
        ("tables", C.POINTER(OraTable) * capi.LDB_MAX_SIDES),
        ("rowids", C.c_void_p * capi.LDB_MAX_SIDES),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


def arrow_type_to_ldb(t):
    if pa.types.is_int8(t):
        return capi.T_INT8, 0, 0, 1
    if pa.types.is_int16(t):
        return capi.T_INT16, 0, 0, 2
    if pa.types.is_int32(t):
        return capi.T_INT32, 0, 0, 4
    if pa.types.is_int64(t):
        return capi.T_INT64, 0, 0, 8
    if pa.types.is_date32(t):
        return capi.T_DATE32, 0, 0, 4
    if pa.types.is_decimal128(t):
        return capi.T_DECIMAL128, t.precision, t.scale, 16
    if pa.types.is_fixed_size_binary(t) and t.byte_width == 4:
        return capi.T_CHAR4, 0, 0, 4
    if pa.types.is_string(t) or pa.types.is_large_string(t):
        return capi.T_UTF8, 0, 0, 0
    if pa.types.is_float64(t):
        return capi.T_FLOAT64, 0, 0, 8
    if pa.types.is_float32(t):
        return capi.T_FLOAT32, 0, 0, 4
    if pa.types.is_uint8(t):
        return capi.T_BOOL8, 0, 0, 1
    raise TypeError(f"unsupported arrow type {t}")


class HostTable:
    """pyarrow.Table → ora_table (keeps the numpy buffers alive)."""

    def __init__(self, table: pa.Table):
        self.arrow = table
        self.keep = []
        n = table.num_rows
        cols = (OraCol * max(table.num_columns, 1))()
        for i, f in enumerate(table.schema):
            arr = table.column(i).combine_chunks()
            if isinstance(arr, pa.ChunkedArray):
                arr = pa.concat_arrays(arr.chunks) if arr.num_chunks else pa.array([], type=f.type)
            ty, p, s, w = arrow_type_to_ldb(f.type)
            c = cols[i]
            c.type, c.precision, c.scale, c.width = ty, p, s, w
            if arr.null_count:
                bits = np.packbits(np.asarray(arr.is_valid().to_numpy(zero_copy_only=False), dtype=np.uint8), bitorder="little")
                self.keep.append(bits)
                c.validity = bits.ctypes.data
            if ty == capi.T_UTF8:
                la = arr.cast(pa.large_string()) if not pa.types.is_large_string(f.type) else arr
                bufs = la.buffers()
                offs = np.frombuffer(bufs[1], dtype=np.int64)[la.offset : la.offset + n + 1].copy()
                data = np.frombuffer(bufs[2], dtype=np.uint8).copy() if bufs[2] is not None and bufs[2].size else np.zeros(1, np.uint8)
                self.keep += [offs, data]
                c.offsets = offs.ctypes.data
                c.values = data.ctypes.data
            else:
                bufs = arr.buffers()
                raw = np.frombuffer(bufs[1], dtype=np.uint8)[arr.offset * w : (arr.offset + n) * w].copy() if n else np.zeros(16, np.uint8)
                self.keep.append(raw)
                c.values = raw.ctypes.data
        self.cols = cols
        self.struct = OraTable(n, table.num_columns, cols)

    def rel(self):
        return HostRel([(self, None)])


class HostRel:
    """sides: list of (HostTable, rowids uint32 ndarray | None)"""

    def __init__(self, sides, n_rows=None):
        self.sides = [(t, None if r is None else np.ascontiguousarray(r, dtype=np.uint32)) for t, r in sides]
        if n_rows is None:
            t, r = self.sides[0]
            n_rows = t.struct.n_rows if r is None else len(r)
        self.n_rows = int(n_rows)
        s = OraRel()
        s.n_rows = self.n_rows
        s.n_sides = len(self.sides)
        for i, (t, r) in enumerate(self.sides):
            s.tables[i] = C.pointer(t.struct)
            s.rowids[i] = r.ctypes.data if r is not None and len(r) else (None if r is None else np.zeros(1, np.uint32).ctypes.data)
        self.struct = s

    def select(self, idx):
        """restrict to logical rows idx"""
        idx = np.asarray(idx, dtype=np.int64)
        sides = []
        for t, r in self.sides:
            sides.append((t, idx.astype(np.uint32) if r is None else r[idx]))
        return HostRel(sides, len(idx))

    def phys(self, side):
        t, r = self.sides[side]
        return np.arange(self.n_rows, dtype=np.uint32) if r is None else r


def _refs(cols):
    arr = (ColRef * max(len(cols), 1))()
    for i, c in enumerate(cols):
        arr[i] = ColRef(*c)
    return arr


def _preds(plist):
    arr = (FilterDesc * max(len(plist), 1))()
    for i, (d, k) in enumerate(plist):
        arr[i] = d
    return arr


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        P = C.c_void_p
        lib.ora_hash64.restype = C.c_uint64
        lib.ora_hash64.argtypes = [C.c_int64]
        lib.ora_hash_combine.restype = C.c_uint64
        lib.ora_hash_combine.argtypes = [C.c_uint64, C.c_uint64]
        lib.ora_xxh64.restype = C.c_uint64
        lib.ora_xxh64.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64]
        lib.ora_hash_varlen.restype = C.c_uint64
        lib.ora_hash_varlen.argtypes = [C.c_char_p, C.c_uint32]
        lib.ora_hash_i128.restype = C.c_uint64
        lib.ora_hash_i128.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_uint64]
        lib.ora_varlen32_image.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p]
        lib.ora_bloom_mask.restype = C.c_uint16
        lib.ora_bloom_mask.argtypes = [C.c_uint32]
        lib.ora_scan_filter.restype = C.c_int64
        lib.ora_scan_filter.argtypes = [C.POINTER(OraRel), C.POINTER(FilterDesc), C.c_int32, P, C.c_int32]
        lib.ora_hash_keys.argtypes = [C.POINTER(OraRel), C.POINTER(ColRef), C.c_int32, P]
        lib.ora_groupby.restype = C.c_int64
        lib.ora_groupby.argtypes = [C.POINTER(OraRel), C.POINTER(FilterDesc), C.c_int32, C.POINTER(ColRef), C.c_int32, C.POINTER(AggSpec), C.c_int32, C.c_int32, P, P, P, C.c_int64]
        lib.ora_join.restype = C.c_int64
        lib.ora_join.argtypes = [C.POINTER(OraRel), C.POINTER(ColRef), C.POINTER(OraRel), C.POINTER(ColRef), C.c_int32, C.c_int32, C.c_int32, P, P, P, C.c_int64]
        lib.ora_sort.argtypes = [C.POINTER(OraRel), C.POINTER(SortSpec), C.c_int32, P]
        lib.ora_topk.restype = C.c_int64
        lib.ora_topk.argtypes = [C.POINTER(OraRel), C.POINTER(SortSpec), C.c_int32, C.c_int64, P]
        lib.ora_eval_expr.argtypes = [C.POINTER(OraRel), C.POINTER(capi.Expr), P]
        lib.ora_partition_ids.argtypes = [C.POINTER(OraRel), C.POINTER(ColRef), C.c_int32, C.c_int32, P]
        lib.ora_num_cores.restype = C.c_int32
        lib.ora_like.restype = C.c_int32
        lib.ora_like.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]
        lib.ora_substr.restype = None
        lib.ora_substr.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.ora_extract_year.restype = C.c_int64
        lib.ora_extract_year.argtypes = [C.c_int64]
        p64 = C.POINTER(C.c_int64)
        lib.ora_decimal_muldiv.restype = C.c_int32
        lib.ora_decimal_muldiv.argtypes = [p64, p64, C.c_int32, C.c_int32, p64, p64]
        lib.ora_segment_tree.restype = C.c_int32
        lib.ora_segment_tree.argtypes = [P, P, C.c_int64, C.c_int32, P, P, C.c_int64, P, P]
        lib.ora_window.restype = C.c_int32
        lib.ora_window.argtypes = [P, P, P, P, C.c_int64, C.c_int32, C.c_int64, C.c_int64, P, P]
        lib.ora_setop_multiplicity.restype = C.c_int64
        lib.ora_setop_multiplicity.argtypes = [C.c_int32, C.c_int64, C.c_int64]

    # scalar
    def hash64(self, v):
        return self.lib.ora_hash64(v)

    def hash_combine(self, h, total):
        return self.lib.ora_hash_combine(h, total)

    def xxh64(self, b, seed=0):
        return self.lib.ora_xxh64(b, len(b), seed)

    def hash_varlen(self, b):
        return self.lib.ora_hash_varlen(b, len(b))

    # operators
    def scan_filter(self, rel: HostRel, plist, threads=1):
        out = np.empty(max(rel.n_rows, 1), dtype=np.uint32)
        n = self.lib.ora_scan_filter(C.byref(rel.struct), _preds(plist), len(plist), out.ctypes.data, threads)
        return out[:n].copy()

    def hash_keys(self, rel, keys):
        out = np.empty(max(rel.n_rows, 1), dtype=np.uint64)
        self.lib.ora_hash_keys(C.byref(rel.struct), _refs(keys), len(keys), out.ctypes.data)
        return out[: rel.n_rows]

    def groupby(self, rel, keys, aggs, plist=(), threads=1, cap=None):
        """→ (rep_rows uint32[g], vals object array [g][a] of python ints / floats, valid uint8[g][a])"""
        cap = cap or max(rel.n_rows, 1)
        na = len(aggs)
        rep = np.empty(cap, dtype=np.uint32)
        vals = np.empty((cap, max(na, 1), 2), dtype=np.int64)
        valid = np.empty((cap, max(na, 1)), dtype=np.uint8)
        aarr = (AggSpec * max(na, 1))()
        for i, a in enumerate(aggs):
            aarr[i] = a
        plist = list(plist)
        g = self.lib.ora_groupby(C.byref(rel.struct), _preds(plist), len(plist), _refs(keys), len(keys), aarr, na, threads, rep.ctypes.data, vals.ctypes.data, valid.ctypes.data, cap)
        assert g <= cap
        out = []
        for r in range(g):
            row = []
            for a in range(na):
                lo, hi = int(vals[r, a, 0]), int(vals[r, a, 1])
                if aggs[a].arg.is_float and aggs[a].fn not in (capi.AGG_COUNT, capi.AGG_COUNT_STAR):
                    row.append(float(np.int64(lo).view(np.float64)))
                else:
                    row.append((hi << 64) | (lo & 0xFFFFFFFFFFFFFFFF))
            out.append(row)
        return rep[:g].copy(), out, valid[:g, :na].copy()

    def join(self, build, bkeys, probe, pkeys, kind=capi.JOIN_INNER, threads=1):
        n = self.lib.ora_join(C.byref(build.struct), _refs(bkeys), C.byref(probe.struct), _refs(pkeys), len(bkeys), kind, threads, None, None, None, 0)
        cap = max(n, probe.n_rows, 1)
        op = np.zeros(cap, dtype=np.uint32)
        ob = np.zeros(cap, dtype=np.uint32)
        mk = np.zeros(cap, dtype=np.uint8)
        n2 = self.lib.ora_join(C.byref(build.struct), _refs(bkeys), C.byref(probe.struct), _refs(pkeys), len(bkeys), kind, threads, op.ctypes.data, ob.ctypes.data, mk.ctypes.data, cap)
        assert n2 == n
        return op[:n].copy(), ob[:n].copy(), mk[:n].copy()

    def sort(self, rel, specs):
        out = np.empty(max(rel.n_rows, 1), dtype=np.uint32)
        arr = (SortSpec * len(specs))(*specs)
        self.lib.ora_sort(C.byref(rel.struct), arr, len(specs), out.ctypes.data)
        return out[: rel.n_rows]

    def topk(self, rel, specs, k):
        out = np.empty(max(min(rel.n_rows, k), 1), dtype=np.uint32)
        arr = (SortSpec * len(specs))(*specs)
        n = self.lib.ora_topk(C.byref(rel.struct), arr, len(specs), k, out.ctypes.data)
        return out[:n]

    def eval_expr(self, rel, e):
        out = np.empty((max(rel.n_rows, 1), 2), dtype=np.int64)
        self.lib.ora_eval_expr(C.byref(rel.struct), C.byref(e), out.ctypes.data)
        return [((int(h) << 64) | (int(l) & 0xFFFFFFFFFFFFFFFF)) for l, h in out[: rel.n_rows]]

    def partition_ids(self, rel, keys, nparts):
        out = np.empty(max(rel.n_rows, 1), dtype=np.int32)
        self.lib.ora_partition_ids(C.byref(rel.struct), _refs(keys), len(keys), nparts, out.ctypes.data)
        return out[: rel.n_rows]

    def num_cores(self):
        return self.lib.ora_num_cores()

    def like(self, s, pattern):
        s = s.encode() if isinstance(s, str) else s
        pattern = pattern.encode() if isinstance(pattern, str) else pattern
        return bool(self.lib.ora_like(s, len(s), pattern, len(pattern)))

    def substr(self, s, start, length):
        b = s.encode() if isinstance(s, str) else s
        b0, b1 = C.c_int64(), C.c_int64()
        self.lib.ora_substr(b, len(b), start, length, C.byref(b0), C.byref(b1))
        return b[b0.value : b1.value]

    def extract_year(self, days):
        return int(self.lib.ora_extract_year(int(days)))

    def decimal_muldiv(self, num, mul, mul_div_pow10, pow10, den):
        """((num * mul) sdiv 10^mul_div_pow10) * 10^pow10 sdiv den on 128-bit wrapping integers; None if den == 0"""

        def words(v):
            v &= (1 << 128) - 1
            return (C.c_int64 * 2)(C.c_int64(v & ((1 << 64) - 1)).value, C.c_int64(v >> 64).value)

        out = (C.c_int64 * 2)()
        ok = self.lib.ora_decimal_muldiv(words(num), words(mul), mul_div_pow10, pow10, words(den), out)
        if not ok:
            return None
        v = ((out[1] & ((1 << 64) - 1)) << 64) | (out[0] & ((1 << 64) - 1))
        return v - (1 << 128) if v >> 127 else v


def _lohi(vals):
    """python ints → int64[n, 2] (lo, hi) two's complement pairs"""
    out = np.empty((len(vals), 2), dtype=np.int64)
    for i, v in enumerate(vals):
        v &= (1 << 128) - 1
        out[i, 0] = C.c_int64(v & ((1 << 64) - 1)).value
        out[i, 1] = C.c_int64(v >> 64).value
    return out


def _from_lohi(a):
    out = []
    for lo, hi in a.tolist():
        v = ((hi & ((1 << 64) - 1)) << 64) | (lo & ((1 << 64) - 1))
        out.append(v - (1 << 128) if v >> 127 else v)
    return out


def segment_tree(oracle, vals, valid, fn, frm, to):
    """ora_segment_tree: vals python ints, valid 0/1, queries (from[q], to[q]) inclusive → [(value | None)]"""
    n, nq = len(vals), len(frm)
    v, ok = _lohi(vals), np.asarray(valid, dtype=np.uint8)
    f, t = np.asarray(frm, dtype=np.int64), np.asarray(to, dtype=np.int64)
    ov, ook = np.zeros((nq, 2), np.int64), np.zeros(nq, np.uint8)
    st = oracle.lib.ora_segment_tree(v.ctypes.data, ok.ctypes.data, n, fn, f.ctypes.data, t.ctypes.data, nq, ov.ctypes.data, ook.ctypes.data)
    assert st == 0, st
    return [x if k else None for x, k in zip(_from_lohi(ov), ook.tolist())]


def window(oracle, vals, valid, part_start, part_end, fn, frame_from, frame_to):
    """ora_window over rows already in window order → [(value | None)] per row"""
    n = len(part_start)
    v = _lohi(vals) if vals is not None else np.zeros((n, 2), np.int64)
    ok = np.asarray(valid, dtype=np.uint8) if valid is not None else np.ones(n, np.uint8)
    ps, pe = np.asarray(part_start, dtype=np.int64), np.asarray(part_end, dtype=np.int64)
    ov, ook = np.zeros((n, 2), np.int64), np.zeros(n, np.uint8)
    st = oracle.lib.ora_window(v.ctypes.data, ok.ctypes.data, ps.ctypes.data, pe.ctypes.data, n, fn, frame_from, frame_to, ov.ctypes.data, ook.ctypes.data)
    assert st == 0, st
    return [x if k else None for x, k in zip(_from_lohi(ov), ook.tolist())]


def load():
    build()  # make: a no-op when the library is newer than its sources, so a stale checker is never loaded
    return Oracle(C.CDLL(LIB))
