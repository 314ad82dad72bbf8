"""Thin object layer over the C-ABI (tests / bench harness).  Every call goes through
liblingodb_gpu.so; nothing here computes query results on the CPU."""
import ctypes as C

import numpy as np
import pyarrow as pa

from . import capi
from .capi import (AggSpec, ColRef, ColType, Expr, Factor, FilterDesc, SortSpec, Term, check, check_plan)


# ------------------------------------------------------------------ descriptor builders
def colref(side, col):
    return ColRef(int(side), int(col))


def _split128(v):
    v = int(v)
    lo = v & 0xFFFFFFFFFFFFFFFF
    hi = (v >> 64) & 0xFFFFFFFFFFFFFFFF
    if hi >= 1 << 63:
        hi -= 1 << 64
    return lo, hi


class _Keep:
    """holds python objects referenced by raw pointers inside descriptor structs"""

    def __init__(self):
        self.objs = []

    def add(self, o):
        self.objs.append(o)
        return o


def pred(col, op, value=None, *, rhs_col=None, values=None, keep=None):
    """One conjunct.  col=(side, col); value int | bytes/str | float; rhs_col=(side,col); values=list for IN."""
    keep = keep if keep is not None else _Keep()
    d = FilterDesc()
    d.col = colref(*col)
    d.op = op
    if op == capi.F_NOTNULL:
        return d, keep
    if rhs_col is not None:
        d.rhs_kind = capi.RHS_COLUMN
        d.rhs_col = colref(*rhs_col)
        return d, keep
    if op == capi.F_IN:
        vals = list(values)
        d.n_in = len(vals)
        if vals and isinstance(vals[0], (bytes, str)):
            bs = [v.encode() if isinstance(v, str) else v for v in vals]
            arr = (C.c_char_p * len(bs))(*bs)
            lens = (C.c_int32 * len(bs))(*[len(b) for b in bs])
            keep.add(bs), keep.add(arr), keep.add(lens)
            d.rhs_kind = capi.RHS_STRING
            d.in_strs = arr
            d.in_str_lens = lens
        else:
            flat = []
            for v in vals:
                if isinstance(v, float):
                    flat += [int(np.float64(v).view(np.int64)), 0]
                else:
                    lo, hi = _split128(v)
                    flat += [lo if lo < 1 << 63 else lo - (1 << 64), hi]
            arr = (C.c_int64 * len(flat))(*flat)
            keep.add(arr)
            d.rhs_kind = capi.RHS_FLOAT if vals and isinstance(vals[0], float) else capi.RHS_INT
            d.in_values = arr
        return d, keep
    if isinstance(value, (bytes, str)):
        b = value.encode() if isinstance(value, str) else value
        keep.add(b)
        d.rhs_kind = capi.RHS_STRING
        d.str = b
        d.str_len = len(b)
    elif isinstance(value, float):
        d.rhs_kind = capi.RHS_FLOAT
        d.value_f64 = value
    else:
        d.rhs_kind = capi.RHS_INT
        d.value_lo, d.value_hi = _split128(value)
    return d, keep


def preds_array(plist):
    """plist: list of (FilterDesc, keep) → (ctypes array, n, keepalive)"""
    n = len(plist)
    arr = (FilterDesc * max(n, 1))()
    keeps = []
    for i, (d, k) in enumerate(plist):
        arr[i] = d
        keeps.append(k)
    return arr, n, keeps


def factor(a=0, b=0, col=None):
    f = Factor()
    f.has_col = 0 if col is None else 1
    if col is not None:
        f.col = colref(*col)
    f.a = int(a)
    f.b = int(b)
    return f


def expr(terms, is_float=False):
    """terms: list of dicts {factors:[Factor], negate:bool, div_pow10:int}"""
    e = Expr()
    e.n_terms = len(terms)
    e.is_float = 1 if is_float else 0
    for t, tm in enumerate(terms):
        e.t[t].n_factors = len(tm["factors"])
        e.t[t].negate = 1 if tm.get("negate") else 0
        e.t[t].div_pow10 = int(tm.get("div_pow10", 0))
        for f, fa in enumerate(tm["factors"]):
            e.t[t].f[f] = fa
    return e


def col_expr(col, is_float=False):
    return expr([{"factors": [factor(0, 1, col)]}], is_float)


def agg(fn, e=None, *, wide=False, out_type=capi.T_INT64, p=0, s=0, preds=(), avg_pow10=0):
    a = AggSpec()
    a.fn = fn
    a.wide = 1 if wide else 0
    if e is not None:
        a.arg = e
    a.n_preds = len(preds)
    keeps = []
    for i, (d, k) in enumerate(preds):
        a.preds[i] = d
        keeps.append(k)
    a.avg_pow10 = avg_pow10
    a.out_type = out_type
    a.out_precision = p
    a.out_scale = s
    a._keep = keeps
    return a


def sort_spec(col, descending=False):
    s = SortSpec()
    s.col = colref(*col)
    s.descending = 1 if descending else 0
    return s


def _refs(cols):
    n = len(cols)
    arr = (ColRef * max(n, 1))()
    for i, c in enumerate(cols):
        arr[i] = colref(*c)
    return arr, n


# ------------------------------------------------------------------ handles
class Table:
    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    def release(self):
        if self.h and self.ctx.h:
            check(self.ctx.lib.ldb_gpu_table_release(self.ctx.h, self.h))
        self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    @property
    def rows(self):
        return self.ctx.lib.ldb_gpu_table_rows(self.h)

    @property
    def n_cols(self):
        return self.ctx.lib.ldb_gpu_table_cols(self.h)

    def col(self, name):
        i = self.ctx.lib.ldb_gpu_table_col_index(self.h, name.encode())
        if i < 0:
            raise KeyError(name)
        return i

    def col_name(self, i):
        return self.ctx.lib.ldb_gpu_table_col_name(self.h, i).decode()

    def coltype(self, i):
        t = ColType()
        check(self.ctx.lib.ldb_gpu_table_coltype(self.h, i, C.byref(t)))
        return t

    def dict_size(self, i):
        """entries of the column's utf8 dictionary, -1 = not dictionary-encoded"""
        return self.ctx.lib.ldb_gpu_table_dict_size(self.h, i)

    def dict_encode(self, i):
        n = C.c_int32()
        check(self.ctx.lib.ldb_gpu_table_dict_encode(self.ctx.h, self.h, i, C.byref(n)))
        return n.value

    def col_width(self, i):
        return self.ctx.lib.ldb_gpu_table_col_width(self.h, i)

    def col_ptrs(self, i):
        v, o, b = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nb = C.c_int64()
        check(self.ctx.lib.ldb_gpu_table_col_ptrs(self.h, i, C.byref(v), C.byref(o), C.byref(b), C.byref(nb)))
        return v.value, o.value, b.value, nb.value

    def read_fixed(self, i):
        """raw bytes of a fixed-width column as a numpy uint8 array"""
        w = self.col_width(i)
        out = np.empty(self.rows * w, dtype=np.uint8)
        check(self.ctx.lib.ldb_gpu_table_read_fixed(self.ctx.h, self.h, i, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def rel(self):
        r = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_rel_from_table(self.ctx.h, self.h, C.byref(r)))
        return Rel(self.ctx, r, [self])

    def to_arrow(self):
        """D2H export through the Arrow C Data Interface (ldb_gpu_export)."""
        schema, array = capi.ArrowSchema(), capi.ArrowArray()
        check(self.ctx.lib.ldb_gpu_export(self.ctx.h, self.h, C.byref(schema), C.byref(array)))
        batch = pa.RecordBatch._import_from_c(C.addressof(array), C.addressof(schema))
        return pa.Table.from_batches([batch])


class Rel:
    def __init__(self, ctx, handle, deps=()):
        self.ctx, self.h = ctx, handle
        self.deps = list(deps)  # keep tables / build relations alive

    def release(self):
        if self.h and self.ctx.h:
            check(self.ctx.lib.ldb_gpu_rel_release(self.ctx.h, self.h))
        self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    @property
    def rows(self):
        return self.ctx.lib.ldb_gpu_rel_rows(self.ctx.h, self.h)

    @property
    def sides(self):
        return self.ctx.lib.ldb_gpu_rel_sides(self.h)

    def rowids(self, side=0):
        out = np.empty(max(self.rows, 1), dtype=np.uint32)
        check(self.ctx.lib.ldb_gpu_rel_read_rowids(self.ctx.h, self.h, side, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size))
        return out[: self.rows]

    def join_nl(self, build_rel, residual=(), kind=capi.JOIN_INNER):
        """nested-loop join of this (probe) relation with `build_rel`: residual = [(probe_col, op, build_col)] is the whole
        join predicate (none = cross product); every build row is visited per probe row — small build sides only"""
        ra = (capi.JoinResidual * max(1, len(residual)))()
        for i, (pc, op, bc) in enumerate(residual):
            ra[i].probe_col, ra[i].op, ra[i].build_col = colref(*pc), op, colref(*bc)
        r, m = C.c_void_p(), C.c_void_p()
        check(self.ctx.lib.ldb_gpu_join_nl(self.ctx.h, self.h, build_rel.h, kind, ra, len(residual), C.byref(r), C.byref(m)))
        out = Rel(self.ctx, r, self.deps + build_rel.deps + [build_rel])
        if kind == capi.JOIN_MARK:
            return out, Table(self.ctx, m)
        return out

    # ---- operators
    def scan_filter(self, plist):
        arr, n, keep = preds_array(plist)
        r = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_scan_filter(self.ctx.h, self.h, arr, n, C.byref(r)))
        return Rel(self.ctx, r, self.deps)

    def scan_filter_dnf(self, clauses):
        """clauses: list of predicate lists; rows satisfying ANY clause (each a conjunction)"""
        flat = [p for cl in clauses for p in cl]
        arr, n, keep = preds_array(flat)
        sizes = (C.c_int32 * len(clauses))(*[len(cl) for cl in clauses])
        r = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_scan_filter_dnf(self.ctx.h, self.h, arr, sizes, len(clauses), C.byref(r)))
        return Rel(self.ctx, r, self.deps)

    def map_expr(self, prog, out_type=capi.T_INT64, p=0, s=0, name="expr"):
        """prog: postfix list of ("col", (side, col)) | ("const", int) | ("add",) … | ("cmp", F_op) | ("mul10", k) | ("div10", k)"""
        ops = {"add": capi.X_ADD, "sub": capi.X_SUB, "mul": capi.X_MUL, "sdiv": capi.X_SDIV, "neg": capi.X_NEG, "and": capi.X_AND, "or": capi.X_OR, "not": capi.X_NOT,
               "select": capi.X_SELECT, "isnull": capi.X_ISNULL, "coalesce": capi.X_COALESCE}
        arr = (capi.XInstr * len(prog))()
        for i, ins in enumerate(prog):
            if ins[0] == "col":
                arr[i].op, arr[i].col = capi.X_COL, colref(*ins[1])
            elif ins[0] == "const":
                lo, hi = _split128(ins[1])
                arr[i].op, arr[i].lo, arr[i].hi = capi.X_CONST, lo if lo < 1 << 63 else lo - (1 << 64), hi
            elif ins[0] == "cmp":
                arr[i].op, arr[i].arg = capi.X_CMP, ins[1]
            elif ins[0] in ("mul10", "div10"):
                arr[i].op, arr[i].arg = (capi.X_MUL_POW10 if ins[0] == "mul10" else capi.X_SDIV_POW10), ins[1]
            else:
                arr[i].op = ops[ins[0]]
        t = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_map_expr(self.ctx.h, self.h, arr, len(prog), ColType(out_type, p, s, 1), name.encode(), C.byref(t)))
        return Table(self.ctx, t)

    def map_substr(self, col, start, length, name="substr"):
        t = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_map_substr(self.ctx.h, self.h, colref(*col), start, length, name.encode(), C.byref(t)))
        return Table(self.ctx, t)

    def scan_count(self, plist):
        arr, n, keep = preds_array(plist)
        c = C.c_int64()
        check(self.ctx.lib.ldb_gpu_scan_count(self.ctx.h, self.h, arr, n, C.byref(c)))
        return c.value

    def hash_keys(self, keys):
        arr, n = _refs(keys)
        t = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_hash_keys(self.ctx.h, self.h, arr, n, C.byref(t)))
        tab = Table(self.ctx, t)
        return tab.read_fixed(0).view(np.uint64)

    def map_column(self, col, fn=capi.FN_EXTRACT_YEAR, name="year"):
        """scalar function of one column as a new 1-column table (one value per row of this relation)"""
        t = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_map_column(self.ctx.h, self.h, colref(*col), fn, name.encode(), C.byref(t)))
        return Table(self.ctx, t)

    def map_muldiv(self, num, den, mul=1, mul_div_pow10=0, pow10=0, precision=38, scale=6, name="ratio"):
        """decimal `mul * num / den` per row as a new 1-column decimal128(precision, scale) table:
        ((num * mul) sdiv 10^mul_div_pow10) * 10^pow10 sdiv den in wrapping 128-bit arithmetic"""
        t = C.c_void_p()
        m = int(mul) & ((1 << 128) - 1)
        lo, hi = m & ((1 << 64) - 1), m >> 64
        lo = lo - (1 << 64) if lo >> 63 else lo
        hi = hi - (1 << 64) if hi >> 63 else hi
        check(self.ctx.lib.ldb_gpu_map_muldiv(self.ctx.h, self.h, colref(*num), lo, hi, mul_div_pow10, pow10, colref(*den), precision, scale, name.encode(), C.byref(t)))
        return Table(self.ctx, t)

    def zip(self, table):
        """this relation plus `table` (same row count) as a new last side"""
        r = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_rel_zip(self.ctx.h, self.h, table.h, C.byref(r)))
        return Rel(self.ctx, r, self.deps + [table])

    def groupby(self, keys, aggs, plist=(), est_groups=0):
        parr, np_, keep = preds_array(list(plist))
        karr, nk = _refs(keys)
        aarr = (AggSpec * max(len(aggs), 1))()
        for i, a in enumerate(aggs):
            aarr[i] = a
        t = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_groupby(self.ctx.h, self.h, parr, np_, karr, nk, aarr, len(aggs), est_groups, C.byref(t)))
        return Table(self.ctx, t)

    def materialize(self, cols):
        arr, n = _refs(cols)
        t = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_materialize(self.ctx.h, self.h, arr, n, C.byref(t)))
        return Table(self.ctx, t)

    def join_build(self, keys, unique=False):
        arr, n = _refs(keys)
        h = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_join_build(self.ctx.h, self.h, arr, n, 1 if unique else 0, C.byref(h)))
        return HashTable(self.ctx, h, self)

    def sort(self, specs):
        arr = (SortSpec * len(specs))(*specs)
        r = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_sort(self.ctx.h, self.h, arr, len(specs), C.byref(r)))
        return Rel(self.ctx, r, self.deps)

    def set_op(self, other, op, cols=None, other_cols=None):
        """UNION [ALL] / INTERSECT [ALL] / EXCEPT [ALL] of the listed columns (default: all columns of side 0) → Table"""
        n = self.deps[0].n_cols if cols is None else len(cols)
        a, na = _refs(cols if cols is not None else [(0, c) for c in range(n)])
        b, nb = _refs(other_cols if other_cols is not None else (cols if cols is not None else [(0, c) for c in range(n)]))
        assert na == nb
        t = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_set_op(self.ctx.h, self.h, a, other.h, b, na, op, C.byref(t)))
        return Table(self.ctx, t)

    def window(self, part_keys, order, fns, frame=(capi.FRAME_UNBOUNDED_PRECEDING, 0)):
        """fns: [(capi.WIN_*, (side, col) | None)] → (rows in window order: Rel, one column per function: Table)"""
        pk, npk = _refs(part_keys)
        oarr = (SortSpec * max(len(order), 1))(*order)
        farr = (capi.WindowFn * len(fns))()
        for i, (fn, col) in enumerate(fns):
            farr[i].fn = fn
            farr[i].col = colref(*(col if col is not None else (0, 0)))
        r, t = C.c_void_p(), C.c_void_p()
        check(self.ctx.lib.ldb_gpu_window(self.ctx.h, self.h, pk, npk, oarr, len(order), frame[0], frame[1], farr, len(fns), C.byref(r), C.byref(t)))
        return Rel(self.ctx, r, self.deps), Table(self.ctx, t)

    def topk(self, specs, k):
        arr = (SortSpec * len(specs))(*specs)
        r = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_topk(self.ctx.h, self.h, arr, len(specs), k, C.byref(r)))
        return Rel(self.ctx, r, self.deps)

    def partition(self, keys, nparts, cols):
        karr, nk = _refs(keys)
        carr, nc = _refs(cols)
        counts = (C.c_int64 * nparts)()
        t = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_partition(self.ctx.h, self.h, karr, nk, nparts, carr, nc, C.byref(t), counts))
        return Table(self.ctx, t), list(counts)


class HashTable:
    def __init__(self, ctx, handle, build_rel):
        self.ctx, self.h, self.build = ctx, handle, build_rel

    def release(self):
        if self.h and self.ctx.h:
            check(self.ctx.lib.ldb_gpu_hashtable_release(self.ctx.h, self.h))
        self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    @property
    def slots(self):
        return self.ctx.lib.ldb_gpu_hashtable_slots(self.h)

    @property
    def table_bytes(self):
        return self.ctx.lib.ldb_gpu_hashtable_bytes(self.h)

    def probe(self, probe_rel, keys, kind=capi.JOIN_INNER, residual=()):
        """residual: [(probe_col, op, build_col)] — extra `probe_col OP build_col` conjuncts of the join predicate"""
        arr, n = _refs(keys)
        r, m = C.c_void_p(), C.c_void_p()
        if residual:
            ra = (capi.JoinResidual * len(residual))()
            for i, (pc, op, bc) in enumerate(residual):
                ra[i].probe_col, ra[i].op, ra[i].build_col = colref(*pc), op, colref(*bc)
            check(self.ctx.lib.ldb_gpu_join_probe_residual(self.ctx.h, self.h, probe_rel.h, arr, n, kind, ra, len(residual), C.byref(r), C.byref(m)))
        else:
            check(self.ctx.lib.ldb_gpu_join_probe(self.ctx.h, self.h, probe_rel.h, arr, n, kind, C.byref(r), C.byref(m)))
        out = Rel(self.ctx, r, probe_rel.deps + self.build.deps + [self.build])
        if kind == capi.JOIN_MARK:
            return out, Table(self.ctx, m)
        return out

    def probe_semi_anti_build(self, probe_rel, keys, anti_preds, residual=()):
        """build rows with a partner among the probe rows and none among the probe rows that also pass `anti_preds`
        (ldb_gpu_join_probe_semi_anti_build: SEMI_BUILD + ANTI_BUILD over the same table in one pass)"""
        arr, n = _refs(keys)
        ra = (capi.JoinResidual * max(len(residual), 1))()
        for i, (pc, op, bc) in enumerate(residual):
            ra[i].probe_col, ra[i].op, ra[i].build_col = colref(*pc), op, colref(*bc)
        parr, np_, keep = preds_array(anti_preds)
        r = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_join_probe_semi_anti_build(self.ctx.h, self.h, probe_rel.h, arr, n, ra, len(residual), parr, np_, C.byref(r)))
        del keep
        return Rel(self.ctx, r, probe_rel.deps + self.build.deps + [self.build])

    def probe_count(self, probe_rel, keys):
        arr, n = _refs(keys)
        c = C.c_int64()
        check(self.ctx.lib.ldb_gpu_join_probe_count(self.ctx.h, self.h, probe_rel.h, arr, n, C.byref(c)))
        return c.value


class Comm:
    """Communicator of the library (one rank per ldb_ctx): the 128-byte id made on rank 0 reaches the others
    through `exchange_id(bytes | None) -> bytes` (a torch.distributed broadcast, a file, …).
    transport "rccl" (one rank per GPU, xGMI) or "shm" (host-staged; ranks of one node, may share a GPU)."""

    def __init__(self, ctx, rank, world, exchange_id, transport="rccl"):
        # a Python process that will import torch must do so BEFORE librccl is bound: torch ships its own
        # librccl / HIP runtime copies and a second copy loaded afterwards aborts at interpreter exit
        if transport == "rccl":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        self.ctx, self.rank, self.world = ctx, rank, world
        buf = C.create_string_buffer(128)
        err = None
        if rank == 0:
            ctx.lib.ldb_gpu_set_option(b"comm_transport", 1 if transport == "shm" else 0)
            try:
                check(ctx.lib.ldb_gpu_comm_unique_id(buf))
            except Exception as e:  # the peers are waiting for the id: hand them a sentinel first, then fail everywhere
                err = e
        ident = exchange_id((b"" if err else buf.raw) if rank == 0 else None)
        if err:
            raise err
        if not ident:
            raise capi.LdbError(-1, "rank 0 could not create a communicator id")
        h = C.c_void_p()
        check(ctx.lib.ldb_gpu_comm_create(ctx.h, rank, world, C.create_string_buffer(ident, 128), C.byref(h)))
        self.h = h
        self.transport = ctx.lib.ldb_gpu_comm_transport(h).decode()

    def close(self):
        if self.h:
            self.ctx.lib.ldb_gpu_comm_destroy(self.h)
            self.h = None

    def stats(self, reset=False):
        """traffic and time of this rank's exchanges since the last reset (ldb_gpu_comm_stats)"""
        s = capi.CommStats()
        check(self.ctx.lib.ldb_gpu_comm_stats(self.h, C.byref(s), 1 if reset else 0))
        return {f: getattr(s, f) for f, _ in capi.CommStats._fields_}

    def allgather(self, table, name="gathered"):
        t = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_allgather(self.ctx.h, self.h, table.h, name.encode(), C.byref(t)))
        return Table(self.ctx, t)

    def alltoall(self, table, send_counts, name="exchanged"):
        cnt = (C.c_int64 * self.world)(*[int(c) for c in send_counts])
        t = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_alltoall(self.ctx.h, self.h, table.h, cnt, name.encode(), C.byref(t)))
        return Table(self.ctx, t)

    def shuffle(self, rel, keys, cols, name="shuffled"):
        karr, nk = _refs(keys)
        carr, ncol = _refs(cols)
        t = C.c_void_p()
        check(self.ctx.lib.ldb_gpu_shuffle(self.ctx.h, self.h, rel.h, karr, nk, carr, ncol, name.encode(), C.byref(t)))
        return Table(self.ctx, t)


def describe_ipc(path):
    """schema and batch sizes of an Arrow IPC file as the library's own reader sees them (no device needed)"""
    import json

    buf = C.create_string_buffer(1 << 20)
    check(capi.gpu_lib().ldb_gpu_ipc_describe(str(path).encode(), buf, len(buf)))
    return json.loads(buf.value.decode(errors="replace"))


def translate_subop_dump(dump, name="subop_dump"):
    """(plan text, per-step placement report) for a dump of the reference's `mlir-subop-to-json`; raises LdbError
    with the offending execution step when a step has no device pattern (needs no GPU)"""
    import json
    import os

    text = dump
    if not dump.lstrip().startswith("["):
        with open(dump) as f:
            text = f.read()
    lib = capi.host_lib()
    need = C.c_int64()
    buf = C.create_string_buffer(1 << 16)
    st = lib.ldb_subop_translate(text.encode(), name.encode(), buf, len(buf), C.byref(need))
    if st == capi.LDB_ERR_INVALID and need.value > len(buf):
        buf = C.create_string_buffer(need.value)
        st = lib.ldb_subop_translate(text.encode(), name.encode(), buf, len(buf), C.byref(need))
    report = json.loads(lib.ldb_subop_report().decode())
    if st != capi.LDB_OK:
        err = capi.LdbError(st, lib.ldb_subop_last_error().decode(errors="replace"))
        err.report = report
        raise err
    return buf.value.decode(), report


class PreparedPlan:
    """ldb_plan_prepare / ldb_plan_execute (libldb_host.so)"""

    def __init__(self, ctx, text):
        self.ctx = ctx
        self.h = C.c_void_p()
        st = capi.host_lib().ldb_plan_prepare(ctx.h, text.encode(), C.byref(self.h))
        if st != capi.LDB_OK:
            raise capi.LdbError(st, capi.host_lib().ldb_plan_json_last_error().decode(errors="replace"))
        ctx._plans.add(self)

    def execute(self, tables, comm=None):
        names = list(tables)
        narr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        tarr = (C.c_void_p * len(names))(*[tables[n].h for n in names])
        t = C.c_void_p()
        st = capi.host_lib().ldb_plan_execute(self.h, comm.h if comm is not None else None, narr, tarr, len(names), C.byref(t))
        if st != capi.LDB_OK:
            raise capi.LdbError(st, capi.host_lib().ldb_plan_json_last_error().decode(errors="replace"))
        return Table(self.ctx, t)

    def stats(self):
        v = [C.c_int64() for _ in range(4)]
        capi.host_lib().ldb_plan_stats(self.h, *[C.byref(x) for x in v])
        out = dict(zip(("executions", "replays", "misses", "readbacks"), (x.value for x in v)))
        a, b = C.c_double(), C.c_double()
        capi.host_lib().ldb_plan_times(self.h, C.byref(a), C.byref(b))
        out["issue_ms"], out["wait_ms"] = a.value, b.value
        return out

    def release(self):
        if self.h and self.ctx.h:
            capi.host_lib().ldb_plan_release(self.h)
        self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Context:
    def __init__(self, device=0, stream=None):
        self.lib = capi.gpu_lib()
        h = C.c_void_p()
        check(self.lib.ldb_gpu_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        import weakref

        self._plans = weakref.WeakSet()  # prepared plans hold a trace of this context: released before it

    def close(self):
        if self.h:
            for p in list(self._plans):
                p.release()
            self.lib.ldb_gpu_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        check(self.lib.ldb_gpu_ctx_sync(self.h))

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, free, total = C.c_int32(), C.c_int64(), C.c_int64()
        check(self.lib.ldb_gpu_device_info(self.h, name, 256, C.byref(cus), C.byref(free), C.byref(total)))
        return {"name": name.value.decode(), "cus": cus.value, "hbm_free": free.value, "hbm_total": total.value}

    # ---- timers (HIP events on the ctx stream)
    def timer(self):
        t = C.c_int32()
        check(self.lib.ldb_gpu_timer_create(self.h, C.byref(t)))
        return t.value

    def timer_start(self, t):
        check(self.lib.ldb_gpu_timer_start(self.h, t))

    def timer_stop(self, t):
        check(self.lib.ldb_gpu_timer_stop(self.h, t))

    def timer_ms(self, t):
        ms = C.c_float()
        check(self.lib.ldb_gpu_timer_elapsed_ms(self.h, t, C.byref(ms)))
        return ms.value

    # ---- per-kernel profiling (HIP events around the dominant kernels)
    def prof_enable(self, on=True):
        check(self.lib.ldb_gpu_prof_enable(self.h, 1 if on else 0))

    def prof_marker(self, marker_id):
        """empty kernel with a grid of `marker_id` workgroups: cuts a rocprofv3 kernel trace into per-query pieces"""
        check(self.lib.ldb_gpu_prof_marker(self.h, int(marker_id)))

    def prof_reset(self):
        check(self.lib.ldb_gpu_prof_reset(self.h))

    def prof_get(self, name):
        n, ms = C.c_int64(), C.c_double()
        check(self.lib.ldb_gpu_prof_get(self.h, name.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def prof_max(self, name):
        ms = C.c_double()
        check(self.lib.ldb_gpu_prof_get_max(self.h, name.encode(), C.byref(ms)))
        return ms.value

    def prof_all(self):
        buf = C.create_string_buffer(4096)
        check(self.lib.ldb_gpu_prof_names(self.h, buf, 4096))
        return {k: self.prof_get(k) for k in buf.value.decode().split("\n") if k}

    # ---- tables
    def register(self, name, table: pa.Table, narrow_decimals=False):
        """Arrow C Data Interface hand-over of host record batches (zero-copy on the host side)."""
        batches = table.to_batches()
        if not batches:
            batches = [pa.RecordBatch.from_arrays([pa.array([], type=f.type) for f in table.schema], schema=table.schema)]
        schema = capi.ArrowSchema()
        table.schema._export_to_c(C.addressof(schema))
        arrays = [capi.ArrowArray() for _ in batches]
        for a, b in zip(arrays, batches):
            b._export_to_c(C.addressof(a))
        ptrs = (C.POINTER(capi.ArrowArray) * len(arrays))(*[C.pointer(a) for a in arrays])
        h = C.c_void_p()
        try:
            check(self.lib.ldb_gpu_table_register(self.h, name.encode(), C.byref(schema), ptrs, len(arrays), int(narrow_decimals), C.byref(h)))
        finally:
            # the library copied everything to the device: release the exported structs
            rel_t = C.CFUNCTYPE(None, C.c_void_p)
            for a in arrays:
                if a.release:
                    rel_t(a.release)(C.addressof(a))
            if schema.release:
                rel_t(schema.release)(C.addressof(schema))
        return Table(self, h)

    def tpch_generate(self, table_id, n_orders, part=0, n_parts=1, cols=None, narrow_decimals=False, all_cols=None):
        mask = 0
        if cols is not None:
            for c in cols:
                mask |= 1 << c
        h = C.c_void_p()
        check(self.lib.ldb_gpu_tpch_generate(self.h, table_id, n_orders, part, n_parts, mask, int(narrow_decimals), C.byref(h)))
        return Table(self, h)

    # ---- plans: data (lingo-db_amd/plans/tpch/qN.json) interpreted by libldb_host.so; these wrappers only name the inputs
    def _tpch(self, q, **tables):
        return self.run_plan("tpch/q%d.json" % q, tables)

    def plan_q1(self, lineitem):
        return self._tpch(1, lineitem=lineitem)

    def plan_q6(self, lineitem):
        return self._tpch(6, lineitem=lineitem)

    def plan_q3(self, customer, orders, lineitem):
        return self._tpch(3, customer=customer, orders=orders, lineitem=lineitem)

    def plan_q4(self, orders, lineitem):
        return self._tpch(4, orders=orders, lineitem=lineitem)

    def plan_q12(self, orders, lineitem):
        return self._tpch(12, orders=orders, lineitem=lineitem)

    def plan_q5(self, customer, orders, lineitem, supplier, nation, region):
        return self._tpch(5, customer=customer, orders=orders, lineitem=lineitem, supplier=supplier, nation=nation, region=region)

    def plan_q7(self, customer, orders, lineitem, supplier, nation):
        return self._tpch(7, customer=customer, orders=orders, lineitem=lineitem, supplier=supplier, nation=nation)

    def plan_q8(self, part, supplier, lineitem, orders, customer, nation, region):
        return self._tpch(8, part=part, supplier=supplier, lineitem=lineitem, orders=orders, customer=customer, nation=nation, region=region)

    def plan_q14(self, part, lineitem):
        return self._tpch(14, part=part, lineitem=lineitem)

    def plan_q11(self, partsupp, supplier, nation):
        return self._tpch(11, partsupp=partsupp, supplier=supplier, nation=nation)

    def plan_q9(self, part, supplier, lineitem, partsupp, orders, nation):
        return self._tpch(9, part=part, supplier=supplier, lineitem=lineitem, partsupp=partsupp, orders=orders, nation=nation)

    def plan_q10(self, customer, orders, lineitem, nation):
        return self._tpch(10, customer=customer, orders=orders, lineitem=lineitem, nation=nation)

    def plan_q15(self, supplier, lineitem):
        return self._tpch(15, supplier=supplier, lineitem=lineitem)

    def plan_q18(self, customer, orders, lineitem):
        return self._tpch(18, customer=customer, orders=orders, lineitem=lineitem)

    def run_plan(self, plan, tables, comm=None):
        """interprets a JSON plan (text, or the name of a file under lingo-db_amd/plans/) over {name: Table};
        with a Comm the plan's allgather / shuffle steps exchange rows with the other ranks"""
        import os

        text = plan
        if not plan.lstrip().startswith("{"):
            path = plan if os.path.exists(plan) else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plans", plan)
            with open(path) as f:
                text = f.read()
        names = list(tables)
        narr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        tarr = (C.c_void_p * len(names))(*[tables[n].h for n in names])
        t = C.c_void_p()
        st = capi.host_lib().ldb_plan_run_json_comm(self.h, comm.h if comm is not None else None, text.encode(), narr, tarr, len(names), C.byref(t))
        if st != capi.LDB_OK:
            raise capi.LdbError(st, capi.host_lib().ldb_plan_json_last_error().decode(errors="replace"))
        return Table(self, t)

    def prepare_plan(self, plan):
        """parses a JSON plan once (ldb_plan_prepare); PreparedPlan.execute runs it — from the second execution over the same
        tables without a host wait between operators and without descriptor uploads"""
        import os

        text = plan
        if not plan.lstrip().startswith("{"):
            path = plan if os.path.exists(plan) else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plans", plan)
            with open(path) as f:
                text = f.read()
        return PreparedPlan(self, text)

    def desc_cache_stats(self):
        h, m, b = C.c_int64(), C.c_int64(), C.c_int64()
        check(self.lib.ldb_gpu_desc_cache_stats(self.h, C.byref(h), C.byref(m), C.byref(b)))
        held, under = C.c_int64(), C.c_int64()
        check(self.lib.ldb_gpu_desc_cache_held(self.h, C.byref(held), C.byref(under)))
        return {"hits": h.value, "misses": m.value, "bytes": b.value, "held": held.value, "underflows": under.value}

    def run_subop_dump(self, dump, tables, name="subop_dump", comm=None):
        """runs a query from the reference's sub-operator dump (tools/ct/mlir-subop-to-json.cpp output, text or path):
        translate_subop_dump → run_plan"""
        return self.run_plan(translate_subop_dump(dump, name)[0], tables, comm=comm)

    def load_ipc(self, name, path, narrow_decimals=False):
        """registers one Arrow IPC file as a table — the reference keeps one `<table>.arrow` IPC
        file per table and reads all its record batches (LingoDBTable.cpp:27-54, loadTable).  The library maps
        and parses the file itself (ldb_gpu_table_load_ipc, csrc/ldb_ipc.hip); pyarrow is not involved."""
        t = C.c_void_p()
        check(self.lib.ldb_gpu_table_load_ipc(self.h, name.encode(), str(path).encode(), int(narrow_decimals), C.byref(t)))
        return Table(self, t)
