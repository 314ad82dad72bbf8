"""ctypes binding of the C-ABI in include/lingodb_gpu.h (+ include/ldb_tpchgen.h, host/ldb_host.hpp).

This is a test/bench harness: the product is liblingodb_gpu.so.  Loading fails loudly when the
HIP library has not been built — there is no CPU fallback anywhere in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)  # lingo-db_amd/
GPU_LIB_PATH = os.path.join(PKG_ROOT, "liblingodb_gpu.so")
HOST_LIB_PATH = os.path.join(PKG_ROOT, "libldb_host.so")

LDB_MAX_SIDES = 6
LDB_NULL_ROW = 0xFFFFFFFF
LDB_MAX_FACTORS = 3
LDB_MAX_TERMS = 2
LDB_MAX_AGG_PREDS = 3

# ldb_status
LDB_OK = 0
LDB_ERR_INVALID = -1
LDB_ERR_UNSUPPORTED = -2
LDB_ERR_NO_DEVICE = -5

# ldb_type
T_INT8, T_INT16, T_INT32, T_INT64, T_DATE32, T_DECIMAL128, T_CHAR4, T_UTF8, T_FLOAT64, T_FLOAT32, T_BOOL8 = range(11)
# ldb_filter_op / rhs kind
F_EQ, F_NEQ, F_LT, F_LTE, F_GT, F_GTE, F_NOTNULL, F_IN, F_LIKE, F_NOT_LIKE = range(10)
RHS_INT, RHS_STRING, RHS_COLUMN, RHS_FLOAT = range(4)
# ldb_agg_fn
AGG_SUM, AGG_MIN, AGG_MAX, AGG_COUNT, AGG_COUNT_STAR, AGG_ANY, AGG_AVG = range(7)
# ldb_join_kind
JOIN_INNER, JOIN_SEMI, JOIN_ANTI, JOIN_LEFT_OUTER, JOIN_MARK, JOIN_SINGLE, JOIN_SEMI_BUILD, JOIN_ANTI_BUILD, JOIN_RIGHT_OUTER, JOIN_FULL_OUTER = range(10)
SET_UNION_ALL, SET_UNION, SET_INTERSECT, SET_INTERSECT_ALL, SET_EXCEPT, SET_EXCEPT_ALL = range(6)
WIN_RANK, WIN_SUM, WIN_MIN, WIN_MAX, WIN_COUNT, WIN_COUNT_STAR = range(6)
FRAME_UNBOUNDED_PRECEDING, FRAME_UNBOUNDED_FOLLOWING = -(2 ** 63), 2 ** 63 - 1
# ldb_scalar_fn
FN_EXTRACT_YEAR = 0


class ColType(C.Structure):
    _fields_ = [("type", C.c_int32), ("precision", C.c_int32), ("scale", C.c_int32), ("nullable", C.c_int32)]


class ColRef(C.Structure):
    _fields_ = [("side", C.c_int32), ("col", C.c_int32)]


class FilterDesc(C.Structure):
    _fields_ = [
        ("col", ColRef),
        ("op", C.c_int32),
        ("rhs_kind", C.c_int32),
        ("value_lo", C.c_uint64),
        ("value_hi", C.c_int64),
        ("value_f64", C.c_double),
        ("str", C.c_char_p),
        ("str_len", C.c_int32),
        ("rhs_col", ColRef),
        ("n_in", C.c_int32),
        ("in_values", C.POINTER(C.c_int64)),
        ("in_strs", C.POINTER(C.c_char_p)),
        ("in_str_lens", C.POINTER(C.c_int32)),
    ]


class Factor(C.Structure):
    _fields_ = [("has_col", C.c_int32), ("col", ColRef), ("a", C.c_int64), ("b", C.c_int64)]


class Term(C.Structure):
    _fields_ = [("n_factors", C.c_int32), ("negate", C.c_int32), ("div_pow10", C.c_int32), ("reserved", C.c_int32), ("f", Factor * LDB_MAX_FACTORS)]


class Expr(C.Structure):
    _fields_ = [("n_terms", C.c_int32), ("is_float", C.c_int32), ("t", Term * LDB_MAX_TERMS)]


class AggSpec(C.Structure):
    _fields_ = [
        ("fn", C.c_int32),
        ("wide", C.c_int32),
        ("arg", Expr),
        ("n_preds", C.c_int32),
        ("preds", FilterDesc * LDB_MAX_AGG_PREDS),
        ("avg_pow10", C.c_int32),
        ("out_type", C.c_int32),
        ("out_precision", C.c_int32),
        ("out_scale", C.c_int32),
        ("has_count_expr", C.c_int32),
        ("count_expr", Expr),
    ]


class SortSpec(C.Structure):
    _fields_ = [("col", ColRef), ("descending", C.c_int32), ("reserved", C.c_int32)]


class WindowFn(C.Structure):
    _fields_ = [("fn", C.c_int32), ("col", ColRef)]


class JoinResidual(C.Structure):
    _fields_ = [("probe_col", ColRef), ("build_col", ColRef), ("op", C.c_int32), ("reserved", C.c_int32)]


class XInstr(C.Structure):
    _fields_ = [("op", C.c_int32), ("arg", C.c_int32), ("col", ColRef), ("lo", C.c_int64), ("hi", C.c_int64)]


X_COL, X_CONST, X_ADD, X_SUB, X_MUL, X_SDIV, X_MUL_POW10, X_SDIV_POW10, X_NEG, X_CMP, X_AND, X_OR, X_NOT, X_SELECT, X_ISNULL, X_COALESCE, X_ROW = range(17)


class CommStats(C.Structure):
    _fields_ = [("groups", C.c_int64), ("bytes_out", C.c_int64), ("bytes_in", C.c_int64), ("max_peer_bytes_out", C.c_int64), ("host_ms", C.c_double), ("device_ms", C.c_double)]


class ArrowSchema(C.Structure):
    pass


ArrowSchema._fields_ = [
    ("format", C.c_char_p),
    ("name", C.c_char_p),
    ("metadata", C.c_char_p),
    ("flags", C.c_int64),
    ("n_children", C.c_int64),
    ("children", C.POINTER(C.POINTER(ArrowSchema))),
    ("dictionary", C.POINTER(ArrowSchema)),
    ("release", C.c_void_p),
    ("private_data", C.c_void_p),
]


class ArrowArray(C.Structure):
    pass


ArrowArray._fields_ = [
    ("length", C.c_int64),
    ("null_count", C.c_int64),
    ("offset", C.c_int64),
    ("n_buffers", C.c_int64),
    ("n_children", C.c_int64),
    ("buffers", C.POINTER(C.c_void_p)),
    ("children", C.POINTER(C.POINTER(ArrowArray))),
    ("dictionary", C.POINTER(ArrowArray)),
    ("release", C.c_void_p),
    ("private_data", C.c_void_p),
]

P = C.c_void_p
PP = C.POINTER(C.c_void_p)
i32, i64, u64 = C.c_int32, C.c_int64, C.c_uint64

# name -> (restype, argtypes); every symbol declared in include/lingodb_gpu.h
GPU_API = {
    "ldb_gpu_ctx_create": (i32, [i32, P, PP]),
    "ldb_gpu_ctx_destroy": (i32, [P]),
    "ldb_gpu_ctx_sync": (i32, [P]),
    "ldb_gpu_last_error": (C.c_char_p, []),
    "ldb_gpu_device_info": (i32, [P, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i64), C.POINTER(i64)]),
    "ldb_gpu_timer_create": (i32, [P, C.POINTER(i32)]),
    "ldb_gpu_timer_start": (i32, [P, i32]),
    "ldb_gpu_timer_stop": (i32, [P, i32]),
    "ldb_gpu_timer_elapsed_ms": (i32, [P, i32, C.POINTER(C.c_float)]),
    "ldb_gpu_prof_enable": (i32, [P, i32]),
    "ldb_gpu_prof_marker": (i32, [P, i32]),
    "ldb_gpu_like_plan": (i32, [C.c_char_p, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "ldb_gpu_prof_reset": (i32, [P]),
    "ldb_gpu_prof_get": (i32, [P, C.c_char_p, C.POINTER(i64), C.POINTER(C.c_double)]),
    "ldb_gpu_prof_get_max": (i32, [P, C.c_char_p, C.POINTER(C.c_double)]),
    "ldb_gpu_prof_names": (i32, [P, C.c_char_p, i32]),
    "ldb_gpu_jit_stats": (i32, [C.POINTER(i64), C.POINTER(i64), C.POINTER(C.c_double)]),
    "ldb_gpu_jit_compile_check": (i32, [C.c_char_p, i32]),
    "ldb_gpu_jit_wait": (i32, [i64, C.POINTER(i64)]),
    "ldb_gpu_jit_info": (i32, [C.POINTER(i64), i32]),
    "ldb_gpu_jit_shutdown": (i32, []),
    "ldb_gpu_jit_cache_selftest": (i32, [C.c_char_p, i32]),
    "ldb_gpu_set_option": (i32, [C.c_char_p, i64]),
    "ldb_gpu_get_option": (i64, [C.c_char_p]),
    "ldb_gpu_table_register": (i32, [P, C.c_char_p, C.POINTER(ArrowSchema), C.POINTER(C.POINTER(ArrowArray)), i64, i32, PP]),
    "ldb_gpu_table_load_ipc": (i32, [P, C.c_char_p, C.c_char_p, i32, PP]),
    "ldb_gpu_table_zones": (i64, [P, P, i32]),
    "ldb_gpu_ipc_describe": (i32, [C.c_char_p, C.c_char_p, i64]),
    "ldb_gpu_table_alloc": (i32, [P, C.c_char_p, i32, C.POINTER(ColType), C.POINTER(C.c_char_p), i64, C.POINTER(i64), i32, PP]),
    "ldb_gpu_table_release": (i32, [P, P]),
    "ldb_gpu_table_rows": (i64, [P]),
    "ldb_gpu_table_cols": (i32, [P]),
    "ldb_gpu_table_coltype": (i32, [P, i32, C.POINTER(ColType)]),
    "ldb_gpu_table_col_index": (i32, [P, C.c_char_p]),
    "ldb_gpu_table_col_name": (C.c_char_p, [P, i32]),
    "ldb_gpu_table_rename_col": (i32, [P, i32, C.c_char_p]),
    "ldb_gpu_table_col_width": (i32, [P, i32]),
    "ldb_gpu_table_col_ptrs": (i32, [P, i32, PP, PP, PP, C.POINTER(i64)]),
    "ldb_gpu_table_set_rows": (i32, [P, i64]),
    "ldb_gpu_table_read_fixed": (i32, [P, P, i32, P, i64]),
    "ldb_gpu_table_dict_encode": (i32, [P, P, i32, C.POINTER(C.c_int32)]),
    "ldb_gpu_table_dict_size": (i32, [P, i32]),
    "ldb_gpu_table_row_valid": (i32, [P, P, i32, i64, C.POINTER(C.c_int32)]),
    "ldb_gpu_table_write_fixed": (i32, [P, P, i32, P, i64]),
    "ldb_gpu_memcpy_d2d": (i32, [P, P, P, i64]),
    "ldb_gpu_export": (i32, [P, P, C.POINTER(ArrowSchema), C.POINTER(ArrowArray)]),
    "ldb_gpu_rel_from_table": (i32, [P, P, PP]),
    "ldb_gpu_rel_release": (i32, [P, P]),
    "ldb_gpu_rel_rows": (i64, [P, P]),
    "ldb_gpu_rel_sides": (i32, [P]),
    "ldb_gpu_rel_read_rowids": (i32, [P, P, i32, C.POINTER(C.c_uint32), i64]),
    "ldb_gpu_materialize": (i32, [P, P, C.POINTER(ColRef), i32, PP]),
    "ldb_gpu_scan_filter": (i32, [P, P, C.POINTER(FilterDesc), i32, PP]),
    "ldb_gpu_scan_filter_dnf": (i32, [P, P, C.POINTER(FilterDesc), C.POINTER(i32), i32, PP]),
    "ldb_gpu_map_expr": (i32, [P, P, C.POINTER(XInstr), i32, ColType, C.c_char_p, PP]),
    "ldb_gpu_map_substr": (i32, [P, P, ColRef, i64, i64, C.c_char_p, PP]),
    "ldb_gpu_scan_count": (i32, [P, P, C.POINTER(FilterDesc), i32, C.POINTER(i64)]),
    "ldb_gpu_hash_keys": (i32, [P, P, C.POINTER(ColRef), i32, PP]),
    "ldb_gpu_map_column": (i32, [P, P, ColRef, i32, C.c_char_p, PP]),
    "ldb_gpu_rel_zip": (i32, [P, P, P, PP]),
    "ldb_gpu_map_muldiv": (i32, [P, P, ColRef, i64, i64, i32, i32, ColRef, i32, i32, C.c_char_p, PP]),
    "ldb_gpu_groupby": (i32, [P, P, C.POINTER(FilterDesc), i32, C.POINTER(ColRef), i32, C.POINTER(AggSpec), i32, i64, PP]),
    "ldb_gpu_join_build": (i32, [P, P, C.POINTER(ColRef), i32, i32, PP]),
    "ldb_gpu_hashtable_release": (i32, [P, P]),
    "ldb_gpu_table_index": (i32, [P, P, C.POINTER(C.c_int32), i32, PP]),
    "ldb_gpu_hashtable_slots": (i64, [P]),
    "ldb_gpu_hashtable_bytes": (i64, [P]),
    "ldb_gpu_join_probe": (i32, [P, P, P, C.POINTER(ColRef), i32, i32, PP, PP]),
    "ldb_gpu_join_probe_residual": (i32, [P, P, P, C.POINTER(ColRef), i32, i32, C.POINTER(JoinResidual), i32, PP, PP]),
    "ldb_gpu_join_probe_semi_anti_build": (i32, [P, P, P, C.POINTER(ColRef), i32, C.POINTER(JoinResidual), i32, C.POINTER(FilterDesc), i32, PP]),
    "ldb_gpu_join_nl": (i32, [P, P, P, i32, C.POINTER(JoinResidual), i32, PP, PP]),
    "ldb_gpu_join_probe_count": (i32, [P, P, P, C.POINTER(ColRef), i32, C.POINTER(i64)]),
    "ldb_gpu_sort": (i32, [P, P, C.POINTER(SortSpec), i32, PP]),
    "ldb_gpu_topk": (i32, [P, P, C.POINTER(SortSpec), i32, i64, PP]),
    "ldb_gpu_partition": (i32, [P, P, C.POINTER(ColRef), i32, i32, C.POINTER(ColRef), i32, PP, C.POINTER(i64)]),
    "ldb_gpu_comm_unique_id": (i32, [P]),
    "ldb_gpu_comm_create": (i32, [P, i32, i32, P, PP]),
    "ldb_gpu_comm_destroy": (i32, [P]),
    "ldb_gpu_comm_rank": (i32, [P]),
    "ldb_gpu_comm_world": (i32, [P]),
    "ldb_gpu_set_op": (i32, [P, P, C.POINTER(ColRef), P, C.POINTER(ColRef), i32, i32, PP]),
    "ldb_gpu_window": (i32, [P, P, C.POINTER(ColRef), i32, C.POINTER(SortSpec), i32, i64, i64, C.POINTER(WindowFn), i32, PP, PP]),
    "ldb_gpu_comm_transport": (C.c_char_p, [P]),
    "ldb_gpu_comm_stats": (i32, [P, C.POINTER(CommStats), i32]),
    "ldb_gpu_comm_available": (i32, []),
    "ldb_gpu_comm_create_host": (i32, [i32, i32, P, PP]),
    "ldb_gpu_comm_alltoall_bytes": (i32, [P, P, C.POINTER(C.c_int64), P, C.POINTER(C.c_int64)]),
    "ldb_gpu_allgather": (i32, [P, P, P, C.c_char_p, PP]),
    "ldb_gpu_alltoall": (i32, [P, P, P, C.POINTER(i64), C.c_char_p, PP]),
    "ldb_gpu_shuffle": (i32, [P, P, P, C.POINTER(ColRef), i32, C.POINTER(ColRef), i32, C.c_char_p, PP]),
    "ldb_gpu_option_epoch": (i64, []),
    "ldb_gpu_trace_create": (i32, [P, PP]),
    "ldb_gpu_trace_destroy": (i32, [P, P]),
    "ldb_gpu_trace_begin": (i32, [P, P, i32]),
    "ldb_gpu_trace_end": (i32, [P, C.POINTER(i32)]),
    "ldb_gpu_trace_replayable": (i32, [P]),
    "ldb_gpu_comm_agree": (i32, [P, P, i32, C.POINTER(i32)]),
    "ldb_gpu_trace_stats": (i32, [P, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
    "ldb_gpu_desc_cache_stats": (i32, [P, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
    "ldb_gpu_desc_cache_held": (i32, [P, C.POINTER(i64), C.POINTER(i64)]),
    "ldb_gpu_order_dependent_misses": (i64, []),
    "ldb_gpu_table_stamp": (u64, [P]),
    # include/ldb_tpchgen.h (device generator)
    "ldb_gpu_tpch_generate": (i32, [P, i32, i64, i32, i32, u64, i32, PP]),
}

HOST_API = {
    "ldb_tpch_host_rows": (i64, [i32, i64, i32, i32]),
    "ldb_tpch_host_column": (i64, [i32, i32, i64, i32, i32, P, C.POINTER(i64), C.POINTER(i64)]),
    "ldb_plan_last_error": (C.c_char_p, []),
    "ldb_plan_run_json": (i32, [P, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(P), i32, PP]),
    "ldb_plan_run_json_comm": (i32, [P, P, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(P), i32, PP]),
    "ldb_plan_json_last_error": (C.c_char_p, []),
    "ldb_plan_prepare": (i32, [P, C.c_char_p, PP]),
    "ldb_plan_execute": (i32, [P, P, C.POINTER(C.c_char_p), C.POINTER(P), i32, PP]),
    "ldb_plan_release": (i32, [P]),
    "ldb_plan_stats": (i32, [P, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
    "ldb_plan_times": (i32, [P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ldb_plan_json_check": (i32, [C.c_char_p, C.POINTER(C.c_char_p), i32]),
    "ldb_subop_translate": (i32, [C.c_char_p, C.c_char_p, C.c_char_p, i64, C.POINTER(i64)]),
    "ldb_subop_last_error": (C.c_char_p, []),
    "ldb_subop_report": (C.c_char_p, []),
    "ldb_host_parse_date32": (i32, [C.c_char_p, C.POINTER(i32)]),
    "ldb_host_parse_decimal": (i32, [C.c_char_p, i32, C.POINTER(i64), C.POINTER(i64)]),
    "ldb_host_decimal_type": (None, [i32, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]),
}


class LibraryMissing(RuntimeError):
    pass


def _bind(lib, api):
    for name, (res, args) in api.items():
        fn = getattr(lib, name)  # AttributeError = symbol missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


_gpu = None
_host = None


def gpu_lib():
    """liblingodb_gpu.so (HIP kernels + C-ABI).  Raises if it has not been built."""
    global _gpu
    if _gpu is None:
        if not os.path.exists(GPU_LIB_PATH):
            raise LibraryMissing(f"{GPU_LIB_PATH} not built — run `python -c 'import __graft_entry__ as g; g.build()'` " "(there is no CPU fallback)")
        _gpu = _bind(C.CDLL(GPU_LIB_PATH, mode=C.RTLD_GLOBAL), GPU_API)
        import atexit

        atexit.register(_gpu.ldb_gpu_jit_shutdown)  # compile workers stop before the interpreter (and hiprtc's statics) go away
    return _gpu


def host_lib():
    global _host
    if _host is None:
        gpu_lib()
        if not os.path.exists(HOST_LIB_PATH):
            raise LibraryMissing(f"{HOST_LIB_PATH} not built")
        _host = _bind(C.CDLL(HOST_LIB_PATH), HOST_API)
    return _host


class LdbError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"ldb status {status}: {msg}")
        self.status = status


def check(status):
    if status != LDB_OK:
        raise LdbError(status, gpu_lib().ldb_gpu_last_error().decode(errors="replace"))


def check_plan(status):
    if status != LDB_OK:
        msg = host_lib().ldb_plan_last_error().decode(errors="replace")
        raise LdbError(status, msg)
