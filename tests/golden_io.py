"""Loaders of tests/golden/ (answers of the reference's own runtime objects, written by
tests/golden/make_ref_golden.py) shared by the oracle test and the GPU test."""
import json
import os

import numpy as np
import pyarrow as pa

from lingodb_amd import api, capi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FOP = {"EQ": capi.F_EQ, "NEQ": capi.F_NEQ, "LT": capi.F_LT, "LTE": capi.F_LTE, "GT": capi.F_GT, "GTE": capi.F_GTE, "IN": capi.F_IN}


def types_table():
    with pa.OSFile(os.path.join(GOLDEN, "ref_types.arrow"), "rb") as f:
        return pa.ipc.open_file(f).read_all()


def hash_cases():
    """[(key column list, uint64 hashes)]"""
    z = np.load(os.path.join(GOLDEN, "ref_types_hash.npz"))
    return [([int(c) for c in name.split(",")], z[name]) for name in z.files]


def filter_cases():
    with open(os.path.join(GOLDEN, "ref_filters.json")) as f:
        meta = json.load(f)
    z = np.load(os.path.join(GOLDEN, "ref_filters.npz"))
    return meta, [(case, z[str(i)]) for i, case in enumerate(meta["cases"])]


def constant_of(field, text):
    """the reference parses the constant of a pushed-down filter from its SQL text
    (Restrictions.cpp:441-480: date string → days, decimal string → unscaled at the column's scale,
    char(1) → its byte); the C-ABI takes the parsed value — this is the binding's share"""
    import datetime
    import decimal

    t = field.type
    if pa.types.is_date32(t):
        return (datetime.date.fromisoformat(text) - datetime.date(1970, 1, 1)).days
    if pa.types.is_decimal(t):
        v = decimal.Decimal(str(text)).scaleb(t.scale)
        assert v == v.to_integral_value()
        return int(v)
    if pa.types.is_fixed_size_binary(t):
        return ord(text)
    return text


def preds_of(case, schema):
    out = []
    for f in case:
        col = schema.get_field_index(f["col"])
        op = FOP[f["op"]]
        if op == capi.F_IN:
            out.append(api.pred((0, col), op, values=f["in"]))
        else:
            out.append(api.pred((0, col), op, constant_of(schema.field(col), f["v"])))
    return out


def npz(name):
    return np.load(os.path.join(GOLDEN, name))


def like_cases():
    with open(os.path.join(GOLDEN, "ref_like.json")) as f:
        return json.load(f)


def year_cases():
    with open(os.path.join(GOLDEN, "ref_extract_year.json")) as f:
        return json.load(f)


# ---------------------------------------------------------------- test/sqlite-small/join.test
def sqlite_join_cases():
    with open(os.path.join(GOLDEN, "sqlite_small_join.json")) as f:
        return json.load(f)


def run_sqlite_join_case(case, join):
    """Evaluates one transcribed join.test case with `join(build_values, probe_values, kind) ->
    (probe_rows, build_rows, marks)` (row indices into the two value lists; build_rows holds
    capi.LDB_NULL_ROW for the unmatched rows of an outer join; marks only for JOIN_MARK) and
    returns the result rows in the reference's `rowsort` order.  The mapping SQL → operator is
    the reference's: the preserved side of an outer join probes, EXISTS / NOT EXISTS are semi /
    anti joins of the outer query's rows, `= some` is a mark join
    (RelAlgToSubOp.cpp:1217-1294), a full outer join adds the unmatched build rows."""
    s, t, kind = case["s"], case["t"], case["kind"]
    val = lambda side, r: None if r == capi.LDB_NULL_ROW else side[r]  # noqa: E731
    if kind == "inner":
        pr, br, _ = join(t, s, capi.JOIN_INNER)
        rows = [[s[p]] for p in pr]
    elif kind == "left_outer":
        pr, br, _ = join(t, s, capi.JOIN_LEFT_OUTER)
        rows = [[s[p], val(t, b)] for p, b in zip(pr, br)]
    elif kind == "right_outer":
        pr, br, _ = join(s, t, capi.JOIN_LEFT_OUTER)
        rows = [[val(s, b), t[p]] for p, b in zip(pr, br)]
    elif kind == "full_outer":
        pr, br, _ = join(t, s, capi.JOIN_LEFT_OUTER)
        rows = [[s[p], val(t, b)] for p, b in zip(pr, br)]
        ur, _, _ = join(t, s, capi.JOIN_ANTI_BUILD)
        rows += [[None, t[u]] for u in ur]
    elif kind in ("semi", "anti"):
        pr, _, _ = join(t, s, capi.JOIN_SEMI if kind == "semi" else capi.JOIN_ANTI)
        rows = [[s[p]] for p in pr]
    else:
        pr, _, marks = join(t, s, capi.JOIN_MARK)
        rows = [[s[p], bool(m)] for p, m in zip(pr, marks)]
    key = lambda r: [(v is None, v) for v in r]  # noqa: E731
    return sorted(rows, key=key), sorted(case["expected"], key=key)
