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
