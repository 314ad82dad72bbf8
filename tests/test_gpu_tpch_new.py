"""TPC-H Q2 / Q13 / Q16 / Q17 / Q19 / Q20 / Q21 / Q22 as JSON plans (lingo-db_amd/plans/tpch/*.json →
libldb_host.so's plan interpreter → C-ABI → HIP kernels) against an independent evaluation of the
SQL text (resources/sql/tpch/*.sql of the reference) with pandas / Python integers over the same
generated tables (tests/tpch_sql.py).  Decimals are compared as unscaled integers: bit-exact.
Runs twice (conftest kernel_mode): library defaults, then every kernel run-time specialised and every
base-table filter fused lazily into its consumer."""
import pyarrow as pa
import pytest

import tpch_data as T
import tpch_sql

pytestmark = pytest.mark.gpu
N = 300_000  # SF 0.2


def result_rows(table):
    out = []
    for i in range(table.num_columns):
        col = table.column(i).combine_chunks()
        t = col.type
        if pa.types.is_decimal(t):
            out.append([None if v.as_py() is None else int(v.as_py().scaleb(t.scale)) for v in col])
        elif pa.types.is_date32(t):
            out.append([(v.as_py() - tpch_sql.EPOCH).days for v in col])
        else:
            out.append(col.to_pylist())
    return list(zip(*out)) if out else []


@pytest.fixture(scope="module")
def host():
    return tpch_sql.Tables(N)


@pytest.fixture(scope="module")
def dev(ctx, host):
    cache = {}

    def get(q):
        tabs = {}
        for name, (tid, cols) in tpch_sql.INPUTS[q].items():
            key = (tid, tuple(cols))
            if key not in cache:
                cache[key] = ctx.register(name, host.arrow(tid, cols))
            tabs[name] = cache[key]
        return tabs

    return get


MIN_ROWS = {2: 50, 13: 10, 16: 1000, 17: 1, 19: 1, 20: 3, 21: 20, 22: 7}


@pytest.mark.parametrize("q", sorted(tpch_sql.SQL))
def test_query(ctx, host, dev, q):
    want = tpch_sql.evaluate(q, host)
    assert len(want) >= MIN_ROWS[q] and want[0][0] is not None
    got = ctx.run_plan("tpch/q%d.json" % q, dev(q)).to_arrow()
    if q == 19:
        assert got.schema.field(0).type == pa.decimal128(33, 4)
    if q == 17:
        assert got.schema.field(0).type == pa.decimal128(17, 6)
    assert result_rows(got) == want


def test_plan_errors_are_reported(ctx, dev):
    """unknown columns / operators come back as an error string, never as a crash"""
    from lingodb_amd import capi

    gna = dev(20)["nation"]
    for bad in ('{"steps": [{"op": "filter", "in": "nation", "out": "x", "preds": [{"col": "nope", "op": "EQ", "value": 1}]}], "result": "x"}',
                '{"steps": [{"op": "frobnicate", "out": "x"}], "result": "x"}', '{"steps": [', '{"steps": [], "result": "nation"}'):
        with pytest.raises(capi.LdbError):
            ctx.run_plan(bad, {"nation": gna})
