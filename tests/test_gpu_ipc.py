"""f3 on the device: ldb_gpu_table_load_ipc (mmap + flatbuffer walk + register from the mapping) gives the table pyarrow
reads from the same file — every supported type, NULLs, several record batches, narrowed decimals, an empty file — and
a loaded table behaves like a registered one in a plan."""
import pyarrow as pa
import pytest

from test_ipc_loader import sample_table, write_ipc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("batch_rows,narrow", [(None, False), (97, False), (500, True)])
def test_load_ipc_round_trip(ctx, tmp_path, batch_rows, narrow):
    t = sample_table(3000)
    p = tmp_path / "t.arrow"
    write_ipc(p, t, batch_rows)
    got = ctx.load_ipc("t", p, narrow_decimals=narrow).to_arrow()
    assert got.num_rows == 3000 and got.schema.names == t.schema.names
    for name in t.schema.names:
        a, b = got.column(name).to_pylist(), t.column(name).to_pylist()
        assert a == b, name


def test_load_ipc_empty_and_errors(ctx, tmp_path):
    from lingodb_amd import capi

    write_ipc(tmp_path / "e.arrow", sample_table(0))
    e = ctx.load_ipc("e", tmp_path / "e.arrow")
    assert e.rows == 0 and e.n_cols == 9
    with pytest.raises(capi.LdbError):
        ctx.load_ipc("x", tmp_path / "missing.arrow")


def test_loaded_table_runs_a_plan_like_a_registered_one(ctx, tmp_path):
    import json

    t = sample_table(5000)
    write_ipc(tmp_path / "t.arrow", t, 1024)
    plan = json.dumps({"name": "p", "inputs": ["t"], "steps": [
        {"op": "groupby", "in": "t", "keys": ["tiny"], "aggs": [{"fn": "sum", "expr": "price", "as": "s"}, {"fn": "count_star", "as": "n"}],
         "preds": [{"col": "day", "op": "GTE", "value": "1994-01-01"}], "out": "g"},
        {"op": "sort", "in": "g", "by": ["tiny"], "out": "s"},
        {"op": "materialize", "in": "s", "cols": ["tiny", "s", "n"], "out": "result"}], "result": "result"})
    a = ctx.run_plan(plan, {"t": ctx.load_ipc("t", tmp_path / "t.arrow")}).to_arrow().to_pylist()
    b = ctx.run_plan(plan, {"t": ctx.register("t", t)}).to_arrow().to_pylist()
    assert a == b and len(a) == 100
