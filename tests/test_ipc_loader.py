"""f3: the Arrow IPC file reader inside the library (csrc/ldb_ipc.hip; the reference reads one `<table>.arrow` per table
with arrow::ipc::RecordBatchFileReader, LingoDBTable.cpp:27-54).  CPU half: the parse alone against pyarrow's view of
files pyarrow wrote, and malformed / unsupported files.  GPU half: tests/test_gpu_ipc.py."""
import datetime
import decimal
import json
import os

import pyarrow as pa
import pytest

from lingodb_amd import api, capi


def sample_table(n=1000):
    D = decimal.Decimal
    return pa.table({
        "k": pa.array(range(n), pa.int32()),
        "big": pa.array([None if i % 11 == 0 else i * 10**12 for i in range(n)], pa.int64()),
        "price": pa.array([D(i * 7 - 300) / 100 for i in range(n)], pa.decimal128(12, 2)),
        "wide": pa.array([None if i % 5 == 0 else D(i) * D(10)**20 for i in range(n)], pa.decimal128(38, 0)),
        "day": pa.array([datetime.date(1992, 1, 1) + datetime.timedelta(days=i % 2500) for i in range(n)], pa.date32()),
        "name": pa.array([None if i % 13 == 0 else "row %d %s" % (i, "x" * (i % 17)) for i in range(n)], pa.string()),
        "flag": pa.array([bytes([65 + i % 3, 0, 0, 0]) for i in range(n)], pa.binary(4)),
        "f": pa.array([i / 7 for i in range(n)], pa.float64()),
        "tiny": pa.array([i % 100 - 50 for i in range(n)], pa.int8()),
    })


def write_ipc(path, table, batch_rows=None, **opts):
    with pa.OSFile(str(path), "wb") as f:
        with pa.ipc.new_file(f, table.schema, options=pa.ipc.IpcWriteOptions(**opts) if opts else None) as w:
            if table.num_rows == 0:
                pass
            elif batch_rows:
                for b in table.to_batches(max_chunksize=batch_rows):
                    w.write_batch(b)
            else:
                w.write_table(table)


FORMATS = {"k": "i", "big": "l", "price": "d:12,2", "wide": "d:38,0", "day": "tdD", "name": "u", "flag": "w:4", "f": "g", "tiny": "c"}


def test_describe_matches_pyarrow(tmp_path):
    t = sample_table(1000)
    for batch_rows in (None, 64, 333):
        p = tmp_path / ("t%s.arrow" % batch_rows)
        write_ipc(p, t, batch_rows)
        d = api.describe_ipc(p)
        assert [c["name"] for c in d["columns"]] == t.schema.names
        assert {c["name"]: c["format"] for c in d["columns"]} == FORMATS
        assert all(c["nullable"] for c in d["columns"])
        with pa.OSFile(str(p), "rb") as f:
            r = pa.ipc.open_file(f)
            assert d["batches"] == [r.get_batch(i).num_rows for i in range(r.num_record_batches)]
        assert d["rows"] == 1000


def test_empty_file_and_large_utf8(tmp_path):
    t = sample_table(0)
    write_ipc(tmp_path / "e.arrow", t)
    d = api.describe_ipc(tmp_path / "e.arrow")
    assert d["rows"] == 0 and d["batches"] == [] and len(d["columns"]) == 9
    t2 = pa.table({"s": pa.array(["a", None, "ccc"], pa.large_string()), "n": pa.array([1, 2, 3], pa.int16())})
    write_ipc(tmp_path / "l.arrow", t2)
    assert [c["format"] for c in api.describe_ipc(tmp_path / "l.arrow")["columns"]] == ["U", "s"]


def test_unsupported_and_malformed_files_are_rejected(tmp_path):
    lib = capi.gpu_lib()

    def err(path):
        with pytest.raises(capi.LdbError) as e:
            api.describe_ipc(path)
        return e.value

    t = sample_table(200)
    # a stream is not a file
    with pa.OSFile(str(tmp_path / "s.arrows"), "wb") as f:
        with pa.ipc.new_stream(f, t.schema) as w:
            w.write_table(t)
    assert "ARROW1" in str(err(tmp_path / "s.arrows"))
    # compressed bodies, dictionary columns, unsigned / bool / nested columns: reported, not misread
    write_ipc(tmp_path / "z.arrow", t, compression="lz4")
    e = err(tmp_path / "z.arrow")
    assert e.status == capi.LDB_ERR_UNSUPPORTED and "compressed" in str(e)
    for name, col in (("dict", pa.array(["a", "b", "a"]).dictionary_encode()), ("u", pa.array([1, 2, 3], pa.uint32())), ("b", pa.array([True, False, None])),
                      ("l", pa.array([[1], [2, 3], None], pa.list_(pa.int32())))):
        write_ipc(tmp_path / (name + ".arrow"), pa.table({"c": col}))
        e = err(tmp_path / (name + ".arrow"))
        assert e.status == capi.LDB_ERR_UNSUPPORTED and "'c'" in str(e), (name, str(e))
    # truncations and corruptions anywhere in the file: an error, never a crash or an out-of-file read
    write_ipc(tmp_path / "ok.arrow", t, 50)
    raw = (tmp_path / "ok.arrow").read_bytes()
    assert api.describe_ipc(tmp_path / "ok.arrow")["rows"] == 200
    for cut in (0, 5, 20, len(raw) // 2, len(raw) - 12, len(raw) - 1):
        (tmp_path / "cut.arrow").write_bytes(raw[:cut])
        assert err(tmp_path / "cut.arrow").status == capi.LDB_ERR_INVALID
    import random

    rng = random.Random(5)
    footer_len = int.from_bytes(raw[-10:-6], "little")
    for _ in range(300):  # flip bytes inside the footer (offsets, vtables, block table): error or a consistent parse
        b = bytearray(raw)
        at = len(raw) - 10 - rng.randrange(1, footer_len)
        b[at] = rng.randrange(256)
        (tmp_path / "flip.arrow").write_bytes(bytes(b))
        try:
            api.describe_ipc(tmp_path / "flip.arrow")
        except capi.LdbError as e2:
            assert e2.status in (capi.LDB_ERR_INVALID, capi.LDB_ERR_UNSUPPORTED)
    assert lib.ldb_gpu_ipc_describe(None, None, 0) == capi.LDB_ERR_INVALID
    assert err(tmp_path / "missing.arrow").status == capi.LDB_ERR_INVALID


def test_string_offsets_are_checked_row_by_row_and_zero_row_batches_load(tmp_path):
    """(round-3 ADVICE) every utf8 offset is validated on the host — negative, decreasing or past the data buffer —
    not just the last one; a zero-row batch (whatever offsets buffer its writer emitted) is accepted"""
    t = pa.table({"s": pa.array(["aa", "bbb", "", "cccc", "dd"], pa.string()), "k": pa.array(range(5), pa.int32())})
    write_ipc(tmp_path / "s.arrow", t)
    raw = bytearray((tmp_path / "s.arrow").read_bytes())
    want = (0).to_bytes(4, "little") + (2).to_bytes(4, "little") + (5).to_bytes(4, "little") + (5).to_bytes(4, "little") + (9).to_bytes(4, "little") + (11).to_bytes(4, "little")
    at = bytes(raw).find(want)
    assert at > 0, "offsets buffer not found in the file body"
    assert api.describe_ipc(tmp_path / "s.arrow")["rows"] == 5
    for row, bad in ((1, 7), (2, 1), (0, -1), (5, 1 << 20)):  # not monotonic (twice), negative, beyond the data
        b = bytearray(raw)
        b[at + 4 * row : at + 4 * row + 4] = int(bad).to_bytes(4, "little", signed=True)
        (tmp_path / "bad.arrow").write_bytes(bytes(b))
        with pytest.raises(capi.LdbError) as e:
            api.describe_ipc(tmp_path / "bad.arrow")
        assert e.value.status == capi.LDB_ERR_INVALID and "offset" in str(e.value), (row, bad, str(e.value))
    # zero-row batches between ordinary ones
    with pa.OSFile(str(tmp_path / "z.arrow"), "wb") as f:
        with pa.ipc.new_file(f, t.schema) as w:
            w.write_batch(t.to_batches()[0])
            w.write_batch(t.slice(0, 0).to_batches()[0] if t.slice(0, 0).to_batches() else pa.RecordBatch.from_arrays([pa.array([], pa.string()), pa.array([], pa.int32())], schema=t.schema))
            w.write_batch(t.to_batches()[0])
    d = api.describe_ipc(tmp_path / "z.arrow")
    assert d["batches"] == [5, 0, 5] and d["rows"] == 10
