"""Multi-rank exchange logic on CPU: world_size 2 and 3 over the gloo backend (the N>1 path of
bench.py uses the same functions over RCCL).  The packed send buffers are built with the ORACLE's
partition ids — the layout ldb_gpu_partition produces on the device (checked against the oracle in
test_gpu_parity.py::test_partition_matches_reference_hash_radix)."""
import os
import socket
import sys

import numpy as np
import pyarrow as pa
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn_name, ret):
    sys.path.insert(0, os.path.join(ROOT, "lingo-db_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = globals()[fn_name](rank, world)
    finally:
        dist.destroy_process_group()


def _run(world, fn_name):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn_name, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _rank_rows(rank, n):
    rng = np.random.default_rng(100 + rank)
    keys = rng.integers(0, 1000, n).astype(np.int32)
    vals = rng.integers(-(2 ** 40), 2 ** 40, n).astype(np.int64)
    return keys, vals


# ---------------------------------------------------------------- replicate (all-gather)
def _do_allgather(rank, world):
    from lingodb_amd import dist as ldist

    n = 5 + 3 * rank  # ragged: every rank contributes a different number of rows
    keys, vals = _rank_rows(rank, n)
    cols = [torch.from_numpy(keys.view(np.uint8).copy()), torch.from_numpy(vals.view(np.uint8).copy())]
    out, counts = ldist.allgather_columns(dist, cols, [4, 8], n)
    return out[0].numpy().view(np.int32).tolist(), out[1].numpy().view(np.int64).tolist(), counts


@pytest.mark.parametrize("world", [2, 3])
def test_allgather_columns(world):
    res = _run(world, "_do_allgather")
    want_k = np.concatenate([_rank_rows(r, 5 + 3 * r)[0] for r in range(world)]).tolist()
    want_v = np.concatenate([_rank_rows(r, 5 + 3 * r)[1] for r in range(world)]).tolist()
    for k, v, counts in res:
        assert k == want_k and v == want_v and counts == [5 + 3 * r for r in range(world)]


def _do_allgather_ragged_widths(rank, world):
    """three columns of 4 / 16 / 1 bytes per row in ONE packed message; rank 1 contributes nothing"""
    from lingodb_amd import dist as ldist

    n = 0 if rank == 1 else 3 + rank
    a = (np.arange(n) + 10 * rank).astype(np.int32)
    b = (np.arange(2 * n) + 1000 * rank).astype(np.int64)  # 16 bytes per row
    c = (np.arange(n) + rank).astype(np.uint8)
    cols = [torch.from_numpy(x.view(np.uint8).copy()) if n else torch.zeros(1, dtype=torch.uint8) for x in (a, b, c)]
    out, counts = ldist.allgather_columns(dist, cols, [4, 16, 1], n)
    return out[0].numpy().view(np.int32).tolist(), out[1].numpy().view(np.int64).tolist(), out[2].numpy().tolist(), counts


def test_allgather_packed_columns_with_an_empty_rank():
    res = _run(3, "_do_allgather_ragged_widths")
    ns = [3, 0, 5]
    want_a = sum([(np.arange(n) + 10 * r).tolist() for r, n in enumerate(ns)], [])
    want_b = sum([(np.arange(2 * n) + 1000 * r).tolist() for r, n in enumerate(ns)], [])
    want_c = sum([(np.arange(n) + r).tolist() for r, n in enumerate(ns)], [])
    for a, b, c, counts in res:
        assert (a, b, c, counts) == (want_a, want_b, want_c, ns)


# ---------------------------------------------------------------- re-partition (all-to-all shuffle)
def _do_alltoall(rank, world):
    import oracle_bind
    from lingodb_amd import dist as ldist

    oracle = oracle_bind.load()
    n = 2000 + 500 * rank
    keys, vals = _rank_rows(rank, n)
    rel = oracle_bind.HostTable(pa.table({"k": pa.array(keys), "v": pa.array(vals)})).rel()
    ids = oracle.partition_ids(rel, [(0, 0)], world)  # (db.hash(key) >> 16) % world
    order = np.argsort(ids, kind="stable")  # rows grouped by destination, stable (= ldb_gpu_partition)
    send_counts = np.bincount(ids, minlength=world).tolist()
    cols = [torch.from_numpy(keys[order].view(np.uint8).copy()), torch.from_numpy(vals[order].view(np.uint8).copy())]
    out, recv_counts = ldist.alltoall_columns(dist, cols, [4, 8], send_counts)
    return out[0].numpy().view(np.int32).tolist(), out[1].numpy().view(np.int64).tolist(), recv_counts


@pytest.mark.parametrize("world", [2, 3])
def test_alltoall_shuffle_by_reference_hash(world, oracle):
    import oracle_bind

    res = _run(world, "_do_alltoall")
    # expectation computed globally: rank d receives, in source-rank order, the rows whose key hashes to d
    per_src = []
    for r in range(world):
        keys, vals = _rank_rows(r, 2000 + 500 * r)
        rel = oracle_bind.HostTable(pa.table({"k": pa.array(keys), "v": pa.array(vals)})).rel()
        per_src.append((keys, vals, oracle.partition_ids(rel, [(0, 0)], world)))
    total = 0
    for d, (k, v, recv_counts) in enumerate(res):
        wk = np.concatenate([keys[ids == d] for keys, vals, ids in per_src]).tolist()
        wv = np.concatenate([vals[ids == d] for keys, vals, ids in per_src]).tolist()
        assert k == wk and v == wv
        assert recv_counts == [int((ids == d).sum()) for _, _, ids in per_src]
        total += len(k)
    assert total == sum(2000 + 500 * r for r in range(world))  # nothing lost, nothing duplicated


def test_alltoall_with_empty_partitions():
    res = _run(2, "_do_alltoall_empty")
    assert res[0] == ([], [0, 0]) and res[1] == (list(range(7)) + list(range(100, 104)), [7, 4])


def _do_alltoall_empty(rank, world):
    from lingodb_amd import dist as ldist

    n = 7 if rank == 0 else 4
    vals = (np.arange(n) + 100 * rank).astype(np.int64)
    cols = [torch.from_numpy(vals.view(np.uint8).copy())]
    out, recv = ldist.alltoall_columns(dist, cols, [8], [0, n])  # everything goes to rank 1
    return out[0].numpy().view(np.int64).tolist(), recv


def test_shard_bounds_tile_exactly():
    from lingodb_amd import dist as ldist

    for n, w in [(0, 2), (1, 3), (10, 4), (1500000, 8)]:
        b = ldist.shard_bounds(n, w)
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
