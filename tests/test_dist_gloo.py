"""Multi-rank exchange on CPU: world_size 2 and 3, rendezvous over torch.distributed's gloo backend (what
bench.py uses to hand out the communicator id), transfers through the LIBRARY's own host-staged transport
(csrc/ldb_comm.hip: ShmTransport in host mode, ldb_gpu_comm_create_host + ldb_gpu_comm_alltoall_bytes).
What runs here without a GPU is the transport protocol the -m gpu tests (test_gpu_dist.py) run the table
exchange over: per-pair segments, in-order matching of a pair's transfers, the two barriers of a round,
bounded waits.  The packed send buffers of the shuffle test are built with the ORACLE's partition ids — the
layout ldb_gpu_partition produces on the device (test_gpu_parity.py::test_partition_matches_reference_hash_radix)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pyarrow as pa
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class HostComm:
    """ldb_gpu_comm_create_host over an id broadcast through gloo"""

    def __init__(self, rank, world):
        from lingodb_amd import capi

        self.lib, self.rank, self.world = capi.gpu_lib(), rank, world
        buf = C.create_string_buffer(128)
        if rank == 0:
            self.lib.ldb_gpu_set_option(b"comm_transport", 1)
            capi.check(self.lib.ldb_gpu_comm_unique_id(buf))
        box = [buf.raw if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        self.h = C.c_void_p()
        capi.check(self.lib.ldb_gpu_comm_create_host(rank, world, C.create_string_buffer(box[0], 128), C.byref(self.h)))
        assert self.lib.ldb_gpu_comm_transport(self.h) == b"shm"

    def alltoall(self, send_chunks, recv_sizes):
        """send_chunks[p]: bytes for peer p; returns the list of received byte strings (by peer)"""
        from lingodb_amd import capi

        send = b"".join(send_chunks)
        sb = (C.c_int64 * self.world)(*[len(c) for c in send_chunks])
        rb = (C.c_int64 * self.world)(*recv_sizes)
        recv = C.create_string_buffer(max(sum(recv_sizes), 1))
        capi.check(self.lib.ldb_gpu_comm_alltoall_bytes(self.h, send, sb, recv, rb))
        out, at = [], 0
        for n in recv_sizes:
            out.append(recv.raw[at:at + n])
            at += n
        return out

    def close(self):
        self.lib.ldb_gpu_comm_destroy(self.h)


def _worker(rank, world, port, fn_name, ret):
    sys.path.insert(0, os.path.join(ROOT, "lingo-db_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = HostComm(rank, world)
        ret[rank] = globals()[fn_name](comm, rank, world)
        comm.close()
    finally:
        dist.destroy_process_group()


def _run(world, fn_name):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn_name, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _payload(src, dst, n):
    return bytes((src * 31 + dst * 7 + i) & 0xFF for i in range(n))


def _size(s, d, rnd):
    return 0 if (s + d + rnd) % 3 == 0 else 1 + 1000 * ((s * 3 + d + rnd) % 5)


# ---------------------------------------------------------------- ragged all-to-all, several rounds, empty pairs
def _do_rounds(comm, rank, world):
    got = []
    for rnd in range(4):  # the segment names carry a round number: rounds must not see each other's data
        recv = comm.alltoall([_payload(rank, p, _size(rank, p, rnd)) for p in range(world)], [_size(p, rank, rnd) for p in range(world)])
        got.append(all(recv[p] == _payload(p, rank, _size(p, rank, rnd)) for p in range(world)))
    return got


@pytest.mark.parametrize("world", [2, 3])
def test_ragged_alltoall_rounds(world):
    assert _run(world, "_do_rounds") == [[True] * 4] * world


# ---------------------------------------------------------------- the verdict of a sharded prepared plan: minimum over the ranks
def _do_agree(comm, rank, world):
    """ldb_gpu_comm_agree (host communicator, ctx = NULL): what ldb_plan_execute asks before a collective replay ("can everybody replay?")
    and after it ("did everybody's replay hold?") — the minimum of the ranks' flags, the same answer on every rank, round after round"""
    from lingodb_amd import capi

    out = []
    for rnd, mine in enumerate([1, 1 if rank != world - 1 else 0, 1, 0 if rank == 0 else 1, 1]):
        got = C.c_int32(-7)
        capi.check(comm.lib.ldb_gpu_comm_agree(None, comm.h, mine, C.byref(got)))
        out.append(got.value)
    return out


@pytest.mark.parametrize("world", [2, 3])
def test_agree_is_the_minimum_over_the_ranks(world):
    assert _run(world, "_do_agree") == [[1, 0, 1, 0, 1]] * world


# ---------------------------------------------------------------- replicate (all-gather = everybody sends everything to everybody)
def _rank_rows(rank, n):
    rng = np.random.default_rng(100 + rank)
    return rng.integers(0, 1000, n).astype(np.int32), rng.integers(-(2 ** 40), 2 ** 40, n).astype(np.int64)


def _do_allgather(comm, rank, world):
    ns = [5 + 3 * r if r != 1 else 0 for r in range(world)]  # ragged, rank 1 contributes nothing
    keys, vals = _rank_rows(rank, ns[rank])
    mine = keys.tobytes() + vals.tobytes()
    recv = comm.alltoall([mine] * world, [12 * n for n in ns])
    k = np.concatenate([np.frombuffer(recv[p][:4 * ns[p]], np.int32) for p in range(world)]).tolist()
    v = np.concatenate([np.frombuffer(recv[p][4 * ns[p]:], np.int64) for p in range(world)]).tolist()
    return k, v


@pytest.mark.parametrize("world", [2, 3])
def test_allgather_with_an_empty_rank(world):
    ns = [5 + 3 * r if r != 1 else 0 for r in range(world)]
    want_k = np.concatenate([_rank_rows(r, ns[r])[0] for r in range(world)]).tolist()
    want_v = np.concatenate([_rank_rows(r, ns[r])[1] for r in range(world)]).tolist()
    for k, v in _run(world, "_do_allgather"):
        assert k == want_k and v == want_v


# ---------------------------------------------------------------- re-partition (hash-radix shuffle of (key, value) rows)
def _do_shuffle(comm, rank, world):
    import oracle_bind

    oracle = oracle_bind.load()
    n = 2000 + 500 * rank
    keys, vals = _rank_rows(rank, n)
    rel = oracle_bind.HostTable(pa.table({"k": pa.array(keys), "v": pa.array(vals)})).rel()
    ids = oracle.partition_ids(rel, [(0, 0)], world)  # (db.hash(key) >> 16) % world
    order = np.argsort(ids, kind="stable")  # rows grouped by destination, stable (= ldb_gpu_partition)
    send_counts = np.bincount(ids, minlength=world)
    # the row counts travel first (the metadata all-to-all of the table exchange), then the rows
    cnt = comm.alltoall([np.int64(c).tobytes() for c in send_counts], [8] * world)
    recv_counts = [int(np.frombuffer(c, np.int64)[0]) for c in cnt]
    off = np.concatenate([[0], np.cumsum(send_counts)])
    ks, vs = keys[order], vals[order]
    recv = comm.alltoall([ks[off[p]:off[p + 1]].tobytes() + vs[off[p]:off[p + 1]].tobytes() for p in range(world)], [12 * c for c in recv_counts])
    k = np.concatenate([np.frombuffer(recv[p][:4 * recv_counts[p]], np.int32) for p in range(world)]).tolist()
    v = np.concatenate([np.frombuffer(recv[p][4 * recv_counts[p]:], np.int64) for p in range(world)]).tolist()
    return k, v, recv_counts


@pytest.mark.parametrize("world", [2, 3])
def test_shuffle_by_reference_hash(world, oracle):
    import oracle_bind

    res = _run(world, "_do_shuffle")
    per_src = []
    for r in range(world):
        keys, vals = _rank_rows(r, 2000 + 500 * r)
        rel = oracle_bind.HostTable(pa.table({"k": pa.array(keys), "v": pa.array(vals)})).rel()
        per_src.append((keys, vals, oracle.partition_ids(rel, [(0, 0)], world)))
    total = 0
    for d, (k, v, recv_counts) in enumerate(res):  # rank d receives, in source-rank order, the rows whose key hashes to d
        assert k == np.concatenate([keys[ids == d] for keys, vals, ids in per_src]).tolist()
        assert v == np.concatenate([vals[ids == d] for keys, vals, ids in per_src]).tolist()
        assert recv_counts == [int((ids == d).sum()) for _, _, ids in per_src]
        total += len(k)
    assert total == sum(2000 + 500 * r for r in range(world))  # nothing lost, nothing duplicated


# ---------------------------------------------------------------- a missing peer is an error, not a hang
def _do_timeout(comm, rank, world):
    from lingodb_amd import capi

    comm.lib.ldb_gpu_set_option(b"comm_timeout_ms", 300)
    if rank == 1:
        import time

        time.sleep(1.5)  # never enters the exchange; shows up (to close) only after the peer gave up
        return "absent"
    try:
        comm.alltoall([b"x"] * world, [1] * world)
    except capi.LdbError as e:
        return "timed out" if "timed out" in str(e) or "peer rank failed" in str(e) else str(e)
    return "no error"


def test_missing_peer_times_out():
    res = _run(2, "_do_timeout")
    assert res == ["timed out", "absent"]
