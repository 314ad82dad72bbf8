"""The in-library RCCL exchange (ldb_gpu_comm_* / allgather / alltoall / shuffle).  With one GPU the
communicator has one rank — every transfer is a send to self inside the grouped batch, which exercises
the whole code path (metadata exchange, values, string lengths + bytes, validity) except the wire;
with two or more GPUs the sharded TPC-H plans run over it in separate processes (skipped here)."""
import os
import subprocess
import sys

import numpy as np
import pyarrow as pa
import pytest

from lingodb_amd import api, capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=["rccl", "shm"])
def comm(ctx, request):
    """one rank over either transport: RCCL (a send to self inside the grouped batch) or the host-staged one"""
    c = api.Comm(ctx, 0, 1, lambda ident: ident, transport=request.param)
    assert c.transport == request.param
    yield c
    c.close()


def sample():
    n = 1000
    rng = np.random.default_rng(7)
    return pa.table({
        "k": pa.array(rng.integers(0, 50, n).astype(np.int32)),
        "v": pa.array([None if i % 7 == 0 else int(x) for i, x in enumerate(rng.integers(-10**6, 10**6, n))], type=pa.int64()),
        "s": pa.array(["row%d" % i * (i % 3) for i in range(n)]),
        "d": pa.array([None if i % 11 == 0 else __import__("decimal").Decimal(int(x)).scaleb(-2) for i, x in enumerate(rng.integers(0, 10**9, n))], type=pa.decimal128(12, 2)),
    })


def test_allgather_one_rank_is_a_copy(ctx, comm):
    t = sample()
    dev = ctx.register("comm_t", t)
    got = comm.allgather(dev).to_arrow()
    assert got.to_pylist() == t.to_pylist()
    empty = ctx.register("comm_e", t.slice(0, 0))
    assert comm.allgather(empty).rows == 0


def test_alltoall_and_shuffle_one_rank(ctx, comm):
    t = sample()
    dev = ctx.register("comm_t2", t.select(["k", "v", "d"]))
    got = comm.alltoall(dev, [t.num_rows]).to_arrow()
    assert got.to_pylist() == t.select(["k", "v", "d"]).to_pylist()
    sh = comm.shuffle(dev.rel(), [(0, 0)], [(0, 0), (0, 1), (0, 2)]).to_arrow()
    assert sorted(sh.to_pylist(), key=repr) == sorted(t.select(["k", "v", "d"]).to_pylist(), key=repr)
    with pytest.raises(capi.LdbError):
        comm.alltoall(dev, [t.num_rows - 1])


def test_sharded_plans_over_rccl_two_gpus():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: the first multi-GPU box runs the sharded plans over the in-library RCCL exchange")
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, WORLD_SIZE="2", LDB_ID_FILE=os.path.join(tmp, "comm.id"), LDB_CHECK_TRANSPORT="rccl")
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_check.py")], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    assert outs[0].count(": OK") == 22 + 2, outs[0]
