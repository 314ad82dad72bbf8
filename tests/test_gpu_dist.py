"""N > 1 on the GPU box: all 22 sharded plans + the in-library exchange must reproduce the single-GPU results
bit-exactly.  The box has ONE GPU, so the ranks share it and the exchange runs over the library's host-staged
transport (comm_transport = 1); every other line of the exchange — metadata all-to-all, displacements, offset
and bitmap rebuild — is the code RCCL runs under on a multi-GPU node."""
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(world, extra_env=None, timeout=900):
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, WORLD_SIZE=str(world), LDB_ID_FILE=os.path.join(tmp, "comm.id"), LDB_CHECK_TRANSPORT="shm", **(extra_env or {}))
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_check.py")], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=timeout)[0])
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
        return [p.returncode for p in procs], outs


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_plans_match_single_gpu(world):
    codes, outs = run_ranks(world)
    assert all(c == 0 for c in codes), "\n".join(o[-3000:] for o in outs)
    assert outs[0].count(": OK") == 22 + 5, outs[0]
    assert "replayed executions under the communicator: OK" in outs[0] and "repeats the execution on every rank: OK" in outs[0]


def test_world8_all_sharded_plans_skew_and_stress():
    """8 ranks on the one GPU (BASELINE's largest world): all 22 sharded plans against the single-GPU plans, a shuffle whose rows
    almost all go to one rank, and 60 back-to-back exchanges of strings / NULLs / mixed widths of changing sizes"""
    codes, outs = run_ranks(8, {"LDB_CHECK_ORDERS": "90006", "LDB_CHECK_STRESS": "60", "LDB_CHECK_REPLAY": "1"}, timeout=1500)
    assert all(c == 0 for c in codes), "\n".join(o[-3000:] for o in outs)
    assert outs[0].count(": OK") == 22 + 6, outs[0]
    assert "exchange statistics" in outs[0]


def test_exchange_stress_loop_world3():
    """200 iterations at world 3 (the race fixed in 78e5108 showed once in 276 tests)"""
    codes, outs = run_ranks(3, {"LDB_CHECK_QUERIES": "6", "LDB_CHECK_STRESS": "200", "LDB_CHECK_SKEW": "0"})
    assert all(c == 0 for c in codes), "\n".join(o[-3000:] for o in outs)
    assert "stress loop x200: OK" in outs[0], outs[0]


def test_sharded_plans_with_narrow_decimals():
    """--narrow-decimals: 8-byte decimal columns next to the 16-byte aggregates group-by produces"""
    codes, outs = run_ranks(2, {"LDB_CHECK_NARROW": "1", "LDB_CHECK_QUERIES": "1,3,10,15,18,11"})
    assert all(c == 0 for c in codes), "\n".join(o[-3000:] for o in outs)
    assert outs[0].count(": OK") == 6 + 5, outs[0]


def test_sharded_plans_on_one_rank_equal_the_single_gpu_plans():
    """the sharded plan text without a communicator (allgather = copy, shuffle = materialize)"""
    sys.path.insert(0, ROOT)
    import lingodb_amd as ldb
    import tpch_plans
    from dist_gpu_check import rows_of, same_result

    ctx = ldb.Context(0)
    queries = list(range(1, 23))
    db = tpch_plans.Database(ctx, 30002, 0, 1, queries, False)
    a, b = tpch_plans.Runner(ctx, db, 1, None, None, force_dist=True), tpch_plans.Runner(ctx, db, 1, None, None)
    for q in queries:
        assert same_result(q, rows_of(a.run(q).to_arrow()), rows_of(b.run(q).to_arrow())), q
    ctx.close()
