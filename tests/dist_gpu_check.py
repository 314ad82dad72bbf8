"""Launched by torch.distributed.run (see test_gpu_dist.py): runs the sharded multi-rank TPC-H plans
(tpch_dist.py) and checks on rank 0 that they return exactly what the single-GPU plans return on
the unsharded database.  With LDB_DIST_BACKEND=gloo all ranks share GPU 0 (functional check of the
N>1 path on a 1-GPU box); with nccl it needs one GPU per rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lingo-db_amd"))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("LDB_DIST_BACKEND", "nccl")
    dev = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend)
    import lingodb_amd as ldb
    import tpch_plans

    n_orders = int(os.environ.get("LDB_CHECK_ORDERS", "150003"))  # enough orders for Q18's HAVING to keep some
    queries = [int(q) for q in os.environ.get("LDB_CHECK_QUERIES", "1,6,3,4,12,18,9,5,7,11,14,8").split(",")]
    ctx = ldb.Context(dev)
    db = tpch_plans.Database(ctx, n_orders, rank, world, queries, False)
    runner = tpch_plans.Runner(ctx, db, world, dist, torch)
    if backend == "nccl":  # the exchange inside the library (RCCL); torch.distributed only carries the communicator id
        from lingodb_amd import api

        def exchange_id(ident):
            box = [ident]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        runner.comm = api.Comm(ctx, rank, world, exchange_id)
    got = {q: runner.run(q).to_arrow() for q in queries}
    ok = True
    if rank == 0:
        full = tpch_plans.Database(ctx, n_orders, 0, 1, queries, False)
        single = tpch_plans.Runner(ctx, full, 1, None, torch)
        for q in queries:
            want = single.run(q).to_arrow()
            # positional rows: the interpreted single-GPU plans name their columns after the SQL text, the sharded plan pieces do not
            va = list(zip(*[c.to_pylist() for c in got[q].columns])) if got[q].num_columns else []
            vb = list(zip(*[c.to_pylist() for c in want.columns])) if want.num_columns else []
            a, b = va, vb
            if q == 3:  # ORDER BY revenue desc, o_orderdate: ties beyond the keys are unspecified
                same = [(r[1], r[2]) for r in va] == [(r[1], r[2]) for r in vb]
            elif q == 18:  # ORDER BY o_totalprice desc, o_orderdate
                same = [(r[4], r[3]) for r in va] == [(r[4], r[3]) for r in vb] and sorted(map(repr, va)) == sorted(map(repr, vb)) and len(va) > 0
            elif q == 10:  # ORDER BY revenue desc only
                same = [r[2] for r in va] == [r[2] for r in vb] and sorted(map(repr, va)) == sorted(map(repr, vb)) and len(va) == 20
            elif q == 11:  # ORDER BY value desc only
                same = [r[1] for r in va] == [r[1] for r in vb] and sorted(va) == sorted(vb) and len(va) > 0
            else:
                same = va == vb
            print(f"[dist-check] Q{q}: {'OK' if same else 'MISMATCH'} ({len(a)} rows, world={world}, backend={backend})", flush=True)
            if not same:
                print(a[:3], b[:3], flush=True)
            ok = ok and same
    # NULLs survive the exchange (a shard whose keyless partial SUM saw no row sends a NULL, not a 0)
    import pyarrow as pa
    import tpch_dist

    mine = pa.table({"v": pa.array([None] if rank == 0 else [10 * rank], pa.int64()), "w": pa.array([rank], pa.int32())})
    allv = tpch_dist.replicate(runner, ctx.register("nulls_%d" % rank, mine), "nulls_all").to_arrow()
    same = allv.column(0).to_pylist() == [None] + [10 * r for r in range(1, world)] and allv.column(1).to_pylist() == list(range(world))
    if rank == 0:
        print(f"[dist-check] NULL exchange: {'OK' if same else 'MISMATCH'} {allv.column(0).to_pylist()}", flush=True)
    ok = ok and same
    flag = torch.tensor([1 if ok else 0], device="cuda" if backend == "nccl" else "cpu")
    dist.broadcast(flag, 0)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) else 1)


if __name__ == "__main__":
    main()
