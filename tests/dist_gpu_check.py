"""One rank of the N > 1 check (started `world` times by test_gpu_dist.py, or by torch.distributed.run on a
multi-GPU node): runs the SHARDED TPC-H plans (lingo-db_amd/plans/tpch/dist/) through the library's own
exchange — ldb_gpu_allgather / ldb_gpu_shuffle behind the plan interpreter's allgather / shuffle steps — and
checks on rank 0 that they return exactly what the single-GPU plans return on the unsharded database.

Transport (LDB_CHECK_TRANSPORT): "shm" = host-staged, all ranks may share GPU 0 (the one-GPU box);
"rccl" = one GPU per rank.  The 128-byte communicator id travels through a file (LDB_ID_FILE); no torch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lingo-db_amd"))
sys.path.insert(0, ROOT)


def file_exchange(path, rank):
    def exchange(ident):
        if rank == 0:
            with open(path + ".tmp", "wb") as f:
                f.write(ident)
            os.replace(path + ".tmp", path)
            return ident
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 120:
                raise TimeoutError("communicator id file never appeared")
            time.sleep(0.01)
        with open(path, "rb") as f:
            return f.read()

    return exchange


def rows_of(t):
    return list(zip(*[c.to_pylist() for c in t.columns])) if t.num_columns else []


def same_result(q, va, vb):
    """ORDER BY keys compared in order, ties beyond them as sets (the SQL leaves their order open)"""
    order = {3: lambda r: (r[1], r[2]), 18: lambda r: (r[4], r[3]), 10: lambda r: r[2], 11: lambda r: r[1], 2: lambda r: (r[0], r[2], r[1], r[3]),
             21: lambda r: (r[1], r[0]), 13: lambda r: (r[1], r[0]), 16: lambda r: (r[3], r[0], r[1], r[2])}
    if q in order:
        return [order[q](r) for r in va] == [order[q](r) for r in vb] and sorted(map(repr, va)) == sorted(map(repr, vb))
    return va == vb


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    transport = os.environ.get("LDB_CHECK_TRANSPORT", "shm")
    import lingodb_amd as ldb
    from lingodb_amd import api, capi
    import tpch_plans

    n_dev = capi.gpu_lib().ldb_gpu_device_count() if hasattr(capi.gpu_lib(), "ldb_gpu_device_count") else 1
    dev = (int(os.environ.get("LOCAL_RANK", rank)) % max(n_dev, 1)) if transport == "shm" else int(os.environ.get("LOCAL_RANK", rank))
    n_orders = int(os.environ.get("LDB_CHECK_ORDERS", "150003"))  # enough orders for Q18's HAVING to keep some
    queries = [int(q) for q in os.environ.get("LDB_CHECK_QUERIES", ",".join(str(q) for q in range(1, 23))).split(",")]
    ctx = ldb.Context(dev)
    capi.gpu_lib().ldb_gpu_set_option(b"comm_timeout_ms", 60000)
    comm = api.Comm(ctx, rank, world, file_exchange(os.environ["LDB_ID_FILE"], rank), transport=transport)
    db = tpch_plans.Database(ctx, n_orders, rank, world, queries, bool(int(os.environ.get("LDB_CHECK_NARROW", "0"))))
    runner = tpch_plans.Runner(ctx, db, world, None, None, comm=comm)
    ok = True
    got = {}
    for q in queries:
        got[q] = runner.run(q).to_arrow()
    # the sharded plans again: prepared plans REPLAY their read-back trace under the communicator (all ranks together: ldb_gpu_comm_agree before
    # and after every execution) — same rows as the recording run, and every rank did replay
    n_again = int(os.environ.get("LDB_CHECK_REPLAY", "2"))
    replay_ok = True
    for _ in range(n_again):
        for q in queries:
            again = runner.run(q).to_arrow()
            replay_ok = replay_ok and same_result(q, rows_of(again), rows_of(got[q]))
    if n_again:
        stats = runner.prepared_stats()
        # (a miss = an execution every rank repeats: a count recorded on another path of an operator — caches that fill during the first run;
        # none is expected any more, a few would still be correct)
        replay_ok = replay_ok and stats["replays"] >= len(queries) * (n_again - 1) and stats["misses"] <= max(2, len(queries) // 4)
        if rank == 0:
            print(f"[dist-check] replayed executions under the communicator: {'OK' if replay_ok else 'MISMATCH'} "
                  f"(executions {stats['executions']}, replays {stats['replays']}, misses {stats['misses']})", flush=True)
        ok = ok and replay_ok
    if rank == 0:
        full = tpch_plans.Database(ctx, n_orders, 0, 1, queries, bool(int(os.environ.get("LDB_CHECK_NARROW", "0"))))
        single = tpch_plans.Runner(ctx, full, 1, None, None)
        for q in queries:
            want = single.run(q).to_arrow()
            va, vb = rows_of(got[q]), rows_of(want)
            same = same_result(q, va, vb) and (len(vb) > 0 or q in (15,))
            print(f"[dist-check] Q{q}: {'OK' if same else 'MISMATCH'} ({len(va)} rows, world={world}, transport={comm.transport})", flush=True)
            if not same:
                print("   sharded:", va[:3], "\n   single: ", vb[:3], flush=True)
            ok = ok and same
    # ---- a forced divergence on ONE rank: its table is overwritten through raw device pointers (the one change a trace key cannot see), so its
    # replayed counts are wrong; it runs on over the recorded transfer sizes, reports the miss, and EVERY rank repeats the execution recording
    if n_again:
        import json

        import numpy as np
        import pyarrow as pa

        def piece(r, changed):
            g = np.random.default_rng(500 + r)
            x = g.integers(0, 1000, 60_000).astype(np.int32)
            if changed:
                x = np.where(g.random(len(x)) < 0.5, x, 0).astype(np.int32)  # more rows pass the filter than were recorded
            return x

        def expect(changed_rank):
            n = s_ = 0
            for r in range(world):
                x = piece(r, r == changed_rank)
                sel = np.nonzero(x < 500)[0]
                n += len(sel)
                s_ += int((sel + r * 1_000_000).sum())
            return n, s_

        x0 = piece(rank, False)
        tt = ctx.register("div_%d" % rank, pa.table({"x": pa.array(x0, pa.int32()), "i": pa.array(np.arange(len(x0), dtype=np.int64) + rank * 1_000_000)}))
        plan = ctx.prepare_plan(json.dumps({"name": "dist_replay", "inputs": ["t"], "steps": [
            {"op": "scan", "table": "t", "out": "s"}, {"op": "filter", "in": "s", "preds": [{"col": "x", "op": "LT", "value": 500}], "out": "f"},
            {"op": "shuffle", "in": "f", "keys": ["i"], "cols": ["i", "x"], "out": "sh"},
            {"op": "groupby", "in": "sh", "aggs": [{"fn": "count_star", "as": "n"}, {"fn": "sum", "expr": "i", "as": "s"}], "out": "part"},
            {"op": "allgather", "in": "part", "out": "result"}], "result": "result"}))

        def total():
            res = plan.execute({"t": tt}, comm=comm).to_arrow()
            return sum(res.column(0).to_pylist()), sum(v for v in res.column(1).to_pylist() if v is not None)

        # (every rank makes every collective call whatever it has seen so far: no short-circuit in front of total())
        first = [total() for _ in range(3)]
        div_ok = all(v == expect(-1) for v in first) and plan.stats()["replays"] >= 1 and plan.stats()["misses"] == 0
        victim = world - 1
        if rank == victim:
            y = piece(rank, True)
            other = ctx.register("div_other_%d" % rank, pa.table({"x": pa.array(y, pa.int32())}))
            dst, _, _, nbytes = tt.col_ptrs(0)
            src, _, _, _ = other.col_ptrs(0)
            api.check(ctx.lib.ldb_gpu_memcpy_d2d(ctx.h, dst, src, nbytes))
            ctx.sync()
        after = total()
        st_div = plan.stats()
        again = total()
        # the victim's own miss — on the other ranks the repeat a peer's miss forces — and exactly one of them; the repeated execution and the
        # next (replaying the new record) return the new answer
        div_ok = div_ok and after == expect(victim) and again == expect(victim) and st_div["misses"] == 1 and plan.stats()["misses"] == 1
        # a divergence that also leaves the recorded SEQUENCE of read-backs: no row of the victim passes the filter any more, so its
        # operators take their empty-input paths; it must still meet every exchange of the plan (a rank that stopped at the first unexpected
        # read-back would leave its peers waiting inside the shuffle), and every rank repeats
        if rank == victim:
            none = ctx.register("div_none_%d" % rank, pa.table({"x": pa.array(np.full(len(x0), 999, dtype=np.int32), pa.int32())}))
            dst, _, _, nbytes = tt.col_ptrs(0)
            src, _, _, _ = none.col_ptrs(0)
            api.check(ctx.lib.ldb_gpu_memcpy_d2d(ctx.h, dst, src, nbytes))
            ctx.sync()

        def expect_empty_victim():
            n = s_ = 0
            for r in range(world):
                if r == victim:
                    continue
                sel = np.nonzero(piece(r, False) < 500)[0]
                n += len(sel)
                s_ += int((sel + r * 1_000_000).sum())
            return n, s_

        empty1 = total()
        st_empty = plan.stats()
        empty2 = total()
        div_ok = div_ok and empty1 == expect_empty_victim() and empty2 == expect_empty_victim() and st_empty["misses"] == 2 and plan.stats()["misses"] == 2
        flags = ctx.register("divok_%d" % rank, pa.table({"ok": pa.array([1 if div_ok else 0], pa.int32())}))
        all_ok = comm.allgather(flags, "divok_all").to_arrow().column(0).to_pylist()
        if rank == 0:
            print(f"[dist-check] forced divergence on rank {victim} repeats the execution on every rank: {'OK' if all(all_ok) else 'MISMATCH'} {all_ok} {st_div}", flush=True)
        ok = ok and all(all_ok)
        plan.release()
    # the exchange itself on ragged inputs: strings, NULLs, a narrow (8-byte) decimal key next to a 16-byte aggregate, empty ranks
    import decimal

    import pyarrow as pa

    n_mine = 0 if rank == 1 else 3 + rank
    mine = pa.table({"k": pa.array([rank * 10 + i for i in range(n_mine)], pa.int32()),
                     "s": pa.array([None if i == 1 else "r%d-%s" % (rank, "x" * (i * 5)) for i in range(n_mine)], pa.string()),
                     "d": pa.array([decimal.Decimal(rank * 100 + i) / 100 for i in range(n_mine)], pa.decimal128(12, 2)),
                     "v": pa.array([None if (rank + i) % 3 == 0 else rank * 1000 + i for i in range(n_mine)], pa.int64())})
    t = ctx.register("x_%d" % rank, mine, True)  # narrow: d is 8 bytes wide on the device
    mixed = ctx.run_plan('{"steps": [{"op": "groupby", "in": "t", "keys": ["k"], "aggs": [{"fn": "sum", "expr": "d", "as": "dsum"}], "est_groups": 8, "key_names": ["gk"], "out": "g"},'
                         ' {"op": "join_build", "in": "g", "keys": ["gk"], "unique": true, "out": "h"}, {"op": "join_probe", "ht": "h", "in": "t", "keys": ["k"], "out": "j"},'
                         ' {"op": "materialize", "in": "j", "cols": ["k", "d", "s", "v", "dsum"], "out": "result"}], "result": "result"}', {"t": t})  # d: 8 bytes, dsum: 16 bytes
    assert mixed.col_width(1) == 8 and mixed.col_width(4) == 16, (mixed.col_width(1), mixed.col_width(4))
    allv = comm.allgather(mixed, "mixed_all").to_arrow()
    want_rows = []
    for r in range(world):
        nr = 0 if r == 1 else 3 + r
        for i in range(nr):
            d = decimal.Decimal(r * 100 + i) / 100
            want_rows.append((r * 10 + i, d, None if i == 1 else "r%d-%s" % (r, "x" * (i * 5)), None if (r + i) % 3 == 0 else r * 1000 + i, d))
    same = rows_of(allv) == want_rows
    if rank == 0:
        print(f"[dist-check] mixed-width / string / NULL all-gather: {'OK' if same else 'MISMATCH'}", flush=True)
        if not same:
            print(rows_of(allv)[:4], want_rows[:4], flush=True)
    ok = ok and same
    # hash shuffle: afterwards equal keys are on one rank and nothing is lost
    sh = comm.shuffle(mixed.rel(), [(0, 0)], [(0, 0), (0, 2)], "shuffled").to_arrow()
    keys_here = sh.column(0).to_pylist()
    cnt = ctx.register("cnt_%d" % rank, pa.table({"n": pa.array([len(keys_here)], pa.int64()), "k": pa.array([",".join(map(str, sorted(keys_here)))], pa.string())}))
    allc = comm.allgather(cnt, "cnt_all").to_arrow()
    sets = [set(map(int, s.split(","))) if s else set() for s in allc.column(1).to_pylist()]
    same = sum(allc.column(0).to_pylist()) == len(want_rows) and all(not (sets[a] & sets[b]) for a in range(world) for b in range(a + 1, world))
    if rank == 0:
        print(f"[dist-check] hash shuffle: {'OK' if same else 'MISMATCH'} {allc.column(0).to_pylist()}", flush=True)
    ok = ok and same
    # ---- skew: most rows carry ONE key, so one rank receives almost everything (receive buffers sized from the peers' counts)
    if os.environ.get("LDB_CHECK_SKEW", "1") != "0":
        import numpy as np

        n = 40_000
        rng = np.random.default_rng(100 + rank)
        k = np.where(rng.random(n) < 0.85, 7, rng.integers(0, 5000, n)).astype(np.int32)
        v = rng.integers(0, 1000, n).astype(np.int64)
        tk = ctx.register("skew_%d" % rank, pa.table({"k": pa.array(k, pa.int32()), "v": pa.array(v, pa.int64()), "s": pa.array(["k%d" % x for x in k], pa.string())}))
        shk = comm.shuffle(tk.rel(), [(0, 0)], [(0, 0), (0, 1), (0, 2)], "skewed")
        part = ctx.run_plan('{"steps": [{"op": "groupby", "in": "t", "keys": ["k"], "aggs": [{"fn": "count_star", "as": "n"}, {"fn": "sum", "expr": "v", "as": "sv"}], "est_groups": 6000, "out": "result"}], "result": "result"}',
                            {"t": shk})
        allp = comm.allgather(part, "skew_groups").to_arrow()
        got_groups = sorted(zip(allp.column(0).to_pylist(), allp.column(1).to_pylist(), allp.column(2).to_pylist()))
        want = {}
        for r in range(world):
            rr = np.random.default_rng(100 + r)
            kk = np.where(rr.random(n) < 0.85, 7, rr.integers(0, 5000, n)).astype(np.int32)
            vv = rr.integers(0, 1000, n).astype(np.int64)
            for key, val in zip(kk.tolist(), vv.tolist()):
                c = want.setdefault(key, [0, 0])
                c[0] += 1
                c[1] += val
        same = got_groups == sorted((key, c[0], c[1]) for key, c in want.items())  # every key on exactly one rank, nothing lost, nothing twice
        if rank == 0:
            print(f"[dist-check] skewed shuffle (85 % of the rows on one key): {'OK' if same else 'MISMATCH'} ({shk.rows} rows arrived on rank 0)", flush=True)
        ok = ok and same
    # ---- stress: many exchanges back to back with strings, NULLs and mixed widths of changing sizes (pins the staging-copy race
    # of round 3: a transfer still in flight when the read-back of the received metadata ran)
    iters = int(os.environ.get("LDB_CHECK_STRESS", "0"))
    bad = 0
    for it in range(iters):
        import numpy as np

        def piece(r):
            g = np.random.default_rng(1000 * it + r)
            m = int(g.integers(0, 40)) if (it + r) % 5 else 0
            ks = g.integers(0, 50, m).astype(np.int32)
            return pa.table({"k": pa.array(ks, pa.int32()), "s": pa.array([None if x % 7 == 0 else "s%d-%s" % (x, "y" * int(x % 13)) for x in ks.tolist()], pa.string()),
                             "d": pa.array([decimal.Decimal(int(x)) / 100 for x in ks.tolist()], pa.decimal128(12, 2)), "w": pa.array([None if x % 3 == 0 else int(x) << 33 for x in ks.tolist()], pa.int64())})

        t_it = ctx.register("st_%d_%d" % (it, rank), piece(rank), bool(it % 2))
        ag = rows_of(comm.allgather(t_it, "st_all").to_arrow())
        exp = [row for r in range(world) for row in rows_of(piece(r))]
        if ag != exp:
            bad += 1
        sh2 = comm.shuffle(t_it.rel(), [(0, 0)], [(0, 0), (0, 1), (0, 3)], "st_sh").to_arrow()
        cnt2 = ctx.register("stc_%d_%d" % (it, rank), pa.table({"n": pa.array([sh2.num_rows], pa.int64())}))
        tot = sum(comm.allgather(cnt2, "stc_all").to_arrow().column(0).to_pylist())
        if tot != len(exp):
            bad += 1
    if iters and rank == 0:
        print(f"[dist-check] stress loop x{iters}: {'OK' if bad == 0 else 'MISMATCH (%d)' % bad}", flush=True)
    ok = ok and bad == 0
    st = comm.stats()
    if rank == 0:
        print(f"[dist-check] exchange statistics: {st}", flush=True)
    ok = ok and st["groups"] > 0 and st["bytes_out"] > 0 and st["max_peer_bytes_out"] <= st["bytes_out"]
    comm.close()
    ctx.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
