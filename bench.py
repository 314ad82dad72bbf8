#!/usr/bin/env python3
"""bench.py — TPC-H (synthetic, dbgen-shaped) on the MI355X-native LingoDB operator runtime.

  python bench.py --gpus N --steps K --warmup W [--sf 100] [--queries 1,3,…]

One "step" = one pass of all 22 TPC-H queries over the HBM-resident database, each query ending with its
result rows handed to the host.  N > 1: launched by torch.distributed.run, one rank per GPU; the database is
sharded (orders / lineitem by order ranges, the other tables by rows; strong scaling: the total is SF `--sf`)
and every rank runs the sharded plans (lingo-db_amd/plans/tpch/dist/*.json) whose allgather / shuffle steps go
through the library's exchange — RCCL over xGMI; torch.distributed only hands out the communicator id.
Prints ONE JSON line on rank 0 (DESIGN.md §4 explains every field).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "lingo-db_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ≈6.3 TB/s achievable
ORDERS_PER_SF = 1_500_000

# algorithmic bytes per input row at the resident (Arrow-native) widths, SURVEY §8(d)
Q1_BYTES_PER_ROW = 4 + 4 + 4 + 16 + 16 + 16 + 16  # shipdate, returnflag, linestatus, qty, extprice, discount, tax
Q1_BYTES_PER_ROW_NARROW = 4 + 4 + 4 + 8 + 8 + 8 + 8


def geomean(xs):
    xs = [max(x, 1e-9) for x in xs]
    return math.exp(sum(math.log(x) for x in xs) / len(xs))


def make_comm(ctx, rank, world, dist, torch, backend):
    """The library's communicator for world > 1.  RCCL when every rank can initialise it; otherwise the host-staged
    transport (ranks of one node) — every rank takes the same path (agreed by an all-reduce BEFORE any rank waits for
    an id, so a rank that cannot load librccl cannot leave the others blocked in a broadcast)."""
    from lingodb_amd import api, capi

    red_dev = "cpu" if backend == "gloo" else "cuda"

    def all_ranks(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def exchange_id(ident):
        box = [ident]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def selftest(comm):  # one tiny all-gather with a string column and a NULL before any query depends on it
        import pyarrow as pa

        mine = ctx.register("comm_selftest", pa.table({"r": pa.array([rank, rank], pa.int32()), "s": pa.array(["x" * rank, None], pa.string())}))
        got = comm.allgather(mine, "comm_selftest_all").to_arrow()
        return (got.column(0).to_pylist() == [r for r in range(world) for _ in (0, 1)] and got.column(1).to_pylist() == [v for r in range(world) for v in ("x" * r, None)])

    want = os.environ.get("LDB_COMM", "rccl" if backend == "nccl" else "shm")
    for transport in ([want] if want == "shm" else ["rccl", "shm"]):
        comm, ok = None, True
        if transport == "rccl":
            ok = all_ranks(capi.gpu_lib().ldb_gpu_comm_available() == 1)  # librccl loadable on EVERY rank, checked before the id broadcast
        if ok:
            try:
                comm = api.Comm(ctx, rank, world, exchange_id, transport=transport)
            except Exception as e:
                print(f"[bench] rank {rank}: {transport} communicator failed: {e}", file=sys.stderr, flush=True)
                ok = False
            ok = all_ranks(ok)
        if ok:
            try:
                ok = selftest(comm)
            except Exception as e:
                print(f"[bench] rank {rank}: {transport} all-gather self-test failed: {e}", file=sys.stderr, flush=True)
                ok = False
            ok = all_ranks(ok)
        if ok:
            return comm, ("librccl inside liblingodb_gpu.so (grouped send/recv on the ctx stream)" if transport == "rccl"
                          else "host-staged shared-memory transport inside liblingodb_gpu.so (RCCL not usable on every rank)" if want != "shm"
                          else "host-staged shared-memory transport inside liblingodb_gpu.so (LDB_COMM=shm)")
        if comm is not None:
            comm.close()
    raise SystemExit("bench.py: no exchange transport works on every rank")


def dry_run(args):
    """What `--gpus N --sf S --queries …` needs per rank, computed on the host (the generator is a pure function of (table, row)):
    rows of every table fragment (ldb_tpch_host_rows), bytes of the resident columns (fixed widths; utf8 from a 20 000-order sample
    of the host generator), the largest row-id space, and — for plans with a shuffle — the bytes a rank sends if its rows spread evenly.
    Budgets: 288 GB of HBM per MI355X (80 % usable by tables + intermediates), uint32 row ids (4 294 967 294 rows per fragment)."""
    import ctypes as C

    import tpch_plans
    from lingodb_amd import capi

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tpch_data

    lib = capi.host_lib()
    world = args.gpus
    queries = [int(q) for q in args.queries.split(",") if q] or list(range(1, 23))
    n_orders = int(round(args.sf * ORDERS_PER_SF))
    tables = tpch_plans.Database.tables_for(queries)
    HBM, USABLE, ROWID_MAX = 288e9, 0.8, 4294967294
    sample_orders = 20000

    def col_bytes_per_row(table_id, col):
        typ = tpch_data.SCHEMAS[table_id][col][1]
        import pyarrow as pa

        if pa.types.is_string(typ):
            n = lib.ldb_tpch_host_rows(table_id, sample_orders, 0, 1)
            offs = (C.c_int64 * (n + 1))()
            nbytes = C.c_int64()
            lib.ldb_tpch_host_column(table_id, col, sample_orders, 0, 1, None, offs, C.byref(nbytes))
            return 8.0 + nbytes.value / max(n, 1)  # int64 offsets on the device + the bytes
        if pa.types.is_decimal(typ):
            return 8.0 if args.narrow_decimals else 16.0
        return 4.0

    per_row = {(tid, c): col_bytes_per_row(tid, c) for _, tid, cols in tables for c in cols}
    ranks = []
    for r in range(world):
        rows, resident = {}, 0.0
        for attr, tid, cols in tables:
            n = int(lib.ldb_tpch_host_rows(tid, n_orders, r, world))
            rows[attr] = n
            resident += n * sum(per_row[(tid, c)] for c in cols)
        ranks.append({"rank": r, "rows": rows, "resident_bytes": int(resident), "max_fragment_rows": max(rows.values())})
    # exchange volume of the sharded plans' shuffle steps over base tables (the large ones): the listed columns of every row, (N-1)/N of them leave the rank
    shuffles = []
    for q in queries:
        path = os.path.join(ROOT, "lingo-db_amd", "plans", "tpch", "dist", "q%d.json" % q)
        if world == 1 or not os.path.exists(path):
            continue
        with open(path) as f:
            plan = json.load(f)
        # which base table a value's ROWS come from: a step keeps the row lineage of its "in" (a join probe emits probe rows)
        origin = {name: attr for name, attr in tpch_plans.JSON_PLANS[q].items()}
        width_of = {n: per_row.get((tid, i), 4.0) for _, tid, cols in tables for i, (n, _) in enumerate(tpch_data.SCHEMAS[tid]) if i in cols}
        for st in plan["steps"]:
            src = st.get("in") or st.get("table")
            if st.get("out") and src in origin and st["op"] not in ("groupby", "allgather"):
                origin[st["out"]] = origin[src]
            if st["op"] == "shuffle" and st["in"] in origin and any(a == origin[st["in"]] for a, _, _ in tables):
                attr = origin[st["in"]]
                names = [c if isinstance(c, str) else c["col"] for c in st["cols"]]
                bpr = sum(width_of.get(n, 8.0) for n in names)  # (a computed column: 8 bytes)
                out = max(rk["rows"][attr] for rk in ranks) * bpr * (world - 1) / world
                shuffles.append({"query": q, "input": st["in"], "rows_descend_from": attr, "bytes_per_row": round(bpr, 1), "bytes_out_per_rank_upper_bound": int(out),
                                 "note": "every row of the fragment (the filters and semi joins in front of the shuffle only lower it; Q9's '%green%' keeps 5.4 %)",
                                 "ms_at_xgmi_link_rate_upper_bound": round(out / max(world - 1, 1) / 153e9 * 1e3, 2)})
    worst = max(ranks, key=lambda rk: rk["resident_bytes"])
    biggest_shuffle = max([s["bytes_out_per_rank_upper_bound"] for s in shuffles] or [0])
    # intermediates: a shuffled input arrives once more on the receiving side; hash tables / row-id vectors stay below the largest fragment's touched columns
    need = worst["resident_bytes"] + 2 * biggest_shuffle + 0.25 * worst["resident_bytes"]
    checks = {"row_ids_fit_uint32": all(rk["max_fragment_rows"] <= ROWID_MAX for rk in ranks), "hbm_fits": need <= HBM * USABLE}
    return {"dry_run": True, "n_gpus": world, "sf": args.sf, "queries": queries, "n_orders_total": n_orders, "narrow_decimals": bool(args.narrow_decimals),
            "per_rank": ranks, "shuffles_of_base_tables": shuffles, "hbm_bytes_per_gpu": int(HBM), "hbm_needed_worst_rank": int(need),
            "hbm_model": "resident columns + 2 x the largest shuffled input + 25 % for hash tables and row-id vectors, against 80 % of 288 GB", "checks": checks}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sf", type=float, default=100.0)
    ap.add_argument("--queries", default="", help="TPC-H queries of one step (default: all 22)")
    ap.add_argument("--narrow-decimals", type=int, default=0, help="0: the reference's physical widths (headline); 1: decimals of precision < 19 as 8 bytes; 2: every such decimal and char(1) at the narrowest "
                    "width its column's value range allows (a labelled side line: config.resident_format)")
    ap.add_argument("--cpu-sample-sf", type=float, default=10.0, help="scale of the CPU-baseline sample (0 = skip); the GPU runs the same sample beside it")
    ap.add_argument("--cpu-runs", default="1+3", help="CPU baseline protocol warm-up+measured passes (the reference's tools/scripts/benchmark.py uses 3+10)")
    ap.add_argument("--cpu-reference-legs", type=int, default=1, help="1: cpu_baseline = TPC-H Q1 / Q6 / Q3 at the bench's own scale over the reference's real runtime objects (oracle/_ref, compiled loops, "
                    "3 + 10 runs, the container's CPU quota as thread count); the interpreter legs move to cpu_baseline.interpreter_legs")
    ap.add_argument("--cpu-budget-s", type=float, default=100.0, help="stop starting new CPU legs after this many seconds (the line names the queries measured)")
    ap.add_argument("--oracle-spot-check", type=int, default=1, help="1: Q1 / Q3 / Q6 / Q9 / Q18 by the oracle at the bench's own scale, generated slice by slice on the host and merged (checks.oracle_q*_at_bench_scale; at N > 1 only Q9, 2: all of them)")
    ap.add_argument("--record-runs", type=int, default=3, help="executions per query with replay off after the timed region (per_query_record_ms); 0 = skip")
    ap.add_argument("--plans", default="files", choices=["files", "subop"], help="files: lingo-db_amd/plans/tpch/*.json (the default, the benched configuration); subop: the reference-schema sub-operator dumps "
                    "tests/golden/subop_tpch_qN.json translated by ldb_subop_translate at load time (one GPU) — the plans a LingoDB with the GPU step handler would hand over")
    ap.add_argument("--dry-run", action="store_true", help="no device, no torch: per-rank rows / resident bytes / exchange volume of the configuration against the HBM and row-id budgets")
    args = ap.parse_args()
    if args.dry_run:
        print(json.dumps(dry_run(args)), flush=True)
        return

    # the configuration must fit before anything is generated: per-rank resident bytes + exchange buffers against 288 GB of HBM, fragment rows
    # against the 32-bit row ids (the same arithmetic as --dry-run; a few milliseconds on the host)
    budget = dry_run(args)
    if not all(budget["checks"].values()):
        raise SystemExit("bench.py: --gpus %d --sf %g does not fit: %s (python bench.py --dry-run --gpus %d --sf %g --queries %s shows the per-rank figures)" % (
            args.gpus, args.sf, {k: v for k, v in budget["checks"].items() if not v}, args.gpus, args.sf, args.queries or "1-22"))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback: the product path is the HIP library)")
    # LDB_DIST_BACKEND=gloo lets the N>1 path be exercised functionally on a 1-GPU box (all ranks share device 0, the
    # exchange over the host-staged transport); the real runs use nccl (= RCCL)
    backend = os.environ.get("LDB_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    local_rank = local_rank % n_dev if backend == "gloo" else local_rank
    torch.cuda.set_device(local_rank)
    red_dev = "cpu" if backend == "gloo" else "cuda"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import lingodb_amd as ldb
    import tpch_plans

    queries = [int(q) for q in args.queries.split(",") if q] or list(range(1, 23))
    n_orders = int(round(args.sf * ORDERS_PER_SF))
    ctx = ldb.Context(local_rank)
    info = ctx.device_info()
    t_load = time.perf_counter()
    db = tpch_plans.Database(ctx, n_orders, rank, world, queries, int(args.narrow_decimals))
    ctx.sync()
    load_s = time.perf_counter() - t_load  # one-time: the tables generated straight into HBM (a real deployment registers Arrow batches here)
    comm, exchange = None, "none"
    if world > 1:
        comm, exchange = make_comm(ctx, rank, world, dist, torch, backend)
    runner = tpch_plans.Runner(ctx, db, world, dist if world > 1 else None, torch, comm=comm, plans=args.plans)

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ctx.prof_enable(True)
    # a query ends with its result handed to the host (ldb_gpu_export: the D2H of the result rows is
    # inside the timed region, SURVEY §8(d) protocol); only registration/generation is outside
    # the FIRST execution of a plan pays what the reference reports as compile time (include/lingodb/execution/Timing.h:47-50 lists the lowering /
    # codegen phases beside executionTime): hiprtc specialisation of its kernels, column statistics, hash indexes of the base tables, overflow
    # retries of unestimated tables and the read-back record.  Timed separately (host wall clock, result hand-over included), never part of `value`
    # Round 6: specialisations compile on worker threads while the first execution runs the generic ahead-of-time kernels (the reference's baseline /
    # optimising backend pair, Execution.h:103-104), code objects are kept on disk (~/.cache/ldb_jit).  After the first pass the bench waits for the
    # compilations (untimed, reported as jit.wait_after_first_pass_s) and runs at least one more untimed pass, which loads the specialised modules:
    # the timed region never contains a compilation, a module load or a generic kernel that has a specialised twin
    from lingodb_amd import capi as _capi_jit
    import ctypes as _C

    def _jit_ms():
        n, h, ms = _C.c_int64(), _C.c_int64(), _C.c_double()
        _capi_jit.gpu_lib().ldb_gpu_jit_stats(_C.byref(n), _C.byref(h), _C.byref(ms))
        return ms.value

    def jit_info():
        vals = (_C.c_int64 * 9)()
        _capi_jit.gpu_lib().ldb_gpu_jit_info(vals, 9)
        return dict(zip(("compiled", "memory_hits", "disk_hits", "disk_writes", "outstanding", "failed", "answered_still_compiling", "worker_threads", "taken_from_peer_processes"), [int(v) for v in vals]))

    first_ms = {}
    jit_wait_s = 0.0
    w = 0
    while True:
        asked_before = jit_info()["answered_still_compiling"]
        for q in queries:
            t_q = time.perf_counter()
            runner.run(q).to_arrow()
            if w == 0:
                ctx.sync()
                first_ms[q] = (time.perf_counter() - t_q) * 1000.0
        # every warm-up pass ends with the compile queue drained; a pass that still met a shape for the first time (a group-by sized from the previous
        # execution's group count, a table layout chosen from a statistic cached meanwhile) is followed by another one — at most four extra
        t_w = time.perf_counter()
        pend = _C.c_int64()
        _capi_jit.gpu_lib().ldb_gpu_jit_wait(600_000, _C.byref(pend))
        if w == 0:
            jit_wait_s = time.perf_counter() - t_w
            jit_first = jit_info()
        w += 1
        met_new_shapes = jit_info()["answered_still_compiling"] != asked_before
        if world > 1:  # the ranks run the same number of passes
            flag = torch.tensor([1 if met_new_shapes else 0], device=red_dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            met_new_shapes = bool(flag.item())
        if w >= max(args.warmup, 2) and (not met_new_shapes or w >= max(args.warmup, 2) + 4):
            break
    warmup_passes_run = w
    timers = {q: ctx.timer() for q in queries}
    q_ms = {q: 0.0 for q in queries}
    q_runs = {q: [] for q in queries}
    results = {}
    kernel_ms = {}  # (query, kernel) -> [launches, ms]
    kernel_max = {}  # (query, kernel) -> longest single launch, ms
    # exchange per query (world > 1): ctx-stream time of the transfer groups, bytes this rank sent to other ranks, and what its
    # busiest peer link carried (ldb_gpu_comm_stats) — zeros at N = 1
    exch = {q: {"groups": 0, "bytes_out": 0, "max_peer_bytes_out": 0, "device_ms": 0.0, "host_ms": 0.0} for q in queries}
    if comm is not None:
        comm.stats(reset=True)
    ctx.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for q in queries:
            ctx.timer_start(timers[q])
            results[q] = runner.run(q).to_arrow()
            ctx.timer_stop(timers[q])
            q_runs[q].append(ctx.timer_ms(timers[q]))
            q_ms[q] += q_runs[q][-1]
            if comm is not None:
                st = comm.stats(reset=True)
                for k in exch[q]:
                    exch[q][k] += st[k]
            for k, (n, ms) in ctx.prof_all().items():
                e = kernel_ms.setdefault((q, k), [0, 0.0])
                e[0] += n
                e[1] += ms
                kernel_max[(q, k)] = max(kernel_max.get((q, k), 0.0), ctx.prof_max(k))
            ctx.prof_reset()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1000.0
    # the same plans with replay OFF (plan_replay = 0: every execution records, i.e. waits for each count it reads back — the round-3 execution
    # model), 3 runs per query after the timed region: the cost of the host round trips the replay removes is a number in the line
    record_ms = {}
    if runner.prepared_on and args.record_runs > 0:
        from lingodb_amd import capi as _capi

        lib_opt = _capi.gpu_lib()
        lib_opt.ldb_gpu_set_option(b"plan_replay", 0)
        try:
            for q in queries:
                ts = []
                for _ in range(args.record_runs):
                    ctx.timer_start(timers[q])
                    runner.run(q).to_arrow()
                    ctx.timer_stop(timers[q])
                    ts.append(ctx.timer_ms(timers[q]))
                record_ms[q] = sorted(ts)[len(ts) // 2]
        finally:
            lib_opt.ldb_gpu_set_option(b"plan_replay", 1)
        ctx.prof_reset()
        if world > 1:
            t = torch.tensor([record_ms[q] for q in queries], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            record_ms = {q: float(v) for q, v in zip(queries, t.tolist())}
    per_query = {q: q_ms[q] / args.steps for q in queries}
    if world > 1:
        t = torch.tensor([per_query[q] for q in queries], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per_query = {q: float(v) for q, v in zip(queries, t.tolist())}

    # exchange, reduced over the ranks: time and the busiest link are the maximum (the step waits for the slowest rank), bytes the sum
    ex_ms = [exch[q]["device_ms"] / args.steps for q in queries]
    ex_host = [exch[q]["host_ms"] / args.steps for q in queries]
    ex_peer = [exch[q]["max_peer_bytes_out"] / args.steps for q in queries]
    ex_out = [exch[q]["bytes_out"] / args.steps for q in queries]
    if world > 1:
        tm = torch.tensor([ex_ms, ex_host, ex_peer], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ex_ms, ex_host, ex_peer = tm.tolist()
        ts = torch.tensor(ex_out, device=red_dev, dtype=torch.float64)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        ex_out = ts.tolist()

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel: the fused scan+filter+aggregate kernel of Q1
        roof_q = 1
        n_l, ms_l = kernel_ms.get((roof_q, "k_groupby"), [0, 0.0]) if 1 in queries else (0, 0.0)  # (the 76 B/row model is Q1's: no Q1, no dominant-kernel entry)
        rows_local = db.lineitem.rows
        bpr = Q1_BYTES_PER_ROW_NARROW if args.narrow_decimals else Q1_BYTES_PER_ROW
        if args.narrow_decimals >= 2 and 1 in queries:
            # the compressed resident format (narrowest width per column): the kernel's algorithmic bytes are the widths the columns really have
            bpr = sum(db.lineitem.col_width(db.lineitem.col(c)) for c in ("l_shipdate", "l_returnflag", "l_linestatus", "l_quantity", "l_extendedprice", "l_discount", "l_tax"))
        roofline = None
        if n_l:
            avg_ms = ms_l / n_l
            achieved = rows_local * bpr / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": "k_groupby (TPC-H Q%d: scan+filter+hash aggregate)" % roof_q, "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                        "avg_kernel_ms": round(avg_ms, 4), "launches": n_l, "algorithmic_bytes_per_launch": rows_local * bpr,
                        "bytes_per_row": bpr}
        extras = {}
        if 6 in queries:
            # the scan headline of SURVEY §8(d): Q6-shape filter + sum.  The kernel reads its conjunct columns only for
            # rows that survived the earlier conjuncts, so the HBM bytes it moves are BELOW the full-column figure; the
            # fraction uses the bytes the PMC pass measured for this kernel when a summary of the matching configuration
            # is committed
            n6, ms6 = kernel_ms.get((6, "k_groupby"), [0, 0.0])
            if n6:
                t6 = ms6 / n6 * 1e-3
                extras["scan_q6"] = {"kernel_ms": round(ms6 / n6, 4), "rows_per_s_G": round(rows_local / t6 / 1e9, 1)}
                pmc6 = next((p for p in (os.path.join(ROOT, "profiles", "r%02d_pmc_q6_sf%g.json" % (r, args.sf)) for r in (6, 5, 4, 3, 2, 1)) if os.path.exists(p)), None)
                if world == 1 and pmc6:
                    with open(pmc6) as f:
                        k6 = json.load(f)["kernels"]
                    k6 = k6.get("k_groupby_spec") or k6.get("k_groupby")
                    if k6:
                        gbs = k6["fetch_bytes"] / t6 / 1e9
                        extras["scan_q6"].update({"hbm_gbs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": round(k6["fetch_bytes"]),
                                                  "traffic_source": os.path.relpath(pmc6, ROOT)})
        if roofline:
            # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this same
            # command; tools/pmc_summary.py).  Counters cannot be read from inside the timed process, so the committed
            # summary of the matching configuration is quoted; null when there is none.
            tag = "_narrow" if args.narrow_decimals else ""
            pmc_path = next((p for p in (os.path.join(ROOT, "profiles", "r%02d_pmc_q%d_sf%g%s.json" % (r, roof_q, args.sf, tag)) for r in (6, 5, 4, 3, 2, 1)) if os.path.exists(p)), None)
            if world == 1 and pmc_path:
                with open(pmc_path) as f:
                    pmc = json.load(f)
                k = pmc["kernels"].get("k_groupby_spec") or pmc["kernels"].get("k_groupby")
                if k:
                    roofline["traffic"] = round(k["fetch_bytes"] + k.get("write_raw_bytes", 0.0))
                    roofline["traffic_source"] = os.path.relpath(pmc_path, ROOT)
        probe = runner.probe_microbench() if world == 1 and db.orders is not None and any(q in queries for q in (3, 4, 18)) else None  # needs o_orderkey, o_orderdate, l_orderkey
        ceiling = runner.hbm_ceiling() if world == 1 else None
        # further kernels against the same HBM roofline.  Probe byte model: the COMPULSORY bytes of one launch — the probe key
        # column once (4 B per row) + the table once (every line of it is touched by a 100 % / 10 % match probe) — not
        # "key + one slot per row": with a table of range / 4 bytes (rank bitmap) or 4 B per key value (direct words) most slot
        # reads of neighbouring rows hit the same cache line, and a per-row figure would exceed what DRAM can deliver.
        # SURVEY §8(d)'s per-row figure (4 B key + 8 B slot) is reported beside it as `survey_model_gbs`.
        more = []
        if probe and "probe_ms" in probe:
            for name, sub in (("FK probe, clustered keys (l_orderkey → o_orderkey, 100 % match)", probe), ("FK probe, unclustered keys (random order keys, 100 % match)", probe.get("unclustered")),
                              ("FK probe, selective build side (10 % of orders), clustered keys", probe.get("selective"))):
                if sub and "probe_ms" in sub:
                    rows = sub.get("probe_rows", probe["probe_rows"])
                    tb = sub.get("table_bytes", probe["table_bytes"])
                    gbs = (rows * 4 + tb) / (sub["probe_ms"] * 1e-3) / 1e9
                    more.append({"kernel": "k_join_probe_count: " + name, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                 "compulsory_bytes_per_launch": rows * 4 + tb, "table_bytes": tb, "survey_model_gbs": round(rows * 12 / (sub["probe_ms"] * 1e-3) / 1e9, 1),
                                 "grows_per_s": sub["probe_grows_per_s"], "avg_kernel_ms": sub["probe_ms"]})
                    if sub.get("radix_partitioned"):
                        # the partitioned path MOVES more than the compulsory bytes on purpose: two histogram passes over the keys, two scatter
                        # passes (keys in, (key, row) out; (key, row) in and out), the keys once more for the LDS-staged probe, the table once
                        moved = rows * 4 * 2 + rows * (4 + 8) + rows * 16 + rows * 4 + tb
                        more[-1].update({"kernel": "k_radix_hist + k_radix_scatter + k_join_probe_count (LDS-staged): " + name, "kernels_ms": sub.get("kernels_ms"),
                                         "partitioned_bytes_per_launch": moved, "partitioned_gbs": round(moved / (sub["probe_ms"] * 1e-3) / 1e9, 1),
                                         "partitioned_frac": round(moved / (sub["probe_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        if (18, "k_groupby") in kernel_max:
            # Q18 launches k_groupby twice (the 600 M → 150 M aggregation and a tiny final group-by): the LARGEST launch is
            # priced, and the §8(d) byte model (rows x (key + agg input) + groups x entry x 2) covers the sorted-key pre-pass
            # and the finalisation too, so their time is in the denominator
            wd = 8 if args.narrow_decimals else 16
            b18 = rows_local * (4 + wd) + (db.orders.rows if db.orders else 0) * 32 * 2
            t_main = kernel_max[(18, "k_groupby")]
            t_all = t_main + kernel_max.get((18, "k_gb_sorted_heads"), 0.0) + kernel_max.get((18, "k_gb_finalize"), 0.0)
            for label, tm in (("k_groupby (TPC-H Q18: 600 M rows → 150 M groups; the aggregation launch alone)", t_main),
                              ("k_gb_sorted_heads + k_groupby + k_gb_finalize (TPC-H Q18, everything the byte model covers)", t_all)):
                gbs = b18 / (tm * 1e-3) / 1e9
                more.append({"kernel": label, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                             "algorithmic_bytes_per_launch": b18, "kernel_ms": round(tm, 4)})
        # result checksums (one per query, of the rows handed to the host) + conservation laws that tie the
        # SF100 results to an independent kernel path: the oracle cannot run at this size in seconds
        import zlib

        checks = {"checksum": {"Q%d" % q: "%08x" % (zlib.crc32(repr(results[q].to_pylist()).encode()) & 0xFFFFFFFF) for q in queries if q in results}}
        if world == 1 and 1 in results:
            from lingodb_amd import api, capi

            n_pass = db.lineitem.rel().scan_count([api.pred((0, db.lineitem.col("l_shipdate")), capi.F_LTE, 10471)])
            checks["q1_count_conservation"] = bool(sum(results[1].column(9).to_pylist()) == n_pass)  # Σ count(*) over groups == rows passing the filter (scan kernel)
        if world == 1 and 6 in results and 1 in results:
            checks["q6_rows"] = results[6].num_rows == 1
        if world == 1 and 6 in results and args.oracle_spot_check:
            # the oracle at the bench's OWN scale for one query: Q6 over the host-generated columns, slice by slice (≈ 10 s of host time at SF100)
            try:
                want, secs = tpch_plans.oracle_q6_at_scale(n_orders)
                got = results[6].column(0)[0].as_py()
                got = None if got is None else int(got.scaleb(results[6].schema.field(0).type.scale))
                checks["oracle_q6_at_bench_scale"] = {"equal": bool(got == want), "sf": args.sf, "seconds": round(secs, 1), "value_unscaled": str(want)}
            except Exception as e:  # the spot check must not cost the bench line
                checks["oracle_q6_at_bench_scale"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if args.oracle_spot_check:
            # BASELINE configs[1] / [2] / [4]'s queries at the bench's own scale: Q1 (partials of every slice added, averages at the end), Q3 (order-range
            # slices: all customers x the slice's orders and lineitems, ten best rows per slice, merged), Q9 (order-range slices against the green parts and
            # their partsupp rows, partial sums per nation and year added) and Q18 (a group never crosses an order-range slice; the big orders meet the
            # customers at the end) by the oracle, against the rows the timed executions returned.  At N > 1 every rank holds the whole result of these
            # plans (the partials are all-gathered), so rank 0's rows are compared the same way
            slices = max(8, min(512, n_orders // 250_000))
            for q, fn, kw in ((1, tpch_plans.oracle_q1_at_scale, {"n_parts": slices}), (3, tpch_plans.oracle_q3_at_scale, {"n_parts": slices}),
                              (9, tpch_plans.oracle_q9_at_scale, {}), (18, tpch_plans.oracle_q18_at_scale, {"n_parts": slices})):
                if q not in results or (world > 1 and q != 9 and args.oracle_spot_check < 2):
                    continue
                try:
                    want, secs = fn(n_orders, **kw)
                    ok = tpch_plans.matches_legs(q, tpch_plans._canon(results[q]), want)
                    checks["oracle_q%d_at_bench_scale" % q] = {"equal": bool(ok), "sf": args.sf, "seconds": round(secs, 1), "rows_compared": results[q].num_rows}
                except Exception as e:
                    checks["oracle_q%d_at_bench_scale" % q] = {"error": "%s: %s" % (type(e).__name__, e)}
        # no roofline fraction may exceed what HBM can give a streaming read: the larger of this box's own scan ceiling (+ 10 %: the
        # calibration scan is itself a kernel of this library, not the hardware limit) and the guide's ≈ 6.3 TB/s achievable figure
        if ceiling and "scan_count_gbs" in ceiling:
            cap = max(ceiling["scan_count_gbs"] * 1.1, 6300.0) / HBM_PEAK_GBS
            over = [m["kernel"] for m in more + ([roofline] if roofline else []) if m["frac"] > cap]
            checks["roofline_below_stream_ceiling"] = not over
            if over:
                print("[bench] roofline fractions above the measured streaming ceiling — the byte model of these entries is wrong: %s" % over, file=sys.stderr, flush=True)
        cpu = None
        if world == 1 and args.cpu_sample_sf > 0:
            # release the SF`--sf` database first: the sample database of the same-SF GPU leg needs room only when SF is huge, but the
            # host legs keep whole tables as numpy arrays
            cpu = tpch_plans.cpu_baseline(queries, args.cpu_sample_sf, args.cpu_runs, args.cpu_budget_s, ctx=ctx, narrow=int(args.narrow_decimals), checks=checks)
        if world == 1 and args.cpu_reference_legs and not args.narrow_decimals:
            # the baseline proper (round 6): Q1 / Q6 / Q3 at THIS scale over the reference's real runtime objects, compiled; the interpreter legs above (all 22
            # queries at the sample scale, kind "port") stay beside it as `interpreter_legs`
            try:
                ref_legs = tpch_plans.cpu_reference_legs(ctx, db, n_orders, results, checks)
            except Exception as e:
                ref_legs = None
                checks["reference_objects_error"] = "%s: %s" % (type(e).__name__, e)
            if ref_legs is not None:
                ref_legs["interpreter_legs"] = cpu
                cpu = ref_legs
        # all kernels, helpers included: Σ kernel durations ÷ wall span per query from a rocprofv3 kernel trace of the
        # same plans on the same data (tools/query_timeline.py + tools/timeline_summary.py; profile, not this run)
        gpu_busy = None
        tl_path = next((p for p in (os.path.join(ROOT, "profiles", "r%02d_query_timeline_sf%g.json" % (r, args.sf)) for r in (6, 5, 4, 3, 2)) if os.path.exists(p)), None)
        if world == 1 and tl_path:
            with open(tl_path) as f:
                tl = json.load(f)
            gpu_busy = {"share": tl["total"]["busy_share"], "busy_ms": tl["total"]["busy"], "span_ms": tl["total"]["span"], "source": os.path.relpath(tl_path, ROOT)}
        # exchange summary: per query the ctx-stream time of its transfer groups, the bytes that left the ranks, and the rate of the
        # busiest peer link (every peer pair of one MI355X node has its own xGMI link, ≈ 153 GB/s)
        XGMI_LINK_GBS = 153.0
        ex_q = {}
        for i, q in enumerate(queries):
            link = ex_peer[i] / (ex_ms[i] * 1e-3) / 1e9 if ex_ms[i] > 0 else 0.0
            ex_q["Q%d" % q] = {"exchange_ms": round(ex_ms[i], 4), "exchange_host_ms": round(ex_host[i], 4), "exchange_bytes_out": int(ex_out[i]), "exchange_bytes_out_max_peer": int(ex_peer[i]),
                               "peer_link_gbs": round(link, 2), "xgmi_link_frac": round(link / XGMI_LINK_GBS, 4)}
        exchange_out = {"transport": exchange, "xgmi_link_peak_gbs": XGMI_LINK_GBS, "exchange_ms_per_step": round(sum(ex_ms), 4), "exchange_bytes_out_per_step": int(sum(ex_out)),
                        "per_query": ex_q if world > 1 else {"all": {"exchange_ms": 0.0, "exchange_bytes_out": 0, "exchange_bytes_out_max_peer": 0, "peer_link_gbs": 0.0, "xgmi_link_frac": 0.0}}}
        plan_stats = runner.prepared_stats()
        if roofline is not None:
            roofline["more"] = more  # (inside `roofline` so that the driver's record of the line keeps the other kernels' fractions)
        # key order: the driver's record keeps the standard keys + config / roofline / cpu_baseline and the TAIL of the line (8 KB) — bulky
        # tables first; what must survive comes last: exchange, prepared-plan statistics, the parity checks, both timing modes, per-query times
        checksums = checks.pop("checksum")
        out = {
            "kernel_ms_per_step": {"Q%d:%s" % (q, k): round(v[1] / args.steps, 4) for (q, k), v in sorted(kernel_ms.items())},
            "roofline_more": more,
            "hbm_ceiling": ceiling,
            "per_query_median_ms": {"Q%d" % q: round(sorted(q_runs[q])[len(q_runs[q]) // 2], 4) for q in queries},
            "per_query_min_ms": {"Q%d" % q: round(min(q_runs[q]), 4) for q in queries},
        }
        if probe:
            out["join_probe"] = probe
        out.update(extras)
        out.update({
            "metric": "tpch_sf%g_geomean_ms" % args.sf,
            "value": round(geomean([per_query[q] for q in queries]), 4),
            "unit": "ms",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": False,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int64/int128 decimal",
            "data": "synthetic",
            "config": {"workload": "TPC-H SF%g %s on %d x MI355X, Arrow columns resident in HBM (synthetic dbgen-shaped data, seed 20260925)" % (
                args.sf, "+".join("Q%d" % q for q in queries), world), "queries": queries, "rows_lineitem_total": int(db.n_lineitem_total),
                "narrow_decimals": (int(args.narrow_decimals) if args.narrow_decimals >= 2 else bool(args.narrow_decimals)), "device": info["name"],
                **({"resident_format": "narrow level 2 — NOT the headline configuration: decimals of precision < 19 at the narrowest of 1 / 2 / 4 / 8 bytes their column's value range allows, "
                                       "char(1) at one byte; widened in registers, arithmetic in i64 / i128 as LowerToStd.cpp:128-132,1479-1486: bit-identical results (checks.*); "
                                       "roofline.frac is on the COMPRESSED bytes (roofline.bytes_per_row)"} if args.narrow_decimals >= 2 else {}), "exchange": exchange, "load_s": round(load_s, 3),
                **({"functional_only": "all %d ranks share ONE GPU and exchange over the host-staged transport (LDB_DIST_BACKEND=gloo): a functional run of the sharded plans, "
                                       "not a scaling measurement" % world} if world > 1 and backend == "gloo" else {}),
                "plans": ("tests/golden/subop_tpch_q*.json: sub-operator dumps in the reference's mlir-subop-to-json schema, translated by ldb_subop_translate (straightforward join trees, no eager aggregation)"
                          if args.plans == "subop" else
                          "lingo-db_amd/plans/tpch/%s*.json: hand-ordered operator plans (join orders, eager aggregation), not LingoDB's optimiser output" % ("dist/" if world > 1 else "")),
                "execution": ("prepared plans (ldb_plan_prepare / ldb_plan_execute): parsed once; executions after the warm-up replay their read-back trace (no host wait between "
                              "operators, one check at the end%s) and reuse cached descriptors" % (", the verdict agreed by all ranks" if world > 1 else "")
                              if runner.prepared_on else "ldb_plan_run_json per execution (LDB_BENCH_PREPARED=0)"),
                "built_during_warmup": "outside the timed region, kept with the base tables like the reference's catalog statistics and persisted PK hash indexes "
                                       "(LingoDBHashIndex): column min / max and sortedness, zone maps (kept where selective), utf8 dictionaries (at registration), hash indexes of "
                                       "unique join_build steps over bare base tables (part, supplier, nation, region, customer, orders), hiprtc kernel variants, read-back traces",
                "row_ids": "uint32: at most 4.29 G rows per GPU fragment"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            # share of the per-query time inside the operators' HIP-event-bracketed kernels (the main kernel of every
            # operator + the group-by finalisation; helper launches — scans, compactions, gathers — are not bracketed)
            "kernel_share": round(sum(v[1] for v in kernel_ms.values()) / max(sum(q_ms.values()), 1e-9), 4),
            "gpu_busy": gpu_busy,
            "result_checksum": checksums,
            "exchange": exchange_out,
            "prepared_plans": plan_stats,
            "checks": checks,
            # timing modes beside `per_query_ms` (replayed executions, what `value` is the geomean of): the first execution of each plan (host wall clock:
            # hiprtc + statistics + indexes + the recording run — the reference's compile-time columns, Timing.h:47-50) and the steady state with replay off
            "jit": dict(jit_info(), compile_ms_total=round(_jit_ms(), 1), wait_after_first_pass_s=round(jit_wait_s, 2), after_first_pass=jit_first, warmup_passes_run=warmup_passes_run,
                        mode="asynchronous: the first execution runs the generic kernels while hiprtc compiles on worker threads; code objects cached on disk"),
            "first_execution_ms": {"Q%d" % q: round(first_ms[q], 1) for q in queries if q in first_ms},
            "per_query_record_ms": {"Q%d" % q: round(record_ms[q], 4) for q in queries if q in record_ms},
            "record_geomean_ms": round(geomean([record_ms[q] for q in queries]), 4) if len(record_ms) == len(queries) else None,
            "per_query_ms": {"Q%d" % q: round(v, 4) for q, v in per_query.items()},
        })
    barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
