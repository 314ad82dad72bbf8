"""Per-query time and kernel table of selected TPC-H plans on ONE GPU without torch (a quick check
beside bench.py, which stays the contract): python tools/query_time.py --sf 100 --queries 10
Times exactly what bench.py times per query (plan + result hand-over, HIP events on the ctx stream)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lingo-db_amd"), ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=100.0)
    ap.add_argument("--queries", default="10")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    import lingodb_amd as ldb
    import tpch_plans

    queries = [int(q) for q in args.queries.split(",") if q]
    ctx = ldb.Context(0)
    db = tpch_plans.Database(ctx, int(round(args.sf * 1_500_000)), 0, 1, queries, False)
    runner = tpch_plans.Runner(ctx, db, 1, None, None)
    ctx.prof_enable(True)
    for _w in range(max(args.warmup, 2)):
        if _w == 1:  # the first pass ran the generic kernels while the specialisations compiled (asynchronous JIT): wait, then warm up on the specialised ones
            tpch_plans._jit_wait()
        for q in queries:
            rows = runner.run(q).to_arrow().num_rows
    out = {"sf": args.sf, "steps": args.steps, "warmup": args.warmup, "lineitem_rows": int(db.lineitem.rows), "queries": {}}
    for q in queries:
        t = ctx.timer()
        ctx.prof_reset()
        ms = []
        for _ in range(args.steps):
            ctx.timer_start(t)
            rows = runner.run(q).to_arrow().num_rows
            ctx.timer_stop(t)
            ms.append(ctx.timer_ms(t))
        kernels = {k: {"launches": n, "avg_ms": round(m / n, 4)} for k, (n, m) in sorted(ctx.prof_all().items(), key=lambda kv: -kv[1][1]) if n}
        out["queries"]["Q%d" % q] = {"ms_median": round(sorted(ms)[len(ms) // 2], 4), "ms_min": round(min(ms), 4), "result_rows": rows, "kernels": kernels}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
