"""Run each TPC-H plan a few times with a trace marker before every run, so that a
`rocprofv3 --kernel-trace` of this process can be cut into per-query timelines:

  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/query_timeline.py --sf 100
  python tools/timeline_summary.py <dir>/*/*_kernel_trace.csv --out profiles/r02_query_timeline_sf100.json

Marker before run r of query q: grid = 100 * q + r (r = 1 … runs); 9999 closes the last run."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lingo-db_amd"), ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=100.0)
    ap.add_argument("--queries", default=",".join(str(q) for q in range(1, 23)))
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--prof", type=int, default=0, help="1 = keep the library's own HIP-event brackets on (as bench.py does)")
    args = ap.parse_args()
    import lingodb_amd as ldb
    import tpch_plans

    queries = [int(q) for q in args.queries.split(",") if q]
    ctx = ldb.Context(0)
    db = tpch_plans.Database(ctx, int(round(args.sf * 1_500_000)), 0, 1, queries, False)
    runner = tpch_plans.Runner(ctx, db, 1, None, None)
    ctx.prof_enable(bool(args.prof))
    for _w in range(max(args.warmup, 2)):
        if _w == 1:  # the first pass ran the generic kernels while the specialisations compiled (asynchronous JIT): wait, then warm up on the specialised ones
            tpch_plans._jit_wait()
        for q in queries:
            runner.run(q).to_arrow()
    ctx.sync()
    for q in queries:
        for r in range(1, args.runs + 1):
            ctx.prof_marker(100 * q + r)
            runner.run(q).to_arrow()
    ctx.prof_marker(9999)
    ctx.sync()


if __name__ == "__main__":
    main()
