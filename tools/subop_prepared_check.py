"""Which translated sub-operator dump is slow as a PREPARED plan: each query three times through Runner(plans="subop") at the given
scale, one line per execution, flushed (a hang shows as the last line): python tools/subop_prepared_check.py --sf 10"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lingo-db_amd"), ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=10.0)
    ap.add_argument("--queries", default="")
    ap.add_argument("--runs", type=int, default=3)
    args = ap.parse_args()
    import lingodb_amd as ldb
    import tpch_plans

    queries = [int(q) for q in args.queries.split(",") if q] or list(range(1, 23))
    ctx = ldb.Context(0)
    t0 = time.time()
    db = tpch_plans.Database(ctx, int(round(args.sf * 1_500_000)), 0, 1, queries, False)
    print("database %.1f s" % (time.time() - t0), flush=True)
    runner = tpch_plans.Runner(ctx, db, 1, None, None, plans="subop")
    for q in queries:
        for r in range(args.runs):
            print("Q%d run %d …" % (q, r), end=" ", flush=True)
            t = time.time()
            rows = runner.run(q).to_arrow().num_rows
            print("%d rows, %.1f ms" % (rows, (time.time() - t) * 1e3), flush=True)
    print(runner.prepared_stats(), flush=True)


if __name__ == "__main__":
    main()
