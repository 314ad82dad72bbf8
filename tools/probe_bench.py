"""FK-probe micro-benchmark alone: clustered l_orderkey, unclustered random order keys and
the selective variant, at --sf; prints one JSON line.  LDB_JIT_DEFINES varies the kernel shape."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lingo-db_amd"), ROOT]
import lingodb_amd as ldb
from lingodb_amd import api, capi
ap = argparse.ArgumentParser(); ap.add_argument("--sf", type=float, default=100.0); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
n = int(a.sf * 1_500_000)
ctx = ldb.Context(0); ctx.prof_enable(True)
od = ctx.tpch_generate(1, n, cols=[0, 4]); li = ctx.tpch_generate(0, n, cols=[0]); pk = ctx.tpch_generate(8, n, cols=[0])
out = {"defines": os.environ.get("LDB_JIT_DEFINES", "")}
def run(name, ht, rel, key):
    ht.probe_count(rel, [(0, key)]); ctx.prof_reset()
    for _ in range(a.reps): m = ht.probe_count(rel, [(0, key)])
    k, ms = ctx.prof_all().get("k_join_probe_count", (0, 0.0))
    out[name] = {"ms": round(ms / k, 4), "grows_per_s": round(rel.rows / (ms / k * 1e-3) / 1e9, 2), "matches": m}
ht = od.rel().join_build([(0, 0)], unique=True)
out["build_ms"] = round(ctx.prof_all().get("k_join_build", (1, 0.0))[1], 3)
L = capi.gpu_lib()
L.ldb_gpu_set_option(b"join_radix", 0)
run("clustered", ht, li.rel(), 0); run("unclustered", ht, pk.rel(), 0)
L.ldb_gpu_set_option(b"join_radix", -1)
def run_radix(name, ht, rel, key):
    ht.probe_count(rel, [(0, key)]); ctx.prof_reset()
    for _ in range(a.reps): m = ht.probe_count(rel, [(0, key)])
    pr = ctx.prof_all()
    tot = sum(pr.get(k, (0, 0.0))[1] for k in ("k_join_probe_count", "k_radix_hist", "k_radix_scatter")) / a.reps
    out[name] = {"ms": round(tot, 4), "grows_per_s": round(rel.rows / (tot * 1e-3) / 1e9, 2), "matches": m, "kernels_ms": {k: round(v[1] / a.reps, 4) for k, v in pr.items()}}
if os.environ.get("PB_RADIX"):
    L.ldb_gpu_set_option(b"join_radix", 1); L.ldb_gpu_set_option(b"join_radix_part_bytes", 1 << 28)
    run_radix("unclustered_radix", ht, pk.rel(), 0)
L.ldb_gpu_set_option(b"join_radix", 0)
for opt in os.environ.get("PB_OPTS", "").split(","):
    if opt:
        k, v = opt.split("="); capi.gpu_lib().ldb_gpu_set_option(k.encode(), int(v))
        ht2 = od.rel().join_build([(0, 0)], unique=True)
        run("clustered[%s]" % opt, ht2, li.rel(), 0); run("unclustered[%s]" % opt, ht2, pk.rel(), 0)
ht.release()
sel = od.rel().scan_filter([api.pred((0, 1), capi.F_LT, 8279)])
hs = sel.join_build([(0, 0)], unique=True)
run("selective", hs, li.rel(), 0); run("selective_unclustered", hs, pk.rel(), 0)
print(json.dumps(out))
