"""Stress check of every probe kernel on the specialised + fused path at SF10 against numpy, with
torch initialised in the process (the configuration bench.py runs in).  Prints one line per kind."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lingo-db_amd"), ROOT]
import numpy as np
import torch; torch.cuda.set_device(0); x = torch.zeros(10, device="cuda")
import lingodb_amd as ldb
from lingodb_amd import api, capi
n = int(float(sys.argv[1]) * 1_500_000) if len(sys.argv) > 1 else 15_000_000
reps = 3
ctx = ldb.Context(0)
li = ctx.tpch_generate(0, n, cols=[0, 2, 10]); su = ctx.tpch_generate(4, n, cols=[0, 1]); od = ctx.tpch_generate(1, n, cols=[0, 4])
lsk = np.frombuffer(li.read_fixed(1).tobytes(), dtype=np.int32); lsd = np.frombuffer(li.read_fixed(2).tobytes(), dtype=np.int32)
lok = np.frombuffer(li.read_fixed(0).tobytes(), dtype=np.int32)
snk = np.frombuffer(su.read_fixed(1).tobytes(), dtype=np.int32); ssk = np.frombuffer(su.read_fixed(0).tobytes(), dtype=np.int32)
sel_s = np.isin(snk, [6, 7]); keys = ssk[sel_s]
passing = (lsd >= 9131) & (lsd <= 9861)
inset = np.isin(lsk, keys)
want_rows = np.nonzero(passing & inset)[0]
key_to_row = {int(k): i for i, k in enumerate(keys)}
want_build = np.array([key_to_row[int(k)] for k in lsk[want_rows]], dtype=np.uint32)
supps = su.rel().scan_filter([api.pred((0, 1), capi.F_IN, values=[6, 7])]).materialize([(0, 0), (0, 1)])
hs = supps.rel().join_build([(0, 0)], unique=True)
preds = lambda: [api.pred((0, 2), capi.F_GTE, 9131), api.pred((0, 2), capi.F_LTE, 9861)]
ok = True
def report(name, good):
    global ok
    ok &= bool(good); print(f"{name:28s} {'ok' if good else 'MISMATCH'}", flush=True)
for lazy in (1, 0):
    capi.gpu_lib().ldb_gpu_set_option(b"lazy_filter", lazy)
    tag = "fused" if lazy else "forced"
    for it in range(reps):
        l1 = li.rel().scan_filter(preds())
        r = hs.probe(l1, [(0, 1)])
        report(f"inner unique {tag} #{it}", r.rows == len(want_rows) and np.array_equal(r.rowids(0), want_rows) and np.array_equal(r.rowids(1), want_build))
        l1 = li.rel().scan_filter(preds())
        report(f"semi {tag} #{it}", np.array_equal(hs.probe(l1, [(0, 1)], capi.JOIN_SEMI).rowids(0), want_rows))
        l1 = li.rel().scan_filter(preds())
        report(f"anti {tag} #{it}", np.array_equal(hs.probe(l1, [(0, 1)], capi.JOIN_ANTI).rowids(0), np.nonzero(passing & ~inset)[0]))
        l1 = li.rel().scan_filter(preds())
        report(f"count {tag} #{it}", hs.probe_count(l1, [(0, 1)]) == len(want_rows))
        l1 = li.rel().scan_filter(preds())
        sb = hs.probe(l1, [(0, 1)], capi.JOIN_SEMI_BUILD)
        report(f"semi_build {tag} #{it}", np.array_equal(sb.rowids(0), np.unique(want_build)))
# duplicate build keys (pairs path): build on the filtered lineitem's l_orderkey, probe with orders
capi.gpu_lib().ldb_gpu_set_option(b"lazy_filter", 1)
ook = np.frombuffer(od.read_fixed(0).tobytes(), dtype=np.int32)
lsel = np.nonzero((lsd >= 9131) & (lsd <= 9200))[0]
hb = li.rel().scan_filter([api.pred((0, 2), capi.F_GTE, 9131), api.pred((0, 2), capi.F_LTE, 9200)]).join_build([(0, 0)])
import collections
cnt = collections.Counter(lok[lsel].tolist())
want_pairs = sum(cnt.get(int(k), 0) for k in ook.tolist()) if len(ook) < 2_000_000 else int(np.isin(lok[lsel], ook).sum())
for it in range(reps):
    r = hb.probe(od.rel(), [(0, 0)])
    a, b = r.rowids(0), r.rowids(1)
    good = r.rows == want_pairs and np.array_equal(ook[a], lok[b]) and len(np.unique(b)) == len(b) and bool(np.isin(b, lsel).all())
    report(f"inner pairs #{it}", good)
# radix clustering of the probe side forced on: same pairs / counts as without it
L = capi.gpu_lib()
pkt = ctx.tpch_generate(8, n, cols=[0])
hod = od.rel().join_build([(0, 0)], unique=True)
pkv = np.frombuffer(pkt.read_fixed(0).tobytes(), dtype=np.int32)
okey_to_row = np.zeros(int(ook.max()) + 1, dtype=np.int64) - 1
okey_to_row[ook] = np.arange(len(ook))
for mode in (0, 1):
    L.ldb_gpu_set_option(b"join_radix", mode)
    report(f"count radix={mode}", hod.probe_count(pkt.rel(), [(0, 0)]) == pkt.rows)
    r = hod.probe(pkt.rel(), [(0, 0)])
    a, b = r.rowids(0), r.rowids(1)
    report(f"inner radix={mode}", r.rows == pkt.rows and r.sides == 2 and len(np.unique(a)) == len(a) and np.array_equal(okey_to_row[pkv[a]], b.astype(np.int64)))
    sb = hod.probe(pkt.rel(), [(0, 0)], capi.JOIN_SEMI_BUILD)
    report(f"semi_build radix={mode}", np.array_equal(sb.rowids(0), np.unique(okey_to_row[pkv]).astype(np.uint32)))
    lo = hs.probe(pkt.rel(), [(0, 0)], capi.JOIN_LEFT_OUTER)  # mostly unmatched: supplier keys vs order keys
    report(f"left_outer radix={mode}", lo.rows == pkt.rows)
L.ldb_gpu_set_option(b"join_radix", -1)
print("ALL OK" if ok else "FAILED")
