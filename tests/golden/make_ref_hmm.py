#!/usr/bin/env python3
"""Golden answers of outer joins that keep the build side, computed over the reference's REAL HashMultiMap
(src/runtime/HashMultiMap.cpp, compiled into oracle/_ref by oracle/ref_build/build_ref.sh; glue ref_hmm_outer_join = the
insert / lookup / flag-scatter / scan-unflagged sequence of translateHJWithMarker, RelAlgToSubOp.cpp:1248-1287): matched
(probe row, build row) pairs, the build rows no probe reached (the NULL-extended rows of a right / full outer join) and the
probe rows without a partner (the other half of a full outer join).  Build keys with heavy duplication, NULLs and keys no
probe has; several initial capacities so that HashMultiMap::resize runs.  Writes tests/golden/ref_hmm.npz — the reference
tree does not exist on the GPU box.  Run in the build container:  python tests/golden/make_ref_hmm.py"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def run(lib, bk, bv, pk, pv, cap0):
    nb, npr = len(bk), len(pk)
    cap = 1 << 22
    op, ob = np.zeros(cap, np.int64), np.zeros(cap, np.int64)
    ub, nu = np.zeros(max(nb, 1), np.int64), C.c_int64()
    pm = np.zeros(max(npr, 1), np.uint8)
    n = lib.ref_hmm_outer_join(bk.ctypes.data, bv.ctypes.data, nb, pk.ctypes.data, pv.ctypes.data, npr, cap0, op.ctypes.data, ob.ctypes.data, cap, ub.ctypes.data, C.byref(nu), pm.ctypes.data)
    assert 0 <= n <= cap
    return op[:n].copy(), ob[:n].copy(), ub[: nu.value].copy(), pm[:npr].copy()


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libldb_ref.so"))
    lib.ref_hmm_outer_join.restype = C.c_int64
    lib.ref_hmm_outer_join.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
    rng = np.random.default_rng(20260927)
    out = {}
    cases = [(7, 8, 20, 4), (3000, 5000, 700, 4), (20000, 30000, 50000, 1024), (1, 0, 5, 4), (0, 9, 5, 4), (5000, 5000, 40, 16)]  # (build rows, probe rows, key range, initial capacity)
    for c, (nb, npr, rng_k, cap0) in enumerate(cases):
        bk = rng.integers(0, rng_k, nb).astype(np.int64)
        pk = rng.integers(-rng_k // 4, rng_k + rng_k // 4, npr).astype(np.int64)
        bv = (rng.integers(0, 10, nb) > 0).astype(np.uint8)
        pv = (rng.integers(0, 12, npr) > 0).astype(np.uint8)
        op, ob, ub, pm = run(lib, bk, bv, pk, pv, cap0)
        # the same through a second capacity: the container's growth must not change the answer
        op2, ob2, ub2, pm2 = run(lib, bk, bv, pk, pv, 4 * cap0 + 8)
        assert sorted(zip(op.tolist(), ob.tolist())) == sorted(zip(op2.tolist(), ob2.tolist())) and sorted(ub.tolist()) == sorted(ub2.tolist()) and (pm == pm2).all()
        out.update({"c%d_bk" % c: bk, "c%d_bv" % c: bv, "c%d_pk" % c: pk, "c%d_pv" % c: pv, "c%d_pairs_p" % c: op, "c%d_pairs_b" % c: ob, "c%d_unmatched_b" % c: ub, "c%d_probe_matched" % c: pm})
        print("case", c, "pairs", len(op), "unmatched build rows", len(ub), "unmatched probe rows", int((pm == 0).sum()))
    out["n_cases"] = np.array([len(cases)])
    np.savez_compressed(os.path.join(HERE, "ref_hmm.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_hmm.npz"))


if __name__ == "__main__":
    main()
