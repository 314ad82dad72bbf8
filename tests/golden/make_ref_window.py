#!/usr/bin/env python3
"""Golden answers of the reference's REAL SegmentTreeView (src/runtime/SegmentTreeView.cpp, compiled into oracle/_ref by
oracle/ref_build/build_ref.sh; glue ref_segment_tree) for window frames: one partition of N rows (values with NULLs), the
frames ROWS BETWEEN from AND to of WindowLowering clamped into the partition as OffsetReferenceByLowering does
(SubOpToControlFlow.cpp:3860-3885), SUM / MIN / MAX / COUNT per row.  Writes tests/golden/ref_segtree.npz — the
reference tree does not exist on the GPU box, so its answers travel as this fixture.  Run in the build container:
    python tests/golden/make_ref_window.py"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
I64_MIN, I64_MAX = -(2 ** 63), 2 ** 63 - 1
FRAMES = [(I64_MIN, 0), (I64_MIN, I64_MAX), (0, I64_MAX), (-3, 0), (-2, 2), (0, 5), (1, 4), (-7, -2), (0, 0), (-1000, 1000)]


def frame_bounds(n, frm, to):
    cur = np.arange(n, dtype=np.int64)
    lo = np.zeros(n, np.int64) if frm == I64_MIN else np.clip(cur + frm, 0, n - 1)
    hi = np.full(n, n - 1, np.int64) if to == I64_MAX else np.clip(cur + to, 0, n - 1)
    return lo, hi


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libldb_ref.so"))
    lib.ref_segment_tree.restype = C.c_int32
    lib.ref_segment_tree.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(20260926)
    n = 1537
    vals = rng.integers(-10 ** 12, 10 ** 12, n).astype(np.int64)
    valid = (rng.integers(0, 5, n) > 0).astype(np.uint8)
    valid[100:140] = 0  # a run of NULLs longer than the short frames: SUM / MIN / MAX over it are NULL
    out = {"vals": vals, "valid": valid, "frames": np.array(FRAMES, dtype=np.int64)}
    for fi, (frm, to) in enumerate(FRAMES):
        lo, hi = frame_bounds(n, frm, to)
        keep = lo <= hi  # (an inverted frame makes the reference's lookup throw; none of FRAMES inverts after clamping)
        assert keep.all()
        for fn in (1, 2, 3, 4):
            ov, ok = np.zeros(n, np.int64), np.zeros(n, np.uint8)
            st = lib.ref_segment_tree(vals.ctypes.data, valid.ctypes.data, n, fn, lo.ctypes.data, hi.ctypes.data, n, ov.ctypes.data, ok.ctypes.data)
            assert st == 0
            out["f%d_fn%d_val" % (fi, fn)] = ov
            out["f%d_fn%d_ok" % (fi, fn)] = ok
    np.savez_compressed(os.path.join(HERE, "ref_segtree.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_segtree.npz"))


if __name__ == "__main__":
    main()
