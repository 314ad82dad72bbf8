#!/bin/bash
# round 6, GPU run 6: the compressed resident format (narrow level 2): its tests, then the bench on it (a labelled side line) — and a short default
# line to see the LIKE scan back at the r5 matcher
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run6
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_narrow2.py -m gpu -q -x > $OUT/tests_narrow2.log 2>&1; tail -15 $OUT/tests_narrow2.log
timeout 1200 python bench.py --narrow-decimals 2 --cpu-sample-sf 10 --cpu-budget-s 60 --steps 5 > $OUT/bench_narrow2.json 2> $OUT/bench_narrow2.err
tail -2 $OUT/bench_narrow2.err
timeout 900 python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --oracle-spot-check 0 --record-runs 0 --steps 5 > $OUT/bench_default_short.json 2> $OUT/bench_default_short.err
python - <<'PY'
import json
for f in ("bench_narrow2.json", "bench_default_short.json"):
    try:
        d = json.loads(open("gpurun_out/r06_run6/" + f).read().strip().splitlines()[-1])
        print(f, "geomean", d["value"], "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_kernel_ms", "bytes_per_row")})
        print("   checks", {k: (v.get("equal") if isinstance(v, dict) and "equal" in v else v) for k, v in d["checks"].items() if "at_bench_scale" in k or k.endswith("_all") or "error" in k})
        print("   ", d["per_query_ms"])
        print("    Q13 scan", d["kernel_ms_per_step"].get("Q13:k_scan_bitmap"))
    except Exception as e:
        print(f, "unreadable", e)
PY
