#!/bin/bash
# round 6, GPU run 26: wide compaction tiles (4 / 8 bitmap words per thread for bitmaps of >= 2 048 such tiles), descriptor guards (LdbDesc), the
# order-dependent miss counter, the micro-benchmarks warmed up past the asynchronous compiler: prepared-plan + join + parity suites, the option on / off
# over all 22 queries, then a bench line
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run26
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0"
LDB_DEBUG_CHECK=1 LDB_JIT_ASYNC=0 timeout 900 $B --oracle-spot-check 0 --steps 2 > $OUT/b_dbg.json 2> $OUT/b_dbg.err; echo "all 22 with debug_check rc=$?"; tail -1 $OUT/b_dbg.err | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_prepared.py tests/test_gpu_joins_more.py tests/test_gpu_parity.py tests/test_gpu_tpch_more.py tests/test_gpu_tpch_new.py tests/test_gpu_sf1_oracle.py tests/test_gpu_z_tpch_q10.py -m gpu -q -x -n 4 --dist loadfile > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
for f in 1 0; do
  LDB_COMPACT_WIDE_TILES=$f timeout 700 $B --oracle-spot-check 0 --steps 5 > $OUT/b_w$f.json 2> $OUT/b_w$f.err
  python - "$OUT/b_w$f.json" $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernel_ms_per_step"]
print("wide tiles", sys.argv[2], d["value"], d["ms_per_step"], "compact total", round(sum(v for n, v in k.items() if "bitmap_compact" in n), 3), {n: v for n, v in k.items() if "bitmap_compact" in n and v > 0.25})
print("   ", d["per_query_ms"])
PY
done
timeout 1200 $B --steps 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run26/bench.json").read().strip().splitlines()[-1])
print("geomean", d["value"], d["ms_per_step"], {k: (v.get("equal") if isinstance(v, dict) else v) for k, v in d["checks"].items() if "at_bench" in k})
print(d["join_probe"].get("probe_ms"), d["join_probe"]["selective"].get("probe_ms"), d["hbm_ceiling"], d["prepared_plans"].get("order_dependent_misses"), d["prepared_plans"]["descriptor_cache"])
PY
