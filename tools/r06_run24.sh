#!/bin/bash
# round 6, GPU run 24: the small LDS key filter in front of the filtered (tile) probes: all 22 queries with the row-id checks on, the join + parity + TPC-H
# suites, then Q21 / Q19 / Q7 / Q5 with the option on and off, then a bench line
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run24
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0"
LDB_DEBUG_CHECK=1 LDB_JIT_ASYNC=0 timeout 900 $B --oracle-spot-check 0 --steps 2 > $OUT/b_dbg.json 2> $OUT/b_dbg.err; echo "all 22 with debug_check rc=$?"; tail -1 $OUT/b_dbg.err | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_joins_more.py tests/test_gpu_parity.py tests/test_gpu_tpch_more.py tests/test_gpu_tpch_new.py tests/test_gpu_sf1_oracle.py tests/test_gpu_z_tpch_q10.py -m gpu -q -x -n 4 --dist loadfile > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
for f in 1 0; do
  LDB_JOIN_COARSE_FILTERED=$f timeout 600 $B --queries 21,19,7,5,12,14 --oracle-spot-check 0 --steps 5 > $OUT/b_f$f.json 2> $OUT/b_f$f.err
  python - "$OUT/b_f$f.json" $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("filtered-path filter", sys.argv[2], d["per_query_ms"], {k: v for k, v in d["kernel_ms_per_step"].items() if "probe" in k and v > 0.9})
PY
done
timeout 900 $B --steps 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run24/bench.json").read().strip().splitlines()[-1])
print("geomean", d["value"], d["ms_per_step"], {k: (v.get("equal") if isinstance(v, dict) else v) for k, v in d["checks"].items() if "at_bench" in k})
print(d["per_query_ms"])
PY
