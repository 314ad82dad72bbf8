#!/bin/bash
# round 6, GPU run 29: Q12 probing the orders index (and Q7 / Q8 with the orders side as the unique build (plans/tpch/q7.json, q8.json and their sharded forms): the suites that run the
# plan files (one rank and two), then a bench line
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run29
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0"
LDB_DEBUG_CHECK=1 LDB_JIT_ASYNC=0 timeout 900 $B --oracle-spot-check 0 --steps 2 --queries 7,8,12 > $OUT/b_dbg.json 2> $OUT/b_dbg.err; echo "Q7 Q8 Q12 with debug_check rc=$?"; tail -1 $OUT/b_dbg.err | cut -c1-300
timeout 1800 python -m pytest tests/test_gpu_prepared.py tests/test_gpu_sf1_oracle.py tests/test_gpu_plans_json.py tests/test_gpu_dist.py tests/test_gpu_tpch_more.py -m gpu -q -x -n 4 --dist loadfile > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
timeout 1200 $B --steps 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run29/bench.json").read().strip().splitlines()[-1])
print("geomean", d["value"], d["ms_per_step"], {k: (v.get("equal") if isinstance(v, dict) else v) for k, v in d["checks"].items() if "at_bench" in k or k.endswith("_all")})
print(d["per_query_ms"])
print({k: v for k, v in d["kernel_ms_per_step"].items() if k.startswith(("Q7:", "Q8:", "Q12:")) and v > 0.15})
PY
