#!/bin/bash
# round 6, GPU run 32: Q21's candidate rows find their order through the orders index: debug-checked run, the plan suites (one rank and two), a bench line
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run32
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0"
LDB_DEBUG_CHECK=1 LDB_JIT_ASYNC=0 timeout 900 $B --oracle-spot-check 0 --steps 2 --queries 21 > $OUT/b_dbg.json 2> $OUT/b_dbg.err; echo "Q21 with debug_check rc=$?"; tail -1 $OUT/b_dbg.err | cut -c1-300
timeout 1800 python -m pytest tests/test_gpu_prepared.py tests/test_gpu_sf1_oracle.py tests/test_gpu_plans_json.py tests/test_gpu_dist.py -m gpu -q -x -n 4 --dist loadfile > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
timeout 1200 $B --steps 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run32/bench.json").read().strip().splitlines()[-1])
print("geomean", d["value"], d["ms_per_step"], {k: (v.get("equal") if isinstance(v, dict) else v) for k, v in d["checks"].items() if "at_bench" in k or k.endswith("_all")})
print(d["per_query_ms"])
print({k: v for k, v in d["kernel_ms_per_step"].items() if k.startswith(("Q21:")) and v > 0.1})
PY
