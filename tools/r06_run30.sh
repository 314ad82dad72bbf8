#!/bin/bash
# round 6, GPU run 30: a conjunction of more than three conjuncts is applied by the scan kernel before a probe (Q12): on / off, the plan suites, a bench line
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run30
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0"
for f in 3 8; do
  LDB_JOIN_FUSE_MAX_CONJUNCTS=$f timeout 600 $B --queries 12,3,7 --oracle-spot-check 0 --steps 5 > $OUT/b_f$f.json 2> $OUT/b_f$f.err
  python - "$OUT/b_f$f.json" $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("fuse at most", sys.argv[2], d["per_query_ms"], {k: v for k, v in d["kernel_ms_per_step"].items() if k.startswith("Q12:") and v > 0.1})
PY
done
timeout 1800 python -m pytest tests/test_gpu_prepared.py tests/test_gpu_sf1_oracle.py tests/test_gpu_plans_json.py tests/test_gpu_dist.py tests/test_gpu_tpch_more.py tests/test_gpu_joins_more.py -m gpu -q -x -n 4 --dist loadfile > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
timeout 1200 $B --steps 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run30/bench.json").read().strip().splitlines()[-1])
print("geomean", d["value"], d["ms_per_step"], {k: (v.get("equal") if isinstance(v, dict) else v) for k, v in d["checks"].items() if "at_bench" in k or k.endswith("_all")})
print(d["per_query_ms"])
PY
