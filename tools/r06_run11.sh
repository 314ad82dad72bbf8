#!/bin/bash
# round 6, GPU run 11: the filter granularity chosen from the pass fractions (fine also where the coarse one passes > 25 %: Q8, Q20) against the tree before
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run11
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_joins_more.py -m gpu -q -x -k "fine_lds or every_layout or selective" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
timeout 900 python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0 --steps 5 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run11/bench.json").read().strip().splitlines()[-1])
print("geomean", d["value"], d["ms_per_step"], d["per_query_ms"])
print({k: v for k, v in d["kernel_ms_per_step"].items() if "exists" in k}, {k: (v.get("equal") if isinstance(v, dict) else v) for k, v in d["checks"].items() if "at_bench" in k})
PY
