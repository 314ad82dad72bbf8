#!/bin/bash
# round 6, GPU run 19: the tree after the revert of the per-chunk probe width: the failing sequence with the row-id checks on, the join suites, a bench line
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r06_run19
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0"
LDB_DEBUG_CHECK=1 LDB_JIT_ASYNC=0 timeout 900 $B --oracle-spot-check 0 --steps 2 > $OUT/b_dbg.json 2> $OUT/b_dbg.err; echo "all 22 with debug_check rc=$?"; tail -1 $OUT/b_dbg.err | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_joins_more.py tests/test_gpu_parity.py -m gpu -q -x -n 4 --dist loadfile > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
timeout 900 $B --steps 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -1 $OUT/bench.err | cut -c1-300
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run19/bench.json").read().strip().splitlines()[-1])
print("geomean", d["value"], d["ms_per_step"], {k: (v.get("equal") if isinstance(v, dict) else v) for k, v in d["checks"].items() if "at_bench" in k})
PY
