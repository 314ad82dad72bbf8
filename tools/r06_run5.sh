#!/bin/bash
# round 6, GPU run 5: LIKE through v_mqsad_u32_u8 (parity first: the goldens pin the instruction's semantics), > 2 residual conjuncts, then the full GPU
# suite + smoke, then a short bench
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run5
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_z_golden.py tests/test_gpu_parity.py tests/test_gpu_plans_json.py -m gpu -q -x -k "like or LIKE or residual or q13 or q16 or q9 or q2 or q14 or q20 or string or filter" > $OUT/tests_like.log 2>&1; tail -4 $OUT/tests_like.log
timeout 1200 python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --steps 5 > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run5/bench.json").read().strip().splitlines()[-1])
print("geomean", d["value"], "ms/step", d["ms_per_step"], "checks", {k: (v.get("equal") if isinstance(v, dict) else v) for k, v in d["checks"].items() if "at_bench_scale" in k})
print(d["per_query_ms"])
for k in sorted(d["kernel_ms_per_step"], key=lambda k: -d["kernel_ms_per_step"][k]):
    if "scan_bitmap" in k: print(k, d["kernel_ms_per_step"][k])
PY
timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 --durations=8 > $OUT/tests_full.log 2>&1
tail -14 $OUT/tests_full.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
