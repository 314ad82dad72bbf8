#!/bin/bash
# round 6, GPU run 4: LIKE (first segment by position, later segments by row), top-k early stop, reverted rarest-byte prefilter; the default bench with
# the reference-object CPU legs at SF100
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run4
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_z_golden.py tests/test_gpu_plans_json.py tests/test_gpu_tpch_more.py tests/test_gpu_tpch_new.py tests/test_gpu_z_tpch_q10.py tests/test_gpu_sf1_oracle.py tests/test_gpu_prepared.py tests/test_gpu_dict.py -m gpu -q -x -n 4 > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run4/bench.json").read().strip().splitlines()[-1])
print("geomean", d["value"], "ms/step", d["ms_per_step"], "checks", {k: (v.get("equal") if isinstance(v, dict) else v) for k, v in d["checks"].items() if "at_bench_scale" in k or k.endswith("_all") or "error" in k})
c = d["cpu_baseline"]
print("cpu_baseline", {k: c[k] for k in c if k not in ("interpreter_legs", "sample")})
print("interp", c.get("interpreter_legs", {}).get("value"))
print(d["per_query_ms"])
for k in sorted(d["kernel_ms_per_step"], key=lambda k: -d["kernel_ms_per_step"][k])[:24]:
    print(k, d["kernel_ms_per_step"][k])
PY
