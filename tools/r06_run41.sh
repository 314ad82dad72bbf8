#!/bin/bash
# round 6, GPU run 41: a specialised scan kernel is compiled for one loop form (whole zones or split): the scan / parity / golden / plan suites, then the
# default bench line with cold caches (how long the compiler takes now)
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run41
mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_scan_split.py tests/test_gpu_parity.py tests/test_gpu_zones.py tests/test_gpu_z_golden.py tests/test_gpu_sf1_oracle.py tests/test_gpu_prepared.py tests/test_gpu_jit_async.py -m gpu -q -x -n 4 --dist loadfile > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
rm -rf ~/.cache/ldb_jit ~/.cache/comgr
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run41/bench_default.json").read().strip().splitlines()[-1])
print("geomean", d["value"], d["ms_per_step"], {k: (v.get("equal") if isinstance(v, dict) and "equal" in v else v) for k, v in d["checks"].items() if "at_bench" in k or k.endswith("_all")})
print(d["per_query_ms"])
print({k: d["jit"][k] for k in ("compiled", "compile_ms_total", "wait_after_first_pass_s", "warmup_passes_run")})
PY
