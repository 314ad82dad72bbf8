#!/bin/bash
# round 6, GPU run 44: the last tree (shared compilation claims in the JIT): smoke, the JIT + prepared-plan tests, the default bench line with cold caches
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run44
mkdir -p $OUT
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python -m pytest tests/test_gpu_jit_async.py tests/test_gpu_prepared.py -m gpu -q -x -n 4 --dist loadfile > $OUT/tests.log 2>&1; tail -1 $OUT/tests.log
rm -rf ~/.cache/ldb_jit ~/.cache/comgr
T0=$(date +%s)
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run44/bench_default.json").read().strip().splitlines()[-1])
print("geomean", d["value"], d["ms_per_step"], {k: (v.get("equal") if isinstance(v, dict) and "equal" in v else v) for k, v in d["checks"].items() if "at_bench" in k or k.endswith("_all")})
print(d["per_query_ms"])
print(d["jit"])
PY
ls ~/.cache/ldb_jit/*/ | grep -c lock
