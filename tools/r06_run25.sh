#!/bin/bash
# round 6, GPU run 25: the final tree — full GPU suite, smoke, the default bench line (cold caches)
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run25
mkdir -p $OUT
bash tools/verify_r06.sh
rm -rf ~/.cache/ldb_jit ~/.cache/comgr
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.err | cut -c1-300
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run25/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["jit"]["warmup_passes_run"], {k: (v.get("equal") if isinstance(v, dict) and "equal" in v else v) for k, v in d["checks"].items() if "at_bench" in k or k.endswith("_all") or "error" in k})
print(d["per_query_ms"])
PY
