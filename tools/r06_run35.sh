#!/bin/bash
# round 6, GPU run 35: the whole-zone scan keeps its compile-time trip count (Q13): the scan suites, then a bench line
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run35
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0"
timeout 1500 python -m pytest tests/test_gpu_scan_split.py tests/test_gpu_parity.py tests/test_gpu_zones.py tests/test_gpu_z_golden.py -m gpu -q -x -n 4 --dist loadfile > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
timeout 1200 $B --steps 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run35/bench.json").read().strip().splitlines()[-1])
print("geomean", d["value"], d["ms_per_step"], {k: (v.get("equal") if isinstance(v, dict) else v) for k, v in d["checks"].items() if "at_bench" in k or k.endswith("_all")})
print(d["per_query_ms"])
k = d["kernel_ms_per_step"]
print({n: v for n, v in k.items() if "scan_bitmap" in n and v > 0.3})
PY
