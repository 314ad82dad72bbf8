#!/bin/bash
# round 6, GPU run 14: filtered probes take JT_PB queue entries per lane for full chunks and two for the tail: join parity suites, TPC-H suites, short bench
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run14
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_joins_more.py tests/test_gpu_parity.py tests/test_gpu_tpch_more.py tests/test_gpu_tpch_new.py tests/test_gpu_z_tpch_q10.py tests/test_gpu_sf1_oracle.py tests/test_gpu_z_golden.py -m gpu -q -x -n 4 --dist loadfile > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
timeout 900 python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0 --steps 5 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run14/bench.json").read().strip().splitlines()[-1])
print("geomean", d["value"], d["ms_per_step"], d["per_query_ms"])
print({k: v for k, v in d["kernel_ms_per_step"].items() if "probe" in k and v > 0.8}, {k: (v.get("equal") if isinstance(v, dict) else v) for k, v in d["checks"].items() if "at_bench" in k})
PY
