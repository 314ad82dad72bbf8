#!/bin/bash
# round 6, GPU run 34: scans of short inputs cut finer (gridDim.y), a LIKE specialised from 64 K rows on, the LDS key filter at the finest granularity
# that fits 16 KB: all 22 with the row-id checks, the scan / join / parity / plan suites, each option on and off, a bench line
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run34
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0"
LDB_DEBUG_CHECK=1 LDB_JIT_ASYNC=0 timeout 900 $B --oracle-spot-check 0 --steps 2 > $OUT/b_dbg.json 2> $OUT/b_dbg.err; echo "all 22 with debug_check rc=$?"; tail -1 $OUT/b_dbg.err | cut -c1-300
timeout 2400 python -m pytest tests/test_gpu_scan_split.py tests/test_gpu_joins_more.py tests/test_gpu_parity.py tests/test_gpu_prepared.py tests/test_gpu_sf1_oracle.py tests/test_gpu_plans_json.py tests/test_gpu_tpch_more.py tests/test_gpu_tpch_new.py tests/test_gpu_z_golden.py tests/test_gpu_zones.py -m gpu -q -x -n 4 --dist loadfile > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
for cfg in "1 1" "0 1" "1 0"; do
  set -- $cfg
  LDB_SCAN_SPLIT=$1 LDB_JOIN_COARSE_FINEST=$2 timeout 700 $B --oracle-spot-check 0 --steps 5 > $OUT/b_$1_$2.json 2> $OUT/b_$1_$2.err
  python - "$OUT/b_$1_$2.json" "$cfg" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("scan_split, coarse_finest", sys.argv[2], d["value"], d["ms_per_step"], d["per_query_ms"])
PY
done
timeout 1200 $B --steps 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run34/bench.json").read().strip().splitlines()[-1])
print("geomean", d["value"], d["ms_per_step"], {k: (v.get("equal") if isinstance(v, dict) else v) for k, v in d["checks"].items() if "at_bench" in k or k.endswith("_all")})
PY
