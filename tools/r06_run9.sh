#!/bin/bash
# round 6, GPU run 9: the fine LDS key filter (one bit per 16 key values, up to 156 KB, one 1024-thread workgroup per CU) — its test, the coarse
# layouts' test, then Q9 / Q5 / Q8 / Q14 / Q17 / Q20 with it on and off
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run9
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_joins_more.py -m gpu -q -x -k "fine_lds or every_layout" > $OUT/tests.log 2>&1; tail -6 $OUT/tests.log
for fine in 1 0; do
  LDB_JOIN_COARSE_FINE=$fine timeout 900 python bench.py --queries 9,5,8,14,17,20 --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0 --steps 5 > $OUT/bench_fine$fine.json 2> $OUT/bench_fine$fine.err
  tail -1 $OUT/bench_fine$fine.err
done
python - <<'PY'
import json
for f in ("bench_fine1.json", "bench_fine0.json"):
    try:
        d = json.loads(open("gpurun_out/r06_run9/" + f).read().strip().splitlines()[-1])
        print(f, d["per_query_ms"], {k: v for k, v in d["kernel_ms_per_step"].items() if "exists" in k}, d["checks"].get("oracle_q9_at_bench_scale"))
    except Exception as e:
        print(f, "unreadable", e)
PY
