#!/bin/bash
# round 5, GPU run 10: resident workgroups per CU for the sorted group-by (Q18): LDB_GB_WGS_PER_CU = default (4), 2, 3, 6, 8
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r05_run10
mkdir -p $OUT
for w in 0 2 3 6 8; do
  LDB_GB_WGS_PER_CU=$w timeout 200 python bench.py --queries 18 --steps 5 --warmup 3 --cpu-sample-sf 0 --oracle-spot-check 0 --record-runs 0 > $OUT/q18_wgs$w.json 2> $OUT/q18_wgs$w.err
  python - <<PY
import json
b=json.loads(open("$OUT/q18_wgs$w.json").read().strip().splitlines()[-1])
k=b["kernel_ms_per_step"]
print("wgs_per_cu=$w", "Q18", b["per_query_ms"]["Q18"], {n.split(":")[1]:v for n,v in k.items() if v>0.3})
PY
done
