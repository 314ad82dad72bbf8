#!/bin/bash
# round 6, GPU run 13: tunables of the filtered probe kernels through LDB_JIT_DEFINES (specialised kernels only; no rebuild): queue entries per lane (JT_PB)
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run13
mkdir -p $OUT
for v in "" "-DJT_PB=8" "-DJT_PB=2" "-DJT_PB=8 -DJT_EXTRA_SYNC=1"; do
  tag=$(echo "base$v" | tr -d ' =-')
  LDB_JIT_DEFINES="$v" LDB_JIT_CACHE_DIR=/tmp/jit_$tag timeout 600 python bench.py --queries 3,10,21,5,7,4,22,12 --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0 --oracle-spot-check 0 --steps 5 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - "$OUT/bench_$tag.json" "$v" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(repr(sys.argv[2]), d["per_query_ms"], {k: v for k, v in d["kernel_ms_per_step"].items() if "probe" in k and v > 0.8})
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
done
