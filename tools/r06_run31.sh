#!/bin/bash
# round 6, GPU run 31: which filtered probes gain from applying the filter first (scan kernel + zone maps) — all 22 with fusing off (0), up to one conjunct (1), default (3)
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run31
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0 --oracle-spot-check 0"
for f in 0 1 3; do
  LDB_JOIN_FUSE_MAX_CONJUNCTS=$f timeout 700 $B --steps 5 > $OUT/b_f$f.json 2> $OUT/b_f$f.err
  python - "$OUT/b_f$f.json" $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("fuse at most", sys.argv[2], d["value"], d["ms_per_step"], d["per_query_ms"])
PY
done
