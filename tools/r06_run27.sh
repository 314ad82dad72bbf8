#!/bin/bash
# round 6, GPU run 27: compaction tile shapes — floor of tiles 2 048 / 1 024 / 512, at most 8 / 16 words per thread — over the queries that compact most
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run27
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0 --oracle-spot-check 0"
for cfg in "1 8" "1 16" "1024 8" "1024 16" "512 16" "256 16"; do
  set -- $cfg
  LDB_COMPACT_WIDE_TILES=$1 LDB_COMPACT_MAX_WORDS=$2 timeout 700 $B --queries 3,5,7,9,10,18,20,21 --steps 5 > $OUT/b_$1_$2.json 2> $OUT/b_$1_$2.err
  python - "$OUT/b_$1_$2.json" "$cfg" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernel_ms_per_step"]
print("floor, max words", sys.argv[2], d["value"], d["ms_per_step"], "compact total", round(sum(v for n, v in k.items() if "bitmap_compact" in n), 3), {n.split(":")[0]: v for n, v in k.items() if "bitmap_compact" in n and v > 0.2})
PY
done
