#!/bin/bash
# round 6, GPU run 36: the specialisation threshold (jit_min_rows: 4 M) at 256 K and 1 M rows — what the short queries gain, what the compiler pays
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run36
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0 --oracle-spot-check 0"
for f in 262144 1000000; do
  rm -rf ~/.cache/ldb_jit
  T0=$(date +%s); LDB_JIT_MIN_ROWS=$f timeout 1200 $B --steps 5 > $OUT/b_$f.json 2> $OUT/b_$f.err; echo "wall $(( $(date +%s) - T0 )) s"
  python - "$OUT/b_$f.json" $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("jit_min_rows", sys.argv[2], d["value"], d["ms_per_step"], d["per_query_ms"], {k: d["jit"][k] for k in ("compiled", "compile_ms_total", "wait_after_first_pass_s", "warmup_passes_run")})
PY
done
