#!/bin/bash
# round 6, GPU run 40: host time per plan step under replay (LDB_PLAN_STEP_TRACE) for the queries with the largest issue times
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run40
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0 --oracle-spot-check 0"
for q in 7 2 20 5 16; do
  LDB_PLAN_STEP_TRACE=1 timeout 600 $B --queries $q --steps 3 > $OUT/b_q$q.json 2> $OUT/b_q$q.err
  grep "^\[ldb plan\]" $OUT/b_q$q.err > $OUT/steps_q$q.txt; wc -l $OUT/steps_q$q.txt
done
