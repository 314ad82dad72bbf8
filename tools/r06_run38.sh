#!/bin/bash
# round 6, GPU run 38: where the host spends its issue time in Q7 / Q2 / Q20 / Q5 (LDB_HOST_TRACE: host-side calls longer than 20 us)
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run38
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0 --oracle-spot-check 0"
for q in 7 2 20 5; do
  LDB_HOST_TRACE=0.02 timeout 600 $B --queries $q --steps 3 > $OUT/b_q$q.json 2> $OUT/b_q$q.err
  echo "== Q$q"; grep "^\[ldb" $OUT/b_q$q.err | tail -60 | cut -c1-220
done
