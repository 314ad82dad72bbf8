#!/bin/bash
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r06_run16
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0 --oracle-spot-check 0 --steps 2"
for qs in "10" "9,10" "1,2,3,4,5,6,7,8,9,10"; do
  timeout 600 $B --queries $qs > $OUT/b_$qs.json 2> $OUT/b_$qs.err; echo "queries $qs rc=$?"; tail -1 $OUT/b_$qs.err | cut -c1-300
done
LDB_JIT_ASYNC=0 timeout 600 $B --queries 1,2,3,4,5,6,7,8,9,10 > $OUT/b_sync.json 2> $OUT/b_sync.err; echo "sync jit rc=$?"; tail -1 $OUT/b_sync.err | cut -c1-300
LDB_JOIN_ALL_MATCH=0 timeout 600 $B --queries 1,2,3,4,5,6,7,8,9,10 > $OUT/b_noam.json 2> $OUT/b_noam.err; echo "no all-match rc=$?"; tail -1 $OUT/b_noam.err | cut -c1-300
LDB_JOIN_COARSE_FINE=0 timeout 600 $B --queries 1,2,3,4,5,6,7,8,9,10 > $OUT/b_nofine.json 2> $OUT/b_nofine.err; echo "no fine rc=$?"; tail -1 $OUT/b_nofine.err | cut -c1-300
