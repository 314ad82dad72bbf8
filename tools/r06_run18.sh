#!/bin/bash
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r06_run18
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0 --oracle-spot-check 0 --steps 2 --queries 1,2,3,4,5,6,7,8,9,10"
LDB_DEBUG_CHECK=1 LDB_JIT_ASYNC=0 timeout 600 $B > $OUT/b_dbg.json 2> $OUT/b_dbg.err; echo "debug_check rc=$?"; tail -1 $OUT/b_dbg.err | cut -c1-400
timeout 600 $B > $OUT/b.json 2> $OUT/b.err; echo "plain rc=$?"; tail -1 $OUT/b.err | cut -c1-300
