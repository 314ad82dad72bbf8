#!/bin/bash
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r06_run17
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0 --oracle-spot-check 0 --steps 2 --queries 1,2,3,4,5,6,7,8,9,10"
LDB_DEBUG_CHECK=1 LDB_JIT_ASYNC=0 timeout 600 $B > $OUT/b_dbg.json 2> $OUT/b_dbg.err; echo "debug_check rc=$?"; tail -2 $OUT/b_dbg.err | cut -c1-500
LDB_JIT_DEFINES="-DJT_PB=2" LDB_JIT_ASYNC=0 LDB_JIT_CACHE_DIR=/tmp/j2 timeout 600 $B > $OUT/b_pb2.json 2> $OUT/b_pb2.err; echo "PB=2 spec rc=$?"; tail -1 $OUT/b_pb2.err | cut -c1-300
LDB_LAZY_FILTER=0 timeout 600 $B > $OUT/b_nolazy.json 2> $OUT/b_nolazy.err; echo "no lazy filter rc=$?"; tail -1 $OUT/b_nolazy.err | cut -c1-300
LDB_PLAN_REPLAY=0 timeout 600 $B > $OUT/b_norep.json 2> $OUT/b_norep.err; echo "no replay rc=$?"; tail -1 $OUT/b_norep.err | cut -c1-300
