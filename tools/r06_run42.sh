#!/bin/bash
# round 6, GPU run 42: the final tree once more — all 22 queries with the row-id checks on, then the full GPU suite + smoke (tools/verify_r06.sh)
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run42
mkdir -p $OUT
B="python bench.py --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0"
LDB_DEBUG_CHECK=1 LDB_JIT_ASYNC=0 timeout 900 $B --oracle-spot-check 0 --steps 2 > $OUT/b_dbg.json 2> $OUT/b_dbg.err; echo "all 22 with debug_check rc=$?"; tail -1 $OUT/b_dbg.err | cut -c1-300
bash tools/verify_r06.sh
