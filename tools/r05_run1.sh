#!/bin/bash
# round 5, GPU run 1: host trace of the replayed big plans (what blocks while issuing?), the default bench line, then the GPU suite
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r05_run1
mkdir -p $OUT
LDB_HOST_TRACE=0.25 LDB_PLAN_STEP_TRACE=1 timeout 600 python bench.py --steps 2 --warmup 3 --queries 18,21,7,9,10,3 --cpu-sample-sf 0 --oracle-spot-check 0 --record-runs 0 > $OUT/trace_bench.json 2> $OUT/trace_bench.err
tail -c 1500 $OUT/trace_bench.json; echo
grep -c "ldb host" $OUT/trace_bench.err
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json; echo
tail -3 $OUT/bench.err
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $OUT/tests_gpu.log
tail -8 $OUT/tests_gpu.log
