#!/bin/bash
# round 5, GPU run 4: pair32 + residual debug, LIKE pipeline / heads unroll under test, Q7 host trace, bench
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r05_run4
mkdir -p $OUT
timeout 300 python tools/r05_dbg_pair32.py > $OUT/dbg_pair32.log 2>&1
cat $OUT/dbg_pair32.log | tail -20
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=3 -k "like or scan or sorted_keys or run_combining or partitioned" > $OUT/tests_a.log 2>&1
tail -4 $OUT/tests_a.log
timeout 900 python -m pytest tests/test_gpu_sf1_oracle.py -m gpu -q --maxfail=4 -k "test_plan_matches_oracle" > $OUT/tests_b.log 2>&1
tail -3 $OUT/tests_b.log
LDB_HOST_TRACE=0.1 LDB_PLAN_STEP_TRACE=1 timeout 600 python bench.py --steps 2 --warmup 3 --queries 7,12 --cpu-sample-sf 0 --oracle-spot-check 0 --record-runs 0 > $OUT/trace_q7.json 2> $OUT/trace_q7.err
grep "ms host" $OUT/trace_q7.err | tail -52 | awk '{ if ($(NF-2)+0 > 0.08) print }' | tail -30
timeout 900 python bench.py --steps 5 --warmup 3 --cpu-sample-sf 0 --oracle-spot-check 0 > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json; echo
