#!/bin/bash
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r04_run10
mkdir -p $OUT
timeout 120 python -m pytest tests/test_gpu_sf1_oracle.py -q -k "pattern_dumps" > $OUT/patterns.log 2>&1; grep -E "^E|passed|failed" $OUT/patterns.log | head -20
timeout 140 python tools/subop_prepared_check.py --sf 10 > $OUT/prepared.log 2>&1; tail -12 $OUT/prepared.log
