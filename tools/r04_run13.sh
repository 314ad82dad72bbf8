#!/bin/bash
ulimit -c 0
OUT=$PWD/gpurun_out/r04_run13
mkdir -p $OUT
timeout 100 python -m pytest tests/test_gpu_sf1_oracle.py -q -k "pattern_dumps" > $OUT/patterns.log 2>&1; grep -E "^E|passed|failed" $OUT/patterns.log | head -30
