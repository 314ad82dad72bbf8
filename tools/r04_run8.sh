#!/bin/bash
# f1 check on the GPU box: the 128-bit MIN / MAX group aggregate, then every dump → translator → interpreter → oracle leg
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r04_run8
mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_parity.py -q -k "min_max_over_128" > $OUT/minmax.log 2>&1; tail -3 $OUT/minmax.log
timeout 900 python -m pytest tests/test_gpu_sf1_oracle.py -q -k "subop_dump or nested_loop" --maxfail=40 > $OUT/dumps.log 2>&1; tail -40 $OUT/dumps.log
