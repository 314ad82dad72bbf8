#!/bin/bash
ulimit -c 0
OUT=$PWD/gpurun_out/r04_run12
mkdir -p $OUT
timeout 100 python tools/subop_prepared_check.py --sf 10 --queries 18,13,16,9,21 > $OUT/prepared.log 2>&1; tail -20 $OUT/prepared.log
