#!/bin/bash
ulimit -c 0
OUT=$PWD/gpurun_out/r04_run14
mkdir -p $OUT
timeout 40 python tools/subop_prepared_check.py --sf 10 --queries 18,13 --runs 4 > $OUT/prepared.log 2>&1; tail -12 $OUT/prepared.log
