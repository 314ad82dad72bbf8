#!/bin/bash
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r06_run15
mkdir -p $OUT
for sf in 10 100; do
  timeout 600 python tools/r06_dbg_q10.py $sf 10,3,4,21 > $OUT/dbg_sf$sf.log 2>&1; tail -12 $OUT/dbg_sf$sf.log
done
