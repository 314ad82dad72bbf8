#!/bin/bash
# Q1's counter-over-algorithmic ratio against the number of resident workgroups: FETCH_SIZE and kernel time of k_groupby_spec
# for gb_wgs_per_cu = 1, 2, 4 (default), 6, 8.  Output: gpurun_out/q1_sweep/r03_q1_occupancy_sweep.txt
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/q1_sweep
mkdir -p $OUT
: > $OUT/r03_q1_occupancy_sweep.txt
for w in 1 2 4 6 8; do
  ( cd /tmp && LDB_GB_WGS_PER_CU=$w timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p$w -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample-sf 0 --queries 1 > $OUT/p$w.json 2> $OUT/p$w.err )
  F=$(ls $OUT/p$w/*/*counter_collection.csv 2>/dev/null | head -1)
  echo "gb_wgs_per_cu=$w (under --pmc FETCH_SIZE)" >> $OUT/r03_q1_occupancy_sweep.txt
  [ -n "$F" ] && python $R/tools/pmc_counters.py $F k_groupby_spec k_scan_count_spec >> $OUT/r03_q1_occupancy_sweep.txt
  LDB_GB_WGS_PER_CU=$w timeout 200 python $R/bench.py --steps 5 --warmup 2 --cpu-sample-sf 0 --queries 1 2> /dev/null | python -c "import json,sys; b=json.loads(sys.stdin.readline()); print('  untraced: avg_kernel_ms', b['roofline']['avg_kernel_ms'], 'frac', b['roofline']['frac'], 'Q1 ms', b['per_query_ms']['Q1'])" >> $OUT/r03_q1_occupancy_sweep.txt
  rm -rf $OUT/p$w
done
cat $OUT/r03_q1_occupancy_sweep.txt
