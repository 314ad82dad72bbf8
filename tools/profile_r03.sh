#!/bin/bash
# Round-3 profile passes (run on the GPU box from the repo root; outputs under gpurun_out/prof_r03).
#   1. per-kernel time of the default bench command (kernel-trace + stats only)
#   2. FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes for Q1 and Q6 (no other trace domains)
ulimit -c 0
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r03
mkdir -p $OUT
B="python $PWD/bench.py --steps 3 --warmup 1 --cpu-sample-sf 0"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats_bench.json 2> $OUT/stats_bench.err
for q in 1 6; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_q${q}_$c -- $B --queries $q > $OUT/pmc_q${q}_$c.json 2> $OUT/pmc_q${q}_$c.err
  done
done
cd $OUT/..
for q in 1 6; do
  F=$(ls $OUT/pmc_q${q}_FETCH_SIZE/*/*counter_collection.csv | head -1)
  W=$(ls $OUT/pmc_q${q}_WRITE_SIZE/*/*counter_collection.csv | head -1)
  python ../tools/pmc_summary.py --fetch $F --write $W --calib-kernel k_scan_count_spec --calib-bytes 9600000000,2400000000 --out $OUT/r03_pmc_q${q}_sf100.json
done
# keep only the summaries (the traces are large)
for d in $OUT/stats/*; do cp $d/*kernel_stats.csv $OUT/r03_kernel_stats_sf100_default.csv 2>/dev/null; cp $d/*agent_info.csv $OUT/r03_agent_info.csv 2>/dev/null; done
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -size +20M -delete
ls -la $OUT
