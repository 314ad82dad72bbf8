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
# Q1's 1.125x counter-over-algorithmic ratio: request-level counters of the same kernel (one set per pass), and the
# same FETCH_SIZE pass with 8-byte decimals (half the stride of the wide columns)
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_MISS_sum TCC_REQ_sum TCC_HIT_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_IO_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_q1_req$i -- $B --queries 1 > $OUT/pmc_q1_req$i.json 2> $OUT/pmc_q1_req$i.err )
  F=$(ls $OUT/pmc_q1_req$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$F" ]; then python ../tools/pmc_counters.py $F k_groupby_spec k_scan_count_spec > $OUT/r03_pmc_q1_requests_$i.json; else tail -2 $OUT/pmc_q1_req$i.err; fi
done
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_q1n_FETCH -- $B --queries 1 --narrow-decimals 1 > $OUT/pmc_q1n.json 2> $OUT/pmc_q1n.err )
F=$(ls $OUT/pmc_q1n_FETCH/*/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$F" ] && python ../tools/pmc_counters.py $F k_groupby_spec k_scan_count_spec > $OUT/r03_pmc_q1_narrow_fetch.json
# keep only the summaries (the traces are large)
for d in $OUT/stats/*; do cp $d/*kernel_stats.csv $OUT/r03_kernel_stats_sf100_default.csv 2>/dev/null; cp $d/*agent_info.csv $OUT/r03_agent_info.csv 2>/dev/null; done
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -size +20M -delete
ls -la $OUT
