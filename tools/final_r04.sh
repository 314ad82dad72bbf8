#!/bin/bash
# Round-4 closing measurements on the GPU box (outputs under gpurun_out/final_r04, the summaries are copied to profiles/ by hand):
#   1. the default bench line                                   2. rocprofv3 --kernel-trace --stats of the same command (short)
#   3. FETCH_SIZE / WRITE_SIZE passes for Q1 and Q6 (separate)  4. the per-query timeline
#   5. the radix sweep (g1) + FETCH / WRITE passes over the partitioned probe's kernels
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/final_r04
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity.py -q -k groupby > $OUT/tests_groupby.log 2>&1; tail -3 $OUT/tests_groupby.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
head -c 200 $OUT/bench_default.json; echo; tail -2 $OUT/bench_default.err
B="python $R/bench.py --steps 3 --warmup 3 --cpu-sample-sf 0"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats_bench.json 2> $OUT/stats_bench.err
for q in 1 6; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_q${q}_$c -- $B --queries $q > $OUT/pmc_q${q}_$c.json 2> $OUT/pmc_q${q}_$c.err
  done
done
cd $R
for q in 1 6; do
  F=$(ls $OUT/pmc_q${q}_FETCH_SIZE/*/*counter_collection.csv | head -1)
  W=$(ls $OUT/pmc_q${q}_WRITE_SIZE/*/*counter_collection.csv | head -1)
  python tools/pmc_summary.py --fetch $F --write $W --calib-kernel k_scan_count_spec --calib-bytes 9600000000,2400000000 --out $OUT/r04_pmc_q${q}_sf100.json
done
for d in $OUT/stats/*; do cp $d/*kernel_stats.csv $OUT/r04_kernel_stats_sf100_default.csv 2>/dev/null; cp $d/*agent_info.csv $OUT/r04_agent_info.csv 2>/dev/null; done
bash tools/r04_timeline.sh final_r04/tl > $OUT/tl.log 2>&1; tail -1 $OUT/tl.log
timeout 400 python tools/radix_sweep.py 100 1 > $OUT/r04_radix_sweep_sf100.json 2> $OUT/radix_sweep.err; tail -1 $OUT/radix_sweep.err
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_radix_$c -- python $R/tools/radix_pmc.py 100 > $OUT/pmc_radix_$c.log 2>&1
  F=$(ls $OUT/pmc_radix_$c/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$F" ] && python $R/tools/pmc_counters.py $F k_wc_hist k_wc_scatter k_join_probe_lds k_join_probe_count > $OUT/r04_pmc_radix_$c.json
done
LDB_JOIN_RADIX=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_direct_FETCH -- python $R/tools/radix_pmc.py 100 > $OUT/pmc_direct.log 2>&1
F=$(ls $OUT/pmc_direct_FETCH/*/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$F" ] && python $R/tools/pmc_counters.py $F k_join_probe_count > $OUT/r04_pmc_direct_probe_FETCH_SIZE.json
cd $R
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -size +20M -delete
ls $OUT
