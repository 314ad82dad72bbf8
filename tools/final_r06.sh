#!/bin/bash
# Round-6 closing measurements on the GPU box (outputs under gpurun_out/final_r06, the summaries are copied to profiles/ by hand):
#   1. the default bench line (cold JIT cache; reference-object CPU legs + sliced oracle checks at SF100)   2. a second process start (disk cache warm)
#   3. rocprofv3 --kernel-trace --stats of the same workload (short)      4. FETCH_SIZE / WRITE_SIZE passes for Q1, Q6 and Q9 (separate runs)
#   5. the per-query timeline
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/final_r06
mkdir -p $OUT
rm -rf ~/.cache/ldb_jit ~/.cache/comgr
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 1200 $OUT/bench_default.json; echo; tail -2 $OUT/bench_default.err
B="python $R/bench.py --steps 3 --warmup 3 --cpu-sample-sf 0 --oracle-spot-check 0 --record-runs 0 --cpu-reference-legs 0"
timeout 600 $B > $OUT/bench_second_start.json 2> $OUT/bench_second_start.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats_bench.json 2> $OUT/stats_bench.err
for q in 1 6 9; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_q${q}_$c -- $B --queries $q > $OUT/pmc_q${q}_$c.json 2> $OUT/pmc_q${q}_$c.err
  done
done
cd $R
for q in 1 6 9; do
  F=$(ls $OUT/pmc_q${q}_FETCH_SIZE/*/*counter_collection.csv | head -1)
  W=$(ls $OUT/pmc_q${q}_WRITE_SIZE/*/*counter_collection.csv | head -1)
  python tools/pmc_summary.py --fetch $F --write $W --calib-kernel k_scan_count_spec --calib-bytes 9600000000,2400000000 --out $OUT/r06_pmc_q${q}_sf100.json
done
for d in $OUT/stats/*; do cp $d/*kernel_stats.csv $OUT/r06_kernel_stats_sf100_default.csv 2>/dev/null; cp $d/*agent_info.csv $OUT/r06_agent_info.csv 2>/dev/null; done
bash tools/r04_timeline.sh final_r06/tl > $OUT/tl.log 2>&1; tail -1 $OUT/tl.log
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -size +20M -delete
ls $OUT
