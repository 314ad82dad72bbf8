#!/bin/bash
# round 6: the same three counter passes over Q13 and Q9 after the 16-key LDS filter (and with it off, LDB_JOIN_COARSE_FINE=0)  Three separate counter passes (wave stalls, L2 requests, LDS) over the two queries.
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_pmc_like_exists
mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python $R/bench.py --queries 13,9 --steps 2 --warmup 3 --cpu-sample-sf 0 --oracle-spot-check 0 --record-runs 0 --cpu-reference-legs 0 > $OUT/p$i.json 2> $OUT/p$i.err
  F=$(ls $OUT/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$F" ]; then python $R/tools/pmc_counters.py $F k_scan_bitmap k_join_probe_exists k_join_probe_unique k_groupby > $OUT/r06_pmc_q13_q9_set$i.json; rm -rf $OUT/p$i; else tail -3 $OUT/p$i.err; fi
  head -c 1500 $OUT/r06_pmc_q13_q9_set$i.json 2>/dev/null
done

i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  LDB_JOIN_COARSE_FINE=0 timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/q$i -- python $R/bench.py --queries 9 --steps 2 --warmup 3 --cpu-sample-sf 0 --oracle-spot-check 0 --record-runs 0 --cpu-reference-legs 0 > $OUT/q$i.json 2> $OUT/q$i.err
  F=$(ls $OUT/q$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$F" ]; then python $R/tools/pmc_counters.py $F k_join_probe_exists > $OUT/r06_pmc_q9_no_fine_filter_set$i.json; rm -rf $OUT/q$i; else tail -3 $OUT/q$i.err; fi
  head -c 800 $OUT/r06_pmc_q9_no_fine_filter_set$i.json 2>/dev/null
done
