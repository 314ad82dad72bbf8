#!/bin/bash
# PMC passes over the FK-probe micro-benchmark (tools/probe_bench.py): memory-side bytes, L2 hit rate,
# wave stall breakdown and L1 traffic of k_join_probe_count_spec per variant (clustered / unclustered /
# selective / selective unclustered, 6 dispatches each in that order).  One counter set per pass.
ulimit -c 0
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_probe
mkdir -p $OUT
R=$PWD
cd /tmp
i=0
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python $R/tools/probe_bench.py --sf 100 --reps 5 > $OUT/p$i.json 2> $OUT/p$i.err
  F=$(ls $OUT/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$F" ]; then python $R/tools/pmc_probe_summary.py $F > $OUT/p$i.summary.txt; rm -rf $OUT/p$i; else tail -3 $OUT/p$i.err; fi
  cat $OUT/p$i.summary.txt 2>/dev/null
done
