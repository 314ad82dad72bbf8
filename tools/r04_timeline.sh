#!/bin/bash
# round 4: per-query kernel timeline of the 22 prepared plans at SF100 (rocprofv3 --kernel-trace, cut at the marker kernels)
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/${1:-r04_tl}
mkdir -p $OUT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/tools/query_timeline.py --sf 100 --runs 3 --warmup 3 > $OUT/run.log 2> $OUT/run.err
cd $R
T=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python tools/timeline_summary.py $T --out $OUT/query_timeline_sf100.json --top 60 > $OUT/summary.log 2>&1
for d in $OUT/trace/*; do cp $d/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; done
find $OUT -name '*kernel_trace.csv' -delete
tail -5 $OUT/summary.log; tail -3 $OUT/run.err
