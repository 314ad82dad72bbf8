#!/bin/bash
# Round-3 closing run on the GPU box: per-query timeline → profiles/, the driver's two checks (GPU suite, smoke),
# the default bench line, then the rocprofv3 summaries of tools/profile_r03.sh.  Outputs under gpurun_out/final_r03.
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/final_r03
mkdir -p $OUT
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python $R/tools/query_timeline.py --sf 100 > $OUT/tl.log 2>&1
cd $R
T=$(ls $OUT/tl/*/*kernel_trace.csv | head -1)
python tools/timeline_summary.py $T --out $OUT/r03_query_timeline_sf100.json --top 40 | head -1
rm -rf $OUT/tl
cp $OUT/r03_query_timeline_sf100.json profiles/r03_query_timeline_sf100.json
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -40 > $OUT/tests.log
tail -4 $OUT/tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
head -c 250 $OUT/bench_default.json; echo
timeout 200 python tools/zone_bench.py 100 > $OUT/r03_zone_bench_sf100.json 2> $OUT/zone_bench.err; tail -2 $OUT/zone_bench.err
bash tools/profile_r03.sh > $OUT/profile.log 2>&1
tail -3 $OUT/profile.log
