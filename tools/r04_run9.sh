#!/bin/bash
# f1 patterns on the GPU box: 128-bit MIN / MAX, the pattern dumps (mark, right outer, group join, set operations), bench.py --plans subop at SF10
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r04_run9
mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_parity.py -q -k "min_max_over_128" > $OUT/minmax.log 2>&1; tail -2 $OUT/minmax.log
timeout 600 python -m pytest tests/test_gpu_sf1_oracle.py -q -k "pattern_dumps or nested_loop" > $OUT/patterns.log 2>&1; tail -30 $OUT/patterns.log
timeout 600 python bench.py --plans subop --sf 10 --steps 3 --warmup 2 --cpu-sample-sf 0 > $OUT/bench_subop_sf10.json 2> $OUT/bench_subop_sf10.err; tail -3 $OUT/bench_subop_sf10.err; head -c 300 $OUT/bench_subop_sf10.json; echo
timeout 600 python bench.py --sf 10 --steps 3 --warmup 2 --cpu-sample-sf 0 > $OUT/bench_files_sf10.json 2> $OUT/bench_files_sf10.err; head -c 300 $OUT/bench_files_sf10.json; echo
