#!/bin/bash
# round 4, GPU run 1: the new scan / compaction / prepared-plan paths under test, then the bench with and without prepared plans
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r04_run1
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_prepared.py -m gpu -q -x 2>&1 | tail -30 > $OUT/tests_prepared.log
tail -5 $OUT/tests_prepared.log
timeout 900 python -m pytest tests/test_gpu_joins_more.py tests/test_gpu_parity.py tests/test_gpu_f4.py -m gpu -q --maxfail=5 2>&1 | tail -30 > $OUT/tests_parity.log
tail -5 $OUT/tests_parity.log
LDB_BENCH_PREPARED=0 timeout 600 python bench.py --steps 5 --warmup 3 --cpu-sample-sf 0 > $OUT/bench_unprepared.json 2> $OUT/bench_unprepared.err
head -c 300 $OUT/bench_unprepared.json; echo
timeout 600 python bench.py --steps 5 --warmup 3 --cpu-sample-sf 0 > $OUT/bench_prepared.json 2> $OUT/bench_prepared.err
head -c 300 $OUT/bench_prepared.json; echo
tail -3 $OUT/bench_prepared.err
