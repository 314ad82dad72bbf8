#!/bin/bash
# round 5, GPU run 11: six workgroups per CU for the sorted group-by — its tests and Q18 / Q1 once more
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r05_run11
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=3 -k "sorted_keys or run_combining" > $OUT/tests_a.log 2>&1
tail -2 $OUT/tests_a.log
timeout 300 python bench.py --queries 18,1 --steps 5 --warmup 3 --cpu-sample-sf 0 --oracle-spot-check 0 --record-runs 0 > $OUT/bench_q18_q1.json 2> $OUT/bench.err
tail -c 300 $OUT/bench_q18_q1.json; echo
