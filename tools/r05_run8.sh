#!/bin/bash
# round 5, GPU run 8: the sorted group-by writes its key column itself — tests, SF1 plans, short bench
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r05_run8
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=3 -k "sorted_keys or run_combining or min_max_over_128 or direct_address" > $OUT/tests_a.log 2>&1
tail -3 $OUT/tests_a.log
timeout 300 python -m pytest tests/test_gpu_sf1_oracle.py tests/test_gpu_prepared.py -m gpu -q --maxfail=4 -k "test_plan_matches_oracle or prepared or replay or changed" > $OUT/tests_b.log 2>&1
tail -3 $OUT/tests_b.log
timeout 600 python bench.py --steps 5 --warmup 3 --cpu-sample-sf 0 --oracle-spot-check 0 --record-runs 0 > $OUT/bench.json 2> $OUT/bench.err
tail -c 700 $OUT/bench.json; echo
