#!/bin/bash
# round 5, GPU run 7: LIKE phase A on 32-bit windows — the LIKE tests (oracle, reference StringRuntime cases, SF1 plans), then a short bench
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r05_run7
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_z_golden.py tests/test_gpu_dict.py -m gpu -q --maxfail=3 -k "like" > $OUT/tests_a.log 2>&1
tail -3 $OUT/tests_a.log
timeout 300 python -m pytest tests/test_gpu_sf1_oracle.py -m gpu -q --maxfail=4 -k "test_plan_matches_oracle" > $OUT/tests_b.log 2>&1
tail -3 $OUT/tests_b.log
timeout 600 python bench.py --steps 5 --warmup 3 --cpu-sample-sf 0 --oracle-spot-check 0 --record-runs 0 > $OUT/bench.json 2> $OUT/bench.err
tail -c 700 $OUT/bench.json; echo
