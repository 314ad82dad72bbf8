#!/bin/bash
# round 5, GPU run 2: the new group-by / join paths under test, the dist replay check with its whole log, a traced bench
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r05_run2
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_joins_more.py tests/test_gpu_prepared.py -m gpu -q -x > $OUT/tests_a.log 2>&1
tail -5 $OUT/tests_a.log
timeout 900 python -m pytest tests/test_gpu_sf1_oracle.py -m gpu -q --maxfail=4 > $OUT/tests_b.log 2>&1
tail -5 $OUT/tests_b.log
timeout 900 python -m pytest "tests/test_gpu_dist.py::test_sharded_plans_match_single_gpu" -m gpu -q -x > $OUT/tests_c.log 2>&1
tail -5 $OUT/tests_c.log
grep -h "replayed executions\|forced divergence" $OUT/tests_c.log | head
LDB_HOST_TRACE=0.25 timeout 900 python bench.py --steps 5 --warmup 3 --cpu-sample-sf 0 --oracle-spot-check 0 > $OUT/bench.json 2> $OUT/bench.err
tail -c 2500 $OUT/bench.json; echo
grep -c "ldb host" $OUT/bench.err; grep "differs" $OUT/bench.err | sort | uniq -c | sort -rn | head -20
