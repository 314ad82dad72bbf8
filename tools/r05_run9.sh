#!/bin/bash
# round 5, GPU run 9 (closing): the modules around the sorted group-by's key column once more (sharded plans, independent Q18 evaluation, f4), then the default bench line
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r05_run9
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_tpch_more.py tests/test_gpu_tpch_new.py tests/test_gpu_f4.py tests/test_gpu_scale.py -m gpu -q --maxfail=5 --durations=5 > $OUT/tests.log 2>&1
tail -12 $OUT/tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 900 $OUT/bench_default.json; echo; tail -2 $OUT/bench_default.err
