#!/bin/bash
# round 5, GPU run 3: partitioned value aggregation, pair32 tables, the fixed sorted-key test; dist replay; a traced bench; the 2-rank bench over shm
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r05_run3
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_joins_more.py -m gpu -q --maxfail=3 > $OUT/tests_a.log 2>&1
tail -6 $OUT/tests_a.log
timeout 900 python -m pytest tests/test_gpu_sf1_oracle.py tests/test_gpu_prepared.py -m gpu -q --maxfail=4 > $OUT/tests_b.log 2>&1
tail -4 $OUT/tests_b.log
timeout 900 python -m pytest "tests/test_gpu_dist.py::test_sharded_plans_match_single_gpu" -m gpu -q -x > $OUT/tests_c.log 2>&1
tail -3 $OUT/tests_c.log
grep -h "replayed executions\|forced divergence" $OUT/tests_c.log | head -4
LDB_HOST_TRACE=0.25 timeout 900 python bench.py --steps 5 --warmup 3 --cpu-sample-sf 0 --oracle-spot-check 0 > $OUT/bench.json 2> $OUT/bench.err
tail -c 1800 $OUT/bench.json; echo
grep "differs" $OUT/bench.err | sort | uniq -c | sort -rn | head -10
LDB_DIST_BACKEND=gloo LDB_COMM=shm timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --sf 10 --steps 3 --warmup 2 --cpu-sample-sf 0 --oracle-spot-check 0 > $OUT/bench_2ranks_shm_sf10.json 2> $OUT/bench_2ranks.err
tail -c 1500 $OUT/bench_2ranks_shm_sf10.json; echo; tail -3 $OUT/bench_2ranks.err
