#!/bin/bash
# round 5, GPU run 5: pair32 (integrated into the probe loop) debug + parity, translated plans at SF100, 8 ranks over shm on the one GPU
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r05_run5
mkdir -p $OUT
timeout 300 python tools/r05_dbg_pair32.py > $OUT/dbg_pair32.log 2>&1
grep -c OK $OUT/dbg_pair32.log; grep DIFFERENT $OUT/dbg_pair32.log | head -5
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=3 -k "two_int32 or composite or fk_pk or null_keys" > $OUT/tests_a.log 2>&1
tail -4 $OUT/tests_a.log
timeout 600 python -m pytest tests/test_gpu_joins_more.py tests/test_gpu_z_golden.py -m gpu -q --maxfail=3 > $OUT/tests_b.log 2>&1
tail -3 $OUT/tests_b.log
timeout 900 python bench.py --plans subop --steps 5 --warmup 3 --cpu-sample-sf 0 --oracle-spot-check 0 > $OUT/bench_subop_sf100.json 2> $OUT/bench_subop.err
tail -c 1200 $OUT/bench_subop_sf100.json; echo; tail -2 $OUT/bench_subop.err
LDB_DIST_BACKEND=gloo LDB_COMM=shm timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --sf 2 --steps 2 --warmup 2 --cpu-sample-sf 0 --oracle-spot-check 0 --record-runs 1 > $OUT/bench_8ranks_shm_sf2.json 2> $OUT/bench_8ranks.err
tail -c 1500 $OUT/bench_8ranks_shm_sf2.json; echo; tail -3 $OUT/bench_8ranks.err
