#!/bin/bash
# round 6, GPU run 1: what the lease looks like (partition modes, host memory), the default bench line with the Q9 / Q18 sliced-oracle checks,
# and BASELINE configs[4] functionally: SF300 Q9 on 8 ranks sharing the one GPU over the shm transport, checked against the sliced oracle
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run1
mkdir -p $OUT
{ echo "== rocm-smi --showcomputepartition"; rocm-smi --showcomputepartition 2>&1 | head -20; echo "== rocm-smi --showmemorypartition"; rocm-smi --showmemorypartition 2>&1 | head -20;
  echo "== amd-smi partition"; timeout 30 amd-smi partition 2>&1 | head -60; echo "== rocminfo agents"; rocminfo | grep -c "Name:.*gfx950";
  echo "== host"; nproc; cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/memory.max; free -g; df -h /dev/shm /tmp | cat; } > $OUT/lease_probe.txt 2>&1
cat $OUT/lease_probe.txt | tail -25
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 1800 $OUT/bench_default.json; echo; tail -3 $OUT/bench_default.err
LDB_DIST_BACKEND=gloo LDB_COMM=shm timeout 2400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 8 --sf 300 --queries 9 --steps 2 --warmup 2 --cpu-sample-sf 0 --record-runs 1 > $OUT/bench_8ranks_shm_sf300_q9.json 2> $OUT/bench_8ranks.err
tail -c 2500 $OUT/bench_8ranks_shm_sf300_q9.json; echo; tail -5 $OUT/bench_8ranks.err
