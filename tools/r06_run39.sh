#!/bin/bash
# round 6, GPU run 39: the sharded plans of the final tree (Q7 / Q8 / Q12 / Q21 re-ordered) — two ranks sharing the one GPU over shm at SF10, sliced-oracle checks on rank 0
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run39
mkdir -p $OUT
LDB_DIST_BACKEND=gloo LDB_COMM=shm timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --sf 10 --steps 3 --warmup 2 --cpu-sample-sf 0 --oracle-spot-check 2 > $OUT/bench_2ranks_shm_sf10.json 2> $OUT/bench_2ranks.err; echo "rc=$?"; tail -1 $OUT/bench_2ranks.err | cut -c1-300
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run39/bench_2ranks_shm_sf10.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["n_gpus"], d["prepared_plans"]["replays"], d["prepared_plans"]["misses"], {k: (v.get("equal") if isinstance(v, dict) else v) for k, v in d["checks"].items()})
PY
