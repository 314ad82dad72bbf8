#!/bin/bash
# round 6, GPU run 21: two side lines on the final tree — the translated sub-operator dumps (with the manifest) as the bench's plans, and two ranks over shm at SF10
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run21
mkdir -p $OUT
timeout 900 python bench.py --plans subop --steps 5 --warmup 3 --cpu-sample-sf 0 --cpu-reference-legs 0 --oracle-spot-check 0 > $OUT/bench_subop.json 2> $OUT/bench_subop.err; tail -1 $OUT/bench_subop.err | cut -c1-300
LDB_DIST_BACKEND=gloo LDB_COMM=shm timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --sf 10 --steps 3 --warmup 2 --cpu-sample-sf 0 --oracle-spot-check 2 > $OUT/bench_2ranks_shm_sf10.json 2> $OUT/bench_2ranks.err; tail -1 $OUT/bench_2ranks.err | cut -c1-300
python - <<'PY'
import json
for f in ("bench_subop.json", "bench_2ranks_shm_sf10.json"):
    try:
        d = json.loads(open("gpurun_out/r06_run21/" + f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["n_gpus"], d["prepared_plans"]["replays"], d["prepared_plans"]["misses"], {k: (v.get("equal") if isinstance(v, dict) else v) for k, v in d["checks"].items() if "at_bench" in k})
    except Exception as e:
        print(f, "unreadable", e)
PY
