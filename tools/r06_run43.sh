#!/bin/bash
# round 6, GPU run 43: two ranks on the one GPU with a cold JIT cache — the ranks share the compiler work through claims in the disk cache
# (jit.taken_from_peer_processes on rank 0), once with the sharing on and once off; the JIT tests
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run43
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_jit_async.py -m gpu -q -x > $OUT/tests.log 2>&1; tail -1 $OUT/tests.log
for share in 1 0; do
  rm -rf ~/.cache/ldb_jit
  T0=$(date +%s)
  LDB_JIT_SHARE_COMPILES=$share LDB_DIST_BACKEND=gloo LDB_COMM=shm timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --sf 10 --steps 3 --warmup 2 --cpu-sample-sf 0 --oracle-spot-check 0 > $OUT/bench_share$share.json 2> $OUT/bench_share$share.err; echo "share=$share rc=$? wall $(( $(date +%s) - T0 )) s"
  python - $share <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r06_run43/bench_share%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: v for k, v in d["jit"].items() if k in ("compiled", "disk_hits", "disk_writes", "taken_from_peer_processes", "compile_ms_total", "wait_after_first_pass_s", "warmup_passes_run")})
PY
done
ls ~/.cache/ldb_jit/*/ | grep -c lock
