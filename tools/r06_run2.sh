#!/bin/bash
# round 6, GPU run 2: asynchronous JIT + disk cache (new test, first_execution_ms cold and on a second process start), the collective-replay
# divergence test, the translated dumps with the emitter manifest, the default bench line
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run2
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_jit_async.py tests/test_gpu_dist.py tests/test_gpu_sf1_oracle.py tests/test_gpu_prepared.py -m gpu -q -x > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
rm -rf ~/.cache/ldb_jit ~/.cache/comgr
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 2600 $OUT/bench_default.json; echo; tail -3 $OUT/bench_default.err
timeout 600 python bench.py --steps 3 --cpu-sample-sf 0 --oracle-spot-check 0 --record-runs 0 > $OUT/bench_second_start.json 2> $OUT/bench_second_start.err
python - <<'PY'
import json
for f in ("bench_default.json", "bench_second_start.json"):
    try:
        d = json.loads(open("gpurun_out/r06_run2/" + f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["jit"], "first max", max(d["first_execution_ms"].values()), "sum", sum(d["first_execution_ms"].values()))
    except Exception as e:
        print(f, "unreadable", e)
PY
du -sh ~/.cache/ldb_jit | cat
