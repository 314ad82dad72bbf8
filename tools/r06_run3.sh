#!/bin/bash
# round 6, GPU run 3: pipelined key-bit existence probe + non-temporal keys, all-match shortcut with shared row-id vectors, LIKE rarest-byte
# prefilter, short top-k select, long IN lists: the parity suites that cover them, then the bench (JIT cache warm from nothing: cold again)
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run3
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_joins_more.py tests/test_gpu_z_golden.py tests/test_gpu_plans_json.py tests/test_gpu_tpch_more.py tests/test_gpu_tpch_new.py tests/test_gpu_z_tpch_q10.py tests/test_gpu_sf1_oracle.py tests/test_gpu_prepared.py tests/test_gpu_f4.py tests/test_gpu_dict.py -m gpu -q -x -n 4 > $OUT/tests.log 2>&1; tail -8 $OUT/tests.log
timeout 1200 python bench.py --cpu-sample-sf 0 > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json; echo; tail -3 $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_run3/bench.json").read().strip().splitlines()[-1])
print("geomean", d["value"], "ms/step", d["ms_per_step"], "checks", {k: (v.get("equal") if isinstance(v, dict) else v) for k, v in d["checks"].items() if k.startswith("oracle_q")})
for k in sorted(d["kernel_ms_per_step"], key=lambda k: -d["kernel_ms_per_step"][k])[:40]:
    print(k, d["kernel_ms_per_step"][k])
PY
