#!/bin/bash
# round 4, GPU run 2: arena / deferred log / fused compose / packed export / gb occupancy scan under the parity suites, bench, timeline
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r04_run2
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_prepared.py tests/test_gpu_joins_more.py tests/test_gpu_parity.py tests/test_gpu_f4.py tests/test_gpu_sf1_oracle.py tests/test_gpu_tpch_more.py tests/test_gpu_z_golden.py tests/test_gpu_plans_json.py tests/test_gpu_dict.py -m gpu -q --maxfail=8 2>&1 | tail -40 > $OUT/tests.log
tail -6 $OUT/tests.log
timeout 600 python bench.py --steps 5 --warmup 3 --cpu-sample-sf 0 > $OUT/bench_prepared.json 2> $OUT/bench_prepared.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_run2/bench_prepared.json'))
print(d['value'], d['ms_per_step'], d['kernel_share'])
print(d['per_query_ms'])
print(d['prepared_plans'])
PY
tail -3 $OUT/bench_prepared.err
bash tools/r04_timeline.sh r04_tl2
