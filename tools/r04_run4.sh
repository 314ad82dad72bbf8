#!/bin/bash
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r04_run4
mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_dict.py tests/test_gpu_prepared.py tests/test_gpu_z_golden.py tests/test_gpu_sf1_oracle.py tests/test_gpu_tpch_new.py tests/test_gpu_dist.py -m gpu -q --maxfail=6 > $OUT/tests_full.log 2>&1
tail -12 $OUT/tests_full.log
grep -n "Error\|error\|assert" $OUT/tests_full.log | head -20
timeout 600 python bench.py --steps 5 --warmup 3 --cpu-sample-sf 0 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_run4/bench.json'))
print(d['value'], d['ms_per_step'], d['kernel_share'])
print(d['per_query_ms'])
PY
