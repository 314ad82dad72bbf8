#!/bin/bash
# round 4, GPU run 3: new Q9 plan, lazy dictionary strings, loop / nested_map, world-8 / skew / stress exchange; bench; timeline
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r04_run3
mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_f4.py tests/test_gpu_dict.py tests/test_gpu_prepared.py tests/test_gpu_sf1_oracle.py tests/test_gpu_tpch_new.py tests/test_gpu_dist.py -m gpu -q --maxfail=10 2>&1 | tail -60 > $OUT/tests.log
tail -8 $OUT/tests.log
timeout 600 python bench.py --steps 5 --warmup 3 --cpu-sample-sf 0 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_run3/bench.json'))
print(d['value'], d['ms_per_step'], d['kernel_share'])
print(d['per_query_ms'])
print({k:v for k,v in d['prepared_plans'].items() if not k.startswith('host_')})
print(d['checks']['checksum'])
PY
tail -3 $OUT/bench.err
bash tools/r04_timeline.sh r04_tl3
