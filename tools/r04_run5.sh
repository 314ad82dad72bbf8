#!/bin/bash
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r04_run5
mkdir -p $OUT
timeout 900 python -m pytest "tests/test_gpu_parity.py::test_groupby_partitioned_lds_count" tests/test_gpu_prepared.py::test_radix_probe_over_the_write_combining_partition tests/test_gpu_joins_more.py tests/test_gpu_z_stress_probe.py -m gpu -q --maxfail=6 > $OUT/tests.log 2>&1
tail -8 $OUT/tests.log
timeout 600 python tools/radix_sweep.py 100 > $OUT/radix_sweep.json 2> $OUT/radix_sweep.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_run5/radix_sweep.json'))
for r in d['runs']:
    print(r['table'], r.get('radix'), r.get('write_combining'), r.get('partitions'), r.get('passes'), r.get('part_bytes'), r['total_ms'], r['kernels_ms'])
PY
tail -3 $OUT/radix_sweep.err
timeout 600 python bench.py --steps 5 --warmup 3 --cpu-sample-sf 0 --queries 13,15,9,16 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_run5/bench.json'))
print(d['per_query_ms'])
print({k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('Q13')})
PY
