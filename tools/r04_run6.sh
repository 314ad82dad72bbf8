#!/bin/bash
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r04_run7
mkdir -p $OUT
timeout 600 python -m pytest "tests/test_gpu_parity.py::test_groupby_partitioned_lds_count" tests/test_gpu_prepared.py::test_radix_probe_over_the_write_combining_partition -m gpu -q --maxfail=3 > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
timeout 600 python tools/radix_sweep.py 100 1 > $OUT/radix_sweep.json 2> $OUT/radix_sweep.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_run7/radix_sweep.json'))
for r in d['runs']:
    print(r['table'], r.get('radix'), r.get('write_combining'), r.get('partitions'), r.get('passes'), r.get('part_bytes'), r['total_ms'], r['kernels_ms'])
PY
tail -3 $OUT/radix_sweep.err
