#!/bin/bash
# Last check of the round on the GPU box: the full GPU suite, smoke, the default bench line (→ profiles/ by hand).
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/verify_r03
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -30 > $OUT/tests.log
tail -4 $OUT/tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
head -c 250 $OUT/bench_default.json; echo
