#!/bin/bash
# Round-5 check on the GPU box: the full GPU suite and smoke (the bench / profile passes are tools/final_r06.sh)
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/verify_r06
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 --durations=15 > $OUT/tests_full.log 2>&1
tail -30 $OUT/tests_full.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
