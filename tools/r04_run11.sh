#!/bin/bash
# last check of the round on the GPU box: every GPU module that runs plans / group-bys / joins (the exchange modules and the scale sweep are unchanged since tools/verify_r04.sh)
ulimit -c 0
R=$PWD
OUT=$R/gpurun_out/r04_run11
mkdir -p $OUT
timeout 630 python -m pytest tests -m gpu -q --maxfail=12 --durations=12 --deselect tests/test_gpu_dist.py --deselect tests/test_gpu_comm.py --deselect tests/test_gpu_scale.py --deselect tests/test_gpu_z_stress_probe.py > $OUT/tests.log 2>&1
tail -30 $OUT/tests.log
