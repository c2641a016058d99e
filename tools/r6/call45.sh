#!/bin/bash
# round 6, GPU call 45: depth tests + a third randomised sweep (seed 23) on the build with the wider gradient scale
set -u
OUT=gpurun_out/r6_45
mkdir -p $OUT
timeout 900 python -m pytest tests/test_depth.py -m gpu -q > $OUT/pytest_depth.txt 2>&1
echo "depth rc=$?"; grep -E "^FAILED|passed|failed" $OUT/pytest_depth.txt | tail -5
timeout 2400 python tools/sweep_generic.py 300 23 > $OUT/sweep_generic.txt 2>&1
echo "rc=$?"; grep -c ": ok" $OUT/sweep_generic.txt; grep -E "FAIL|sweep_generic:" $OUT/sweep_generic.txt | grep -v "^\[pose_gate\|^\[gate" | cut -c1-420 | head -20
