#!/bin/bash
# round 6, GPU call 33: six more width cases of the runtime-planned kernels (three groups per pass, 4 + 1 / 4 + 3 passes, stream parity)
set -u
OUT=gpurun_out/r6_33
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_depth.py -m gpu -q > $OUT/pytest_depth.txt 2>&1
echo "depth rc=$?"; tail -12 $OUT/pytest_depth.txt
