#!/bin/bash
# round 6, GPU call 29: the whole -m gpu suite on the build with the v13 runtime-planned kernels
set -u
OUT=gpurun_out/r6_29
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.txt 2>&1
echo "gpu suite rc=$?"; tail -14 $OUT/pytest_gpu.txt
