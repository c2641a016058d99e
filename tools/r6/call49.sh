#!/bin/bash
# round 6, GPU call 49: the maximum-batch test (26,000,003 poses per launch: past 2^31 elements and 2^32 bytes per tensor)
set -u
OUT=gpurun_out/r6_49
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "maximum_batch or large_batch" --durations=8 > $OUT/max_batch.txt 2>&1
echo "rc=$?"; tail -25 $OUT/max_batch.txt | cut -c1-300
