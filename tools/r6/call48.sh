#!/bin/bash
# round 6, GPU call 48: randomised narrow networks (the fused kernels' zero-padded path)
set -u
OUT=gpurun_out/r6_48
mkdir -p $OUT
timeout 2400 python tools/sweep_narrow.py 150 5 > $OUT/sweep_narrow.txt 2>&1
echo "rc=$?"; grep -c ": ok" $OUT/sweep_narrow.txt; grep -E "FAIL|skipped|sweep_narrow:" $OUT/sweep_narrow.txt | cut -c1-400 | head -20
