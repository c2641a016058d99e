#!/bin/bash
# round 6, GPU call 43: randomised architectures through both forms of the runtime-planned kernels
set -u
OUT=gpurun_out/r6_43
mkdir -p $OUT
timeout 1500 python tools/sweep_generic.py 150 7 > $OUT/sweep_generic.txt 2>&1
echo "rc=$?"; grep -c ": ok" $OUT/sweep_generic.txt; grep -E "FAIL|skipped|sweep_generic:" $OUT/sweep_generic.txt | head -20
