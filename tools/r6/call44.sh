#!/bin/bash
# round 6, GPU call 44: a second randomised sweep of the runtime-planned kernels (300 networks, another seed)
set -u
OUT=gpurun_out/r6_44
mkdir -p $OUT
timeout 2400 python tools/sweep_generic.py 300 11 > $OUT/sweep_generic.txt 2>&1
echo "rc=$?"; grep -c ": ok" $OUT/sweep_generic.txt; grep -E "FAIL|sweep_generic:" $OUT/sweep_generic.txt | grep -v "^\[pose_gate\|^\[gate" | head -20; grep -c "escalated" $OUT/sweep_generic.txt
