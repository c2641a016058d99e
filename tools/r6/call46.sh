#!/bin/bash
# round 6, GPU call 46: do the FUSED split kernels lose a pose's gradient behind a one-unit layer (narrow networks, zero padded)?
set -u
OUT=gpurun_out/r6_46
mkdir -p $OUT
timeout 900 python tools/r6/fused_narrow.py > $OUT/fused_narrow.txt 2>&1
cat $OUT/fused_narrow.txt | tail -30
