#!/bin/bash
# round 6, GPU call 53: the reference's own loop (sample_poses.py:67-74), unchanged, around the drop-in class: time per iteration
set -u
OUT=gpurun_out/r6_53
mkdir -p $OUT
timeout 600 python tools/bench_dropin_loop.py lrelu > $OUT/dropin_loop.jsonl 2> $OUT/dropin_loop.err
echo "rc=$?"; cat $OUT/dropin_loop.jsonl; tail -3 $OUT/dropin_loop.err
