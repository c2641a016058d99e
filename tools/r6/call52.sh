#!/bin/bash
# round 6, GPU call 52: rocprofv3 kernel stats and PMC passes of the plain-bf16 comparison kernel (the "bf16" half of configs[2])
set -u
OUT=gpurun_out/r6_bf16
mkdir -p $OUT
bash tools/gpu_profile.sh $OUT/prof bf16 lrelu > $OUT/prof.log 2>&1
echo "profile rc=$?"
bash tools/collect_profiles.sh $OUT/prof $OUT/collected pndf_fused_bf16_relu_kernel "bench.py --precision bf16 --act lrelu (B=65536 x 100 steps)"
cp profiles/traffic.json $OUT/traffic.json
python tools/power_window.py bf16:lrelu > $OUT/power_window.txt 2> $OUT/power_window.err; tail -2 $OUT/power_window.txt | cut -c1-400
