#!/bin/bash
# round 6, GPU call 34: the runtime-planned kernels on split-precision fp16 MFMAs (precision f16x3), first run: parity, then bench
set -u
OUT=gpurun_out/r6_34
mkdir -p $OUT
timeout 900 python -m pytest tests/test_depth.py -m gpu -q > $OUT/pytest_depth.txt 2>&1
echo "depth rc=$?"; grep -E "^FAILED|passed|failed" $OUT/pytest_depth.txt | tail -30
timeout 900 python tools/bench_generic.py 1 10 11 12 13 14 15 > $OUT/generic_arch.jsonl 2> $OUT/generic_arch.err
python - <<'PY'
import json
for l in open('gpurun_out/r6_34/generic_arch.jsonl'):
    d=json.loads(l)
    print(d.get('arm'), d.get('kernel'), round(d.get('ms',0),2), 'ms', round(d.get('frac_of_fp32_mfma_peak', d.get('frac_of_fp16_mfma_peak', 0)),3), round(d.get('pose_steps_per_s',0)/1e6,1), 'M pose-steps/s', d.get('error','')[:300])
PY
