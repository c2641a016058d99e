#!/bin/bash
# round 6, GPU call 40: split path: four operand buffers (three k steps of look-ahead)
set -u
OUT=gpurun_out/r6_40
mkdir -p $OUT
timeout 900 python -m pytest tests/test_depth.py -m gpu -q > $OUT/pytest_depth.txt 2>&1
echo "depth rc=$?"; grep -E "^FAILED|passed|failed" $OUT/pytest_depth.txt | tail -12
timeout 900 python tools/bench_generic.py 10 11 12 13 14 15 > $OUT/generic_arch.jsonl 2> $OUT/generic_arch.err
python - <<'PY'
import json
for l in open('gpurun_out/r6_40/generic_arch.jsonl'):
    d=json.loads(l)
    print(d.get('arm'), d.get('kernel'), round(d.get('ms',0),2), 'ms', round(d.get('frac_of_fp32_mfma_peak', d.get('frac_of_fp16_mfma_peak', 0)),3), d.get('error','')[:300])
PY
