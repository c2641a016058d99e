#!/bin/bash
# round 6, GPU call 21: runtime-planned kernels v10: reads, pieces and copies spread over the group
set -u
OUT=gpurun_out/r6_21
mkdir -p $OUT
timeout 900 python -m pytest tests/test_depth.py -m gpu -q -x > $OUT/pytest_depth.txt 2>&1
echo "depth rc=$?"; tail -6 $OUT/pytest_depth.txt
timeout 900 python tools/bench_generic.py > $OUT/generic_arch.jsonl 2> $OUT/generic_arch.err
echo "bench_generic rc=$?"; python - <<'PY'
import json
for l in open('gpurun_out/r6_21/generic_arch.jsonl'):
    d=json.loads(l); print(d.get('arm'), d.get('kernel'), round(d.get('ms',0),2), 'ms', round(d.get('frac_of_fp32_mfma_peak',0),3), d.get('error','')[:300])
PY
