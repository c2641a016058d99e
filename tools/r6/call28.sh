#!/bin/bash
# round 6, GPU call 28: runtime-planned kernels v13 + non-temporal epilogue stores: depth parity, whole C-ABI test file, bench
set -u
OUT=gpurun_out/r6_28
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_depth.py tests/test_cabi.py -m gpu -q > $OUT/pytest_depth.txt 2>&1
echo "depth rc=$?"; tail -4 $OUT/pytest_depth.txt
timeout 900 python tools/bench_generic.py > $OUT/generic_arch.jsonl 2> $OUT/generic_arch.err
python - <<'PY'
import json
for l in open('gpurun_out/r6_28/generic_arch.jsonl'):
    d=json.loads(l)
    print(d.get('arm'), d.get('kernel'), round(d.get('ms',0),2), 'ms', round(d.get('frac_of_fp32_mfma_peak',0),3), d.get('error','')[:300])
PY
