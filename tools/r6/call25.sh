#!/bin/bash
# round 6, GPU call 25: runtime-planned kernels v12 (nothing decided at run time inside a k step): parity + bench + arms
set -u
OUT=gpurun_out/r6_25
mkdir -p $OUT
timeout 900 python -m pytest tests/test_depth.py -m gpu -q -x > $OUT/pytest_depth.txt 2>&1
echo "depth rc=$?"; tail -4 $OUT/pytest_depth.txt
timeout 900 python tools/bench_generic.py > $OUT/generic_arch.jsonl 2> $OUT/generic_arch.err
for v in h7 i64 g63; do
  export PNDF_LIBRARY=$PWD/gpurun_ab/lib_$v.so
  echo "{\"variant\": \"$v\"}" >> $OUT/generic_arch.jsonl
  timeout 300 python tools/bench_generic.py 1 >> $OUT/generic_arch.jsonl 2>> $OUT/generic_arch.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r6_25/generic_arch.jsonl'):
    d=json.loads(l)
    if 'variant' in d: print('==', d['variant']); continue
    print(d.get('arm'), d.get('kernel'), round(d.get('ms',0),2), 'ms', round(d.get('frac_of_fp32_mfma_peak',0),3), d.get('error','')[:300])
PY
