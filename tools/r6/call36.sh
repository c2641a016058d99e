#!/bin/bash
# round 6, GPU call 36: cache-policy arms of the runtime-planned kernels' scratch traffic (correct results): fp32 and split, amass dims
set -u
OUT=gpurun_out/r6_36
mkdir -p $OUT
for v in product n512 n1024 n1536 n256; do
  if [ $v = product ]; then unset PNDF_LIBRARY; else export PNDF_LIBRARY=$PWD/gpurun_ab/lib_$v.so; fi
  echo "{\"variant\": \"$v\"}" >> $OUT/arms.jsonl
  timeout 300 python tools/bench_generic.py 1 11 12 >> $OUT/arms.jsonl 2>> $OUT/arms.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r6_36/arms.jsonl'):
    d=json.loads(l)
    if 'variant' in d: print('==', d['variant']); continue
    print('  ', d.get('arm'), round(d.get('ms',0),2), 'ms', d.get('error','')[:200])
PY
