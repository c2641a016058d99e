#!/bin/bash
# round 6, GPU call 41 (after derivative bits + four operand buffers): timing arms of the split-precision runtime-planned kernel (wrong results on purpose)
set -u
OUT=gpurun_out/r6_41
mkdir -p $OUT
for v in product g1 g3 g4 g32 g47 a1 a2 a4 h7; do
  if [ $v = product ]; then unset PNDF_LIBRARY; else export PNDF_LIBRARY=$PWD/gpurun_ab/lib_$v.so; fi
  echo "{\"variant\": \"$v\"}" >> $OUT/arms.jsonl
  timeout 300 python tools/bench_generic.py 11 >> $OUT/arms.jsonl 2>> $OUT/arms.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r6_41/arms.jsonl'):
    d=json.loads(l)
    if 'variant' in d: print('==', d['variant']); continue
    print('  ', d.get('arm'), round(d.get('ms',0),2), 'ms', d.get('error','')[:200])
PY
