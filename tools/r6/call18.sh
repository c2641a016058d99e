#!/bin/bash
# round 6, GPU call 18: where the runtime-planned kernel's time goes -- timing arms of PNDF_GEN_ABLATE (wrong results on purpose)
set -u
OUT=gpurun_out/r6_18
mkdir -p $OUT
for v in g63 h7 h1 h2 h4 a7; do
  if [ $v = product ]; then unset PNDF_LIBRARY; else export PNDF_LIBRARY=$PWD/gpurun_ab/lib_$v.so; fi
  echo "== $v" >> $OUT/gen_ablate.txt
  timeout 300 python tools/bench_generic.py 1 >> $OUT/gen_ablate.txt 2>> $OUT/gen_ablate.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r6_18/gen_ablate.txt'):
    if l.startswith('=='): print(l.strip()); continue
    d=json.loads(l); print('  ', d.get('arm'), round(d.get('ms',0),2), 'ms', round(d.get('frac_of_fp32_mfma_peak',0),3), d.get('error','')[:200])
PY
