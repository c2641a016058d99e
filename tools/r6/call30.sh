#!/bin/bash
# round 6, GPU call 30: the small-batch regime measured
set -u
OUT=gpurun_out/r6_30
mkdir -p $OUT
timeout 600 python tools/bench_small_batch.py lrelu > $OUT/small_batch.jsonl 2> $OUT/small_batch.err
timeout 600 python tools/bench_small_batch.py softplus >> $OUT/small_batch.jsonl 2>> $OUT/small_batch.err
python - <<'PY'
import json
for l in open('gpurun_out/r6_30/small_batch.jsonl'):
    d=json.loads(l); print(d['act'], d['batch'], 'wgs', d['workgroups'], 'f16x3 %.0f us' % d['f16x3_us_per_step'], 'fp32 %.0f us' % d['fp32_us_per_step'], 'torch %.0f us' % d['torch_rocm_us_per_step'], 'x%.1f' % d['speedup_f16x3_vs_torch'])
PY
tail -3 $OUT/small_batch.err
