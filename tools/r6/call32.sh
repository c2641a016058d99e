#!/bin/bash
# round 6, GPU call 32: a box that ran the benchmark line 8 % slow (684 k, 1,292 W, PCI 0000:f1:00.0 in call 31): its description
set -u
OUT=gpurun_out/r6_32
mkdir -p $OUT
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --diagnostics ) > $OUT/bench_diag.json 2> $OUT/bench_diag.err; tail -3 $OUT/bench_diag.err
python - <<'PY'
import json
for l in open('gpurun_out/r6_32/bench_diag.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline'].get('sclk_mhz'), d['roofline'].get('package_w'), d['roofline'].get('telemetry_source'))
        b=d.get('box',{}); print(json.dumps(b)[:3000])
PY
