#!/bin/bash
# round 6, GPU call 31: the driver's round-end sequence on the final tree: smoke(), then the default bench line
set -u
OUT=gpurun_out/r6_31
mkdir -p $OUT
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $OUT/smoke.txt 2>&1; tail -5 $OUT/smoke.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err; tail -4 $OUT/bench.err
python - <<'PY'
import json
for l in open('gpurun_out/r6_31/bench.json'):
    if l.startswith('{'):
        d=json.loads(l); print({k:d[k] for k in ('metric','value','unit','ms_per_step','dtype','vs_baseline')}); print(d['roofline']); print(d['cpu_baseline']); print(d['config'])
PY
