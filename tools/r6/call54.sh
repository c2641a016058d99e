#!/bin/bash
# round 6, GPU call 54: what the driver runs at round end, on the final tree: smoke(), then bench.py with no flags
set -u
OUT=gpurun_out/r6_54
mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -8 $OUT/smoke.txt | cut -c1-200
( time timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; echo "bench rc=$?"; cat $OUT/bench_default.time
python -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['steps'], d['warmup'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], list(d))"
