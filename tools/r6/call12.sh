#!/bin/bash
# round 6, GPU call 12: the whole GPU suite on the build with the debug library split off, the width / depth extremes, both bench lines
set -u
OUT=gpurun_out/r6_12
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1
echo "suite rc=$?"; tail -6 $OUT/pytest_gpu.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
echo "bench rc=$?"; tail -3 $OUT/bench_default.time
python bench.py --steps 5 --warmup 2 --diagnostics > $OUT/bench_diag.json 2> $OUT/bench_diag.err
echo "bench --diagnostics rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6_12/bench_diag.json') if l.startswith('{')][-1])
print(sorted(d.keys())); print(d['value'], d['regions']['ring'], d['box']['power_window'].get('energy_j_per_launch'))
PY
cat /proc/$$/maps > /dev/null; python - <<'PY'
# which native libraries does a plain product run map?  (the debug library must not be among them)
import torch, numpy as np, sys
sys.path.insert(0, '.')
from posendf_amd import PoseNDF, amass_config, synth
net = PoseNDF(amass_config("lrelu", "cuda:0")); net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(0, 2.0, 0.1).items()}); net.eval()
net.project(torch.from_numpy(synth.make_poses(256, seed=1)).cuda(), steps=2); torch.cuda.synchronize()
print(sorted({l.split()[-1] for l in open('/proc/self/maps') if 'posendf' in l}))
PY
