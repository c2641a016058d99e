#!/usr/bin/env python3
"""Two-term vs three-term split kernels on a half-precision checkpoint (benchmark weights rounded to fp16): same results
bit for bit, fewer MFMAs.  Usage: python tools/x2_bench.py [act]"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, %(repo)r)
from posendf_amd import PoseNDF, amass_config, synth
act = %(act)r
cfg = amass_config(act, "cuda:0"); cfg["engine"] = {"precision": "f16x3"}
net = PoseNDF(cfg)
sd = {k: v.astype(np.float16).astype(np.float32) for k, v in synth.make_weights(0, 2.0, 0.1).items()}   # an fp16 checkpoint
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net.eval()
q = torch.from_numpy(synth.make_poses(65536, seed=1234)).cuda()
net.project(q, steps=100); torch.cuda.synchronize()
ms = []
for _ in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out, d = net.project(q, steps=100); e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
print(json.dumps({"kernel": net._engine_for(q.device).kernel_name(), "ms": sorted(ms)[len(ms) // 2],
                  "checksum": float(out.double().sum()), "dmean": float(d.mean())}))
"""
act = sys.argv[1] if len(sys.argv) > 1 else "lrelu"
for rnd in range(2):
    for three in ("0", "1"):
        env = dict(os.environ, PNDF_THREE_TERMS=three)
        p = subprocess.run([sys.executable, "-c", CHILD % dict(repo=REPO, act=act)], env=env, capture_output=True, text=True)
        print(f"[{act}] PNDF_THREE_TERMS={three}:", p.stdout.strip().splitlines()[-1] if p.returncode == 0 else p.stderr[-500:], flush=True)
