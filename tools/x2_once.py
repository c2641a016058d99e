#!/usr/bin/env python3
"""One process, one kernel: project() on a half-precision checkpoint (two-term kernels; PNDF_THREE_TERMS=1 keeps three terms).
For rocprofv3 --pmc passes (L2 requests of the two kernels side by side).  usage: python tools/x2_once.py [act]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from posendf_amd import PoseNDF, amass_config, synth  # noqa: E402

act = sys.argv[1] if len(sys.argv) > 1 else "lrelu"
cfg = amass_config(act, "cuda:0")
cfg["engine"] = {"precision": "f16x3"}
net = PoseNDF(cfg)
sd = {k: v.astype(np.float16).astype(np.float32) for k, v in synth.make_weights(0, 2.0, 0.1).items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.eval()
q = torch.from_numpy(synth.make_poses(65536, seed=1234)).cuda()
for _ in range(3):
    out, d = net.project(q, steps=100)
torch.cuda.synchronize()
print(net._engine_for(q.device).kernel_name(), float(out.double().sum()))
