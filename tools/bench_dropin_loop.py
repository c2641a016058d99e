#!/usr/bin/env python3
"""The reference's OWN loop, statement by statement, around the drop-in class (experiments/sample_poses.py:67-74:
`net(q, train=False)` -> `gradient(q, dist_pred)` -> `q = q - dist_pred * grad`): wall time per iteration at the reference's batch
sizes, beside the fused `project()` and the same loop around the PyTorch-ROCm restatement.  What a user who switches the import and
changes nothing else gets.  One JSON line per batch size.  usage: python tools/bench_dropin_loop.py [act]"""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from posendf_amd import PoseNDF, amass_config, synth  # noqa: E402
from posendf_amd.facade import gradient  # noqa: E402
from oracle.posendf_torch import RefNet  # noqa: E402  (the comparator, never the product path)

ACT = sys.argv[1] if len(sys.argv) > 1 else "lrelu"
ITERS = 10
dev = torch.device("cuda:0")
sd = synth.make_weights(0, 2.0, 0.1)
cfg = amass_config(ACT, "cuda:0")
net = PoseNDF(cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.eval()
ref = RefNet(ACT).to(dev)
ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})


def loop(model, q, call):
    q = q.clone()
    q.requires_grad = True                                       # :66
    for _ in range(ITERS):                                       # :70
        d = call(model, q)                                       # :71
        grad = gradient(q, d).reshape(-1, 84)                    # :73
        q = q - (d * grad).reshape(-1, 21, 4)                    # :74
    return q


def timed(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


for B in (1, 10, 300, 4096, 65536):
    q0 = torch.from_numpy(synth.make_poses(B, seed=1234)).to(dev)
    t_loop = timed(lambda: loop(net, q0, lambda m, q: m(q, train=False)["dist_pred"]))
    t_proj = timed(lambda: net.project(q0, steps=ITERS))
    t_ref = timed(lambda: loop(ref, q0, lambda m, q: m(q)), reps=3)
    print(json.dumps({"batch": B, "iterations": ITERS, "act": ACT,
                      "dropin_loop_us_per_iteration": t_loop / ITERS * 1e6, "fused_project_us_per_step": t_proj / ITERS * 1e6,
                      "torch_rocm_loop_us_per_iteration": t_ref / ITERS * 1e6,
                      "speedup_dropin_vs_torch": t_ref / t_loop, "speedup_project_vs_torch": t_ref / t_proj}), flush=True)
