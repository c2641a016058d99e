#!/usr/bin/env python3
"""Randomised NARROW networks -- six hidden widths within configs/amass.yaml's, which run zero padded on the fused kernels (or, Softplus
with a layer of a few units under split precision, on the runtime-planned ones: csrc/pndf_generic.hip pndf_generic_needed) -- through
both precisions against the numpy oracle with the per-pose gates of tests/conftest.py.  usage: python tools/sweep_narrow.py [n] [seed]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import conftest as cf  # noqa: E402
import test_depth as td  # noqa: E402
from posendf_amd import PoseNDF, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
CAPS = (256, 512, 1024, 512, 256, 64)
fails = 0
for i in range(n):
    hidden = [int(rng.choice([1, 2, 3, 8, 9, 15, 16, 17, c // 2, c - 1, c])) if rng.random() < 0.5 else int(rng.integers(1, c + 1)) for c in CAPS]
    act = str(rng.choice(["lrelu", "relu", "softplus"]))
    enc = bool(rng.random() < 0.7)
    dims = (126 if enc else 84, *hidden, 1)
    try:
        sd = td.live_weights(dims, act)
    except AssertionError:
        print(f"[{i}] {hidden} {act} enc={enc}: dead network, skipped", flush=True)
        continue
    q_np = np.concatenate([synth.make_poses(100, seed=61), synth.make_poses(100, seed=62, signed=True)])
    sig_d, sig_g, d64, g64 = cf.fp32_noise(q_np, sd, act)
    for precision in ("fp32", "f16x3"):
        cfg = td.config_for(hidden, act, enc, "cuda:0")
        cfg["engine"] = {"precision": precision}
        net = PoseNDF(cfg)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net.eval()
        q = torch.from_numpy(q_np).cuda().requires_grad_(True)
        d = net(q, train=False)["dist_pred"]
        (dq,) = torch.autograd.grad(d.sum(), q)
        name = net._engine_for(q.device).kernel_name()
        what = f"{hidden} {act} enc={enc} {precision}"
        try:
            cf.pose_gate(cf.d_rows(d.detach().cpu().numpy(), d64), sig_d, what + " d", escalate=lambda k: cf.escalated_noise(q_np, sd, act, k, d64, kind="d"))
            cf.pose_gate(cf.rel_err_rows(dq.cpu().numpy(), g64), sig_g, what + " dq", exempt=td.kink_exempt(q_np, sd, act),
                         escalate=lambda k: cf.escalated_noise(q_np, sd, act, k, g64, kind="g"))
            print(f"[{i}] {what} {name}: ok", flush=True)
        except AssertionError as exc:
            fails += 1
            print(f"[{i}] {what} {name}: FAIL {str(exc)[:260]}", flush=True)
print(f"sweep_narrow: {fails} failure(s)")
sys.exit(1 if fails else 0)
