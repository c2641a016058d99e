#!/usr/bin/env python3
"""Comparator for north_star's '>= 10x the reference single-GPU PyTorch poses/sec': the PyTorch restatement
of the reference (oracle/posendf_torch.py, same nn.Linear / LeakyReLU / cat / autograd.grad op sequence) run
through stock PyTorch-ROCm on one MI355X, fp32, B = 65,536, timed over a few projection steps and
extrapolated linearly to 100 steps (every step does identical work).  Diagnostic tool, not product code."""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.posendf_torch import RefNet, project   # noqa: E402
from posendf_amd import synth                      # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    sd = synth.make_weights(0, 2.0, 0.1)
    net = RefNet("lrelu").cuda()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    q = torch.from_numpy(synth.make_poses(B, seed=1234)).cuda()
    project(net, q, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    project(net, q, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_step = dt / steps
    print(json.dumps({"what": "PyTorch-ROCm fp32 restatement of the reference on 1 MI355X", "B": B,
                      "timed_steps": steps, "ms_per_step": per_step * 1e3,
                      "projected_poses_per_s_at_100_steps": B / (per_step * 100),
                      "tflops": B * 5450416 / per_step / 1e12, "torch": torch.__version__}))


if __name__ == "__main__":
    main()
