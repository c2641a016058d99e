#!/usr/bin/env python3
"""The small-batch regime (the reference's own default workloads: experiments/sample_poses.py projects 10 poses for 10 steps,
experiments/motion_denoise.py optimises one T-frame sequence): time of ONE projection step as a function of the batch, this
engine (one persistent launch for all steps; f16x3 and exact fp32) beside the PyTorch-ROCm restatement of the reference on the same
GPU.  One JSON line per batch size.  usage: python tools/bench_small_batch.py > profiles/r06/small_batch.jsonl"""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from posendf_amd import PoseNDF, amass_config, synth  # noqa: E402
from oracle.posendf_torch import RefNet, project as torch_project  # noqa: E402  (the comparator, never the product path)

ACT = sys.argv[1] if len(sys.argv) > 1 else "lrelu"
STEPS = 10
dev = torch.device("cuda:0")
sd = synth.make_weights(0, 2.0, 0.1)
nets = {}
for prec in ("f16x3", "fp32"):
    cfg = amass_config(ACT, "cuda:0")
    cfg["engine"] = {"precision": prec}
    n = PoseNDF(cfg)
    n.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    n.eval()
    nets[prec] = n
ref = RefNet(ACT).to(dev)
ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


for B in (10, 64, 300, 1024, 4096, 16384, 65536):
    q = torch.from_numpy(synth.make_poses(B, seed=1234)).to(dev)
    row = {"batch": B, "steps": STEPS, "act": ACT, "workgroups": (B + 63) // 64}
    for prec, n in nets.items():
        t = timed(lambda: n.project(q, steps=STEPS), 5)
        row[f"{prec}_us_per_step"] = t / STEPS * 1e6
        row[f"{prec}_poses_per_s_at_100_steps"] = B / (t / STEPS * 100)
    t = timed(lambda: torch_project(ref, q, STEPS), 3)
    row["torch_rocm_us_per_step"] = t / STEPS * 1e6
    row["speedup_f16x3_vs_torch"] = row["torch_rocm_us_per_step"] / row["f16x3_us_per_step"]
    row["speedup_fp32_vs_torch"] = row["torch_rocm_us_per_step"] / row["fp32_us_per_step"]
    print(json.dumps(row), flush=True)
