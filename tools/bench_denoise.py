#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU's share: motion denoising of S sequences x T frames (default 64 x 300 = the
per-GPU share of 512 sequences on 8 GPUs; --seqs 512 runs all of it on one GPU).  Times Adam steps of
posendf_amd.motion_denoise.MotionDenoise (pose prior on the HIP engine, pose-space temporal / data terms) and the
engine's share of a step."""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from posendf_amd import PoseNDF, amass_config, synth  # noqa: E402
from posendf_amd.motion_denoise import MotionDenoise, axis_angle_to_quaternion  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=64)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--precision", default="f16x3")
    args = ap.parse_args()
    cfg = amass_config("lrelu", "cuda:0")
    cfg["engine"] = {"precision": args.precision}
    net = PoseNDF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(0, 2.0, 0.1).items()})
    g = torch.Generator().manual_seed(0)
    S, T = args.seqs, args.frames
    theta = (torch.cumsum(0.02 * torch.randn(S, T, 69, generator=g), dim=1) + 0.3 * torch.randn(S, 1, 69, generator=g)
             + 0.1 * torch.randn(S, T, 69, generator=g))
    md = MotionDenoise(net, device="cuda:0")
    md.denoise(theta, iterations=1, steps_per_iter=3, record=False)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, _ = md.denoise(theta, iterations=2, steps_per_iter=args.steps // 2, record=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    md.denoise(theta, iterations=1, steps_per_iter=3, fused=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out_f, _ = md.denoise(theta, iterations=2, steps_per_iter=args.steps // 2, fused=True)
    torch.cuda.synchronize()
    df = time.perf_counter() - t2
    # engine share: forward+grad launches alone on the same number of frames
    q = axis_angle_to_quaternion(theta.reshape(S * T, 23, 3)[:, :21].cuda()).contiguous().requires_grad_(True)
    for _ in range(3):
        net(q, train=False)["dist_pred"].sum().backward()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        net(q, train=False)["dist_pred"].sum().backward()
    torch.cuda.synchronize()
    de = time.perf_counter() - t1
    print(json.dumps({"workload": f"motion denoise, {S} sequences x {T} frames, {args.steps} Adam steps, precision {args.precision}",
                      "fused_ms_per_adam_step": df / args.steps * 1e3, "fused_frames_per_s": S * T * args.steps / df,
                      "autograd_driver_ms_per_adam_step": dt / args.steps * 1e3,
                      "fused_vs_autograd_median_abs_diff": float((out_f - out).abs().median()),
                      "engine_fwd_grad_ms_per_step": de / args.steps * 1e3,
                      "finite": bool(torch.isfinite(out).all())}))


if __name__ == "__main__":
    main()
