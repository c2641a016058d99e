#!/usr/bin/env python3
"""BASELINE.json configs[4] with the REFERENCE's objective (pose prior + SMPL vertex temporal term + joint data term,
experiments/motion_denoise.py:74-99) on one GPU: S sequences x T frames of a synthetic SMPL-shaped body model (6,890
vertices, 24 joints, 21 vertex-picked joints; posendf_amd.synth.make_body_model).
Times the fused body-model pass (pndf_lbs_terms_grad: three kernels), the forward, and the whole fused Adam step.
Algorithmic work per frame of the fused pass (real dimensions, forward + reverse):
  pose blend shapes 2 x 207 x 20,670 MACs + skinning transforms 2 x 6,890 x 24 x 12 MACs = 12.53 M MACs = 25.06 MFLOP
against 276 B of pose in + 276 B of gradient out: bound = fp32 MFMA (157.3 TFLOP/s dense)."""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from posendf_amd import BodyModel, PoseNDF, amass_config, synth  # noqa: E402
from posendf_amd.motion_denoise import MotionDenoise  # noqa: E402

FLOP_PER_FRAME = 2 * (2 * 207 * 20670 + 2 * 6890 * 24 * 12)
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_FP16_MFMA_TFLOPS = 2500.0


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=64)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--precision", default="f16x3", help="precision of the distance engine")
    ap.add_argument("--terms-only", action="store_true", help="time the fused body-model pass only (diagnostic builds)")
    ap.add_argument("--lbs-precision", default="f16x3", choices=("f16x3", "fp32"), help="arithmetic of the body-model passes")
    args = ap.parse_args()
    S, T = args.seqs, args.frames
    m = synth.make_body_model(seed=11)
    bm = BodyModel(m, device="cuda:0", extra_joint_vertex=m["extra_joint_vertex"], precision=args.lbs_precision)
    peak = PEAK_FP32_MFMA_TFLOPS if args.lbs_precision == "fp32" else PEAK_FP16_MFMA_TFLOPS
    g = torch.Generator().manual_seed(0)
    theta = (torch.cumsum(0.02 * torch.randn(S, T, 69, generator=g), dim=1) + 0.3 * torch.randn(S, 1, 69, generator=g)
             + 0.1 * torch.randn(S, T, 69, generator=g)).cuda()
    j0 = bm.joints_of(theta + 0.02)
    out = torch.empty_like(theta)
    ms_terms = timed(lambda: bm.terms_grad(theta, j0, 2, out=out), args.reps)
    ms_joints = timed(lambda: bm.joints_of(theta), args.reps)
    if args.terms_only:
        print(json.dumps({"lbs_terms_grad_ms": ms_terms, "lbs_forward_joints_only_ms": ms_joints}))
        return
    cfg = amass_config("lrelu", "cuda:0")
    cfg["engine"] = {"precision": args.precision}
    net = PoseNDF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(0, 2.0, 0.1).items()})
    md = MotionDenoise(net, body_model=bm, device="cuda:0")
    md.denoise(theta, iterations=1, steps_per_iter=2, fused=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, _ = md.denoise(theta, iterations=2, steps_per_iter=5, fused=True)
    torch.cuda.synchronize()
    ms_step = (time.perf_counter() - t0) / 10 * 1e3
    md0 = MotionDenoise(net, device="cuda:0")
    md0.denoise(theta, iterations=1, steps_per_iter=2, fused=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    md0.denoise(theta, iterations=2, steps_per_iter=5, fused=True)
    torch.cuda.synchronize()
    ms_step0 = (time.perf_counter() - t0) / 10 * 1e3
    tf = S * T * FLOP_PER_FRAME / (ms_terms * 1e-3) / 1e12
    print(json.dumps({"workload": f"{S} sequences x {T} frames, SMPL-shaped body model (6,890 vertices), body model {args.lbs_precision}, engine {args.precision}",
                      "lbs_terms_grad_ms": ms_terms, "lbs_frames_per_s": S * T / (ms_terms * 1e-3),
                      "roofline": {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s",
                                   "frac": tf / peak, "algorithmic_flop_per_frame": FLOP_PER_FRAME,
                                   "mfma_issued_per_algorithmic_flop": 1 if args.lbs_precision == "fp32" else 3},
                      "lbs_forward_joints_only_ms": ms_joints,
                      "fused_adam_step_ms_reference_objective": ms_step,
                      "fused_adam_step_ms_pose_space_surrogates": ms_step0,
                      "finite": bool(torch.isfinite(res).all())}))


if __name__ == "__main__":
    main()
