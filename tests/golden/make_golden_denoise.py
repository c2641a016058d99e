#!/usr/bin/env python3
"""Golden vectors for the motion-denoise optimiser step, produced by RUNNING THE REAL REFERENCE NETWORK
inside the reference's optimisation loop.

Dev-container only.  `experiments/motion_denoise.py` itself cannot be imported (pytorch3d / smplx / body_model at module
level), so its loop (:58-99: Adam(lr=0.02, betas=(0.9, 0.999)) :70, zero_grad / loss dict / backward_step / backward /
step :78-99, weights :29-35, pose-prior term :81-83) is restated here around the IMPORTED reference `PoseNDF` -- the same
way make_golden.py restates the projection loop -- with torch autograd and torch.optim.Adam doing the arithmetic.
Two things are not the reference's (and not under /root/reference): pytorch3d's axis_angle_to_quaternion (restated from
its documented convention, parity unpinned) and the SMPL body model, whose vertex / joint terms (:86-94)
are replaced by this repository's pose-space surrogates (per-joint axis-angle differences, same weights and reductions).
Only inputs and outputs are stored.   Usage: python tests/golden/make_golden_denoise.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg     # noqa: E402  (stubs + reference imports + ref_model)


def axis_angle_to_quaternion(a):
    ang = torch.norm(a, p=2, dim=-1, keepdim=True)
    half = 0.5 * ang
    small = ang.abs() < 1e-6
    k = torch.where(small, 0.5 - ang * ang / 48.0, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    return torch.cat([torch.cos(half), a * k], dim=-1)


def noisy_sequence(T, seed):
    g = torch.Generator().manual_seed(seed)
    walk = torch.cumsum(0.02 * torch.randn(T, 69, generator=g), dim=0) + 0.3 * torch.randn(1, 69, generator=g)
    th = walk + 0.1 * torch.randn(T, 69, generator=g)
    th[2, 6:9] = 0.0            # one exact zero rotation: small-angle branch of the conversion and of its Jacobian
    return th


def run(act, regime, dtype, T=16, iterations=2, steps_per_iter=4):
    net = mg.ref_model(act, regime, dtype)
    noisy = noisy_sequence(T, seed=5).to(dtype)
    body_pose = noisy.clone().requires_grad_(True)
    init = noisy.reshape(T, 23, 3)[:, :21].clone()
    optimizer = torch.optim.Adam([body_pose], 0.02, betas=(0.9, 0.999))                     # :70
    weight = {"temp": lambda c, it: 10. ** 1 * c * (1 + it), "data": lambda c, it: 10. ** 2 * c / (1 + it),
              "pose_pr": lambda c, it: 10. ** 7 * c * c / (1 + it)}                         # :29-35
    thetas, terms = [], []
    for it in range(iterations):                                                            # :74
        for _ in range(steps_per_iter):                                                     # :77
            optimizer.zero_grad()
            loss = {}
            pose_quat = axis_angle_to_quaternion(body_pose.view(-1, 23, 3)[:, :21])         # :81
            dis_val = net(pose_quat, train=False)["dist_pred"]                              # :82
            loss["pose_pr"] = torch.mean(dis_val)                                           # :83
            pts = body_pose.view(-1, 23, 3)[:, :21]                # pose-space surrogate of vertices / joints
            t = pts[:-1] - pts[1:]                                                          # :88
            loss["temp"] = torch.mean(torch.sqrt(torch.sum(t * t, dim=2) + 1e-20))          # :89 (+ surrogate guard)
            if it > 0:                                                                      # :92
                dt = pts - init
                loss["data"] = torch.mean(torch.sqrt(torch.sum(dt * dt, dim=2) + 1e-20))    # :93-94
            tot = torch.stack([weight[k](v, it) for k, v in loss.items()]).sum()            # :37-45
            tot.backward()                                                                  # :98
            optimizer.step()                                                                # :99
            thetas.append(body_pose.detach().clone().numpy())
            terms.append([float(weight[k](v, it)) for k, v in sorted(loss.items())] + [float("nan")] * (3 - len(loss)))
    return noisy.numpy(), np.stack(thetas), np.array(terms)


if __name__ == "__main__":
    torch.set_num_threads(8)
    out = {}
    for act in ("lrelu", "softplus"):
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            noisy, th, terms = run(act, "live", dtype)
            out["theta0"] = noisy.astype(np.float32)
            out[f"{act}_theta_{tag}"] = th
            out[f"{act}_terms_{tag}"] = terms
    out["meta"] = np.array("regime live; T=16; iterations=2; steps_per_iter=4; lr=0.02; torch " + torch.__version__)
    path = os.path.join(HERE, "denoise_live.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB; moved", float(np.abs(out["lrelu_theta_f64"][-1] - out["theta0"]).max()))
