#!/usr/bin/env python3
"""Robustness sweep on the GPU: both kernels x three activations x weight seeds / gains / pose distributions against
the fp64 numpy oracle, with the reference-arithmetic (fp32 oracle) error printed next to the kernel's.  Writes every
per-pose error vector to gpurun_out/sweep.npz for offline analysis (how the parity gates were calibrated)."""
import itertools
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import d_err, rel_err_rows  # noqa: E402
from oracle import posendf_np as onp  # noqa: E402
from posendf_amd import PoseNDF, amass_config, synth  # noqa: E402

WEIGHTS = ((0, 2.0, 0.1), (0, 2.5, 0.05), (1, 1.0, 0.2), (2, 3.0, 0.05), (3, 0.5, 0.3), (4, 2.5, 0.05))


def d_rows(a, b64):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b64, np.float64).ravel()
    return np.abs(a - b) / np.maximum(np.abs(b), 0.05 * max(np.abs(b).max(), 1e-30))


def main():
    n = int(os.environ.get("SWEEP_POSES", "1024"))
    dump = {}
    for seed, gain, bias in WEIGHTS:
        sd = synth.make_weights(seed, gain, bias)
        for act, signed in itertools.product(("lrelu", "relu", "softplus"), (False, True)):
            qn = synth.make_poses(n, seed=100 + seed, signed=signed)
            d64, g64 = onp.forward_grad(qn, sd, act, dtype=np.float64)
            d32, g32 = onp.forward_grad(qn, sd, act, dtype=np.float32)
            margin = onp.kink_margin(qn, sd, act)
            ref_d, ref_g = d_rows(d32, d64), rel_err_rows(g32, g64)
            key = f"s{seed}_g{gain}_{act}_{int(signed)}"
            dump[key + "_d64"], dump[key + "_margin"] = d64.ravel(), margin
            dump[key + "_refd"], dump[key + "_refg"] = ref_d, ref_g
            for prec in ("fp32", "f16x3"):
                cfg = amass_config(act, "cuda:0")
                cfg["engine"] = {"precision": prec}
                net = PoseNDF(cfg)
                net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
                q = torch.from_numpy(qn).cuda().requires_grad_(True)
                d = net(q, train=False)["dist_pred"]
                (dq,) = torch.autograd.grad(d, q, grad_outputs=torch.ones_like(d))
                kd, kg = d_rows(d.detach().cpu().numpy(), d64), rel_err_rows(dq.cpu().numpy(), g64)
                dump[f"{key}_{prec}_d"], dump[f"{key}_{prec}_g"] = kd, kg
                print(f"seed {seed} gain {gain:.1f} {act:8s} {prec:5s} signed {int(signed)} | d err max {kd.max():.2e} "
                      f"(ref fp32 {ref_d.max():.2e}) p99 {np.percentile(kd, 99):.2e} ({np.percentile(ref_d, 99):.2e}) | "
                      f"grad median {np.median(kg):.2e} ({np.median(ref_g):.2e}) p95 {np.percentile(kg, 95):.2e} "
                      f"({np.percentile(ref_g, 95):.2e}) frac>1e-4 {(kg > 1e-4).mean():.4f} ({(ref_g > 1e-4).mean():.4f}) "
                      f"| mean d {d64.mean():.3g}", flush=True)
    out = os.path.join(REPO, "gpurun_out", "sweep.npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, **dump)
    print("wrote", out)


if __name__ == "__main__":
    main()
