#!/usr/bin/env python3
"""Robustness sweep on the GPU: both kernels x three activations x weight seeds / gains / pose distributions against
the fp64 numpy oracle (single forward+grad and a 5-step projection).  Prints the worst cases."""
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


def main():
    worst = []
    for seed, gain, bias in ((0, 2.0, 0.1), (1, 1.0, 0.2), (2, 3.0, 0.05), (3, 0.5, 0.3), (4, 2.5, 0.05)):
        sd = synth.make_weights(seed, gain, bias)
        for act, prec, signed in itertools.product(("lrelu", "relu", "softplus"), ("fp32", "f16x3"), (False, True)):
            cfg = amass_config(act, "cuda:0")
            cfg["engine"] = {"precision": prec}
            net = PoseNDF(cfg)
            net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            qn = synth.make_poses(512, seed=100 + seed, signed=signed)
            q = torch.from_numpy(qn).cuda().requires_grad_(True)
            d = net(q, train=False)["dist_pred"]
            (dq,) = torch.autograd.grad(d, q, grad_outputs=torch.ones_like(d))
            d64, g64 = onp.forward_grad(qn, sd, act, dtype=np.float64)
            d32, g32 = onp.forward_grad(qn, sd, act, dtype=np.float32)
            e_d = d_err(d.detach().cpu().numpy().ravel(), d64.ravel())
            rows = rel_err_rows(dq.cpu().numpy(), g64)
            ref_rows = rel_err_rows(g32, g64)
            rec = (seed, gain, act, prec, signed, e_d, float(np.median(rows)), float((rows > 1e-4).mean()),
                   float((ref_rows > 1e-4).mean()), float(d64.mean()))
            worst.append(rec)
            print("seed %d gain %.1f %-8s %-5s signed %d | d err %.2e  grad median %.2e  frac>1e-4 %.4f (oracle fp32 %.4f)  mean d %.3g" % rec,
                  flush=True)
    bad = [r for r in worst if r[5] > 1e-4 or r[6] > 1e-5 or r[7] > 2 * r[8] + 0.01]
    print("\nsuspicious:", len(bad))
    for r in bad:
        print(r)


if __name__ == "__main__":
    main()
