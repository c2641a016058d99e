#!/usr/bin/env python3
"""Golden vectors for the quaternion distance + top-k op, produced by the REFERENCE itself
(/root/reference/data/dist_utils.py, classes `geo` and `euc`) in the dev container.  The module imports smplx,
pytorch3d and ipdb at the top without using them in these classes: they are stubbed.  Inputs come from
posendf_amd.synth (seeded), so only the outputs are stored.   usage: python tests/golden/make_golden_dist.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from posendf_amd import synth  # noqa: E402

for name in ("smplx", "ipdb", "pytorch3d"):
    sys.modules[name] = types.ModuleType(name)
tr = types.ModuleType("pytorch3d.transforms")
for fn in ("axis_angle_to_quaternion", "quaternion_to_axis_angle", "axis_angle_to_matrix"):
    setattr(tr, fn, None)
sys.modules["pytorch3d.transforms"] = tr
sys.path.insert(0, "/root/reference/data")
import dist_utils  # noqa: E402

CASES = [(8, 64, 11), (3, 500, 12)]        # (B, K, seed)


def main():
    out = {"torch_version": np.array(torch.__version__)}
    for B, K, seed in CASES:
        noise, valid = synth.make_candidates(B, K, seed)
        for metric in ("geo", "euc"):
            for weighted in (False, True):
                calc = getattr(dist_utils, metric)(B, device="cpu", weighted=weighted)
                val, idx = calc.dist_calc(torch.from_numpy(noise), torch.from_numpy(valid), K, 5)
                tag = f"{metric}_{'w' if weighted else 'u'}_{B}x{K}"
                out[tag + "_val"] = val.numpy()
                out[tag + "_idx"] = idx.numpy()
    np.savez_compressed(os.path.join(HERE, "quat_dist.npz"), **out)
    print("wrote", sorted(out))


if __name__ == "__main__":
    main()
