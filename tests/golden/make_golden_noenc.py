#!/usr/bin/env python3
"""Golden vectors of the encoder-less configuration (model.StrEnc.use = False, DFNet in_dim = 84: reference
model/posendf.py:40-42,73-74), produced by running the REAL reference exactly like make_golden.py does for the default
configuration.  Writes tests/golden/posendf_noenc_<act>_live.npz (inputs, d, dd/dq, 1/10/100-step projections in
fp32 and fp64).   usage: python tests/golden/make_golden_noenc.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs ipdb / tensorboard, puts the reference on sys.path)

from posendf_amd import synth  # noqa: E402


def ref_model(act, dtype):
    opt = mg.load_config(os.path.join(mg.REF, "configs", "amass.yaml"))
    opt["train"]["device"] = "cpu"
    opt["model"]["DFNet"]["act"] = act
    opt["model"]["StrEnc"]["use"] = False
    opt["model"]["DFNet"]["in_dim"] = 84
    net = mg.PoseNDF(opt)
    sd = synth.make_weights(seed=0, gain=2.0, out_bias=0.1, dims=synth.DFNET_DIMS_NOENC)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    return net.to(dtype)


def one(act):
    out = {"q": mg.make_inputs()}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        net = ref_model(act, dtype)
        q = torch.from_numpy(out["q"]).to(dtype)
        q.requires_grad = True
        pred = net(q, train=False)["dist_pred"]
        out[f"d_{tag}"] = pred.detach().numpy()
        out[f"dq_{tag}"] = mg.gradient(q, pred).detach().numpy()
        snaps, trace = mg.project_ref(net, torch.from_numpy(out["q"]).to(dtype), 100)
        for k, v in snaps.items():
            out[f"q{k}_{tag}"] = v.numpy()
        out[f"dtrace_{tag}"] = trace.numpy()
    out["torch_version"] = np.array(torch.__version__)
    return out


if __name__ == "__main__":
    torch.set_num_threads(8)
    for act in ("lrelu", "softplus"):
        res = one(act)
        path = os.path.join(HERE, f"posendf_noenc_{act}_live.npz")
        np.savez_compressed(path, **res)
        print(path, {k: v.shape for k, v in res.items() if hasattr(v, "shape") and v.ndim > 0})
