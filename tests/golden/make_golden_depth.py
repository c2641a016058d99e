#!/usr/bin/env python3
"""Golden vectors for DFNets of OTHER depths and widths than configs/amass.yaml, produced by RUNNING THE REAL REFERENCE.

reference model/network/net_modules.py:14-28 builds DFNet from a free list of hidden widths (`dims`); the engine runs every
such network on its runtime-planned kernels (posendf_amd/csrc/pndf_generic.hip).  Dev-container only, like make_golden.py:
imports /root/reference/model/posendf.py with the two arithmetic-free stubs, overrides `model.DFNet.dims` (and `StrEnc.use`) in
the loaded configs/amass.yaml, loads the deterministic weights of posendf_amd.synth and records inputs and outputs.  The
projection loop of experiments/sample_poses.py:67-74 is restated around the imported objects as in make_golden.py.

Usage:  python tests/golden/make_golden_depth.py        (writes tests/golden/depth_<name>.npz)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg      # noqa: E402  (stubs, reference imports, project_ref)

from posendf_amd import synth  # noqa: E402

# name: (hidden widths, activation, structure encoder, weights seed / gain / output bias)
CASES = {
    "d1_lrelu": ([300], "lrelu", True, (42, 2.5, 0.1)),                                       # n_dims 3: one hidden layer
    "d4_lrelu": ([192, 320, 160, 48], "lrelu", True, (22, 2.0, 0.1)),                        # four hidden layers
    "d4_softplus": ([192, 320, 160, 48], "softplus", True, (23, 2.0, 0.1)),
    "d7_lrelu": ([128, 256, 512, 1024, 512, 256, 64], "lrelu", True, (24, 2.2, 0.1)),        # n_dims 9: the deepest the engine plans
    "d7_softplus": ([128, 256, 512, 1024, 512, 256, 64], "softplus", True, (41, 2.2, 0.2)),
    "wide_relu": ([512, 1024, 1024, 640, 256, 128], "relu", True, (26, 2.0, 0.1)),           # amass.yaml's depth, wider layers
    "noenc_d3_lrelu": ([200, 100, 50], "lrelu", False, (27, 2.0, 0.1)),                      # StrEnc.use = False, in_dim 84
    # model.StrEnc.act differs from model.DFNet.act (net_modules.py:128 reads its own key): "trunk/encoder"
    "mix_softplus_lreluenc": ([256, 512, 1024, 512, 256, 64], "softplus/lrelu", True, (28, 2.0, 0.1)),      # amass.yaml's dims
    "mix_relu_softplusenc": ([192, 320, 160, 48], "relu/softplus", True, (43, 2.0, 0.3)),
}
NPOSE = 24


def inputs():
    q = np.concatenate([synth.make_poses(NPOSE, seed=31), synth.make_poses(NPOSE, seed=32, signed=True)])
    edge = synth.make_poses(4, seed=33, signed=True)
    edge[0, :, 2] = 0.0
    edge[1] *= 1e-3
    edge[2, :, :] = edge[2, 0:1, :]
    edge[3, 5, :] = 0.0
    return np.concatenate([q, edge]).astype(np.float32)


def ref_model(name, dtype):
    hidden, act, use_enc, (seed, gain, ob) = CASES[name]
    opt = mg.load_config(os.path.join(mg.REF, "configs", "amass.yaml"))
    opt["train"]["device"] = "cpu"
    trunk_act, _, enc_act = act.partition("/")
    opt["model"]["DFNet"]["act"] = trunk_act
    opt["model"]["DFNet"]["dims"] = list(hidden)
    opt["model"]["StrEnc"]["act"] = enc_act or trunk_act
    opt["model"]["StrEnc"]["use"] = use_enc
    if not use_enc:
        opt["model"]["DFNet"]["in_dim"] = 84
    net = mg.PoseNDF(opt)
    dims = (126 if use_enc else 84, *hidden, 1)
    sd = synth.make_weights(seed, gain, ob, dims=dims)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    return net.to(dtype), sd


def one(name):
    out = {"q": inputs()}
    q_np = out["q"]
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        net, _ = ref_model(name, dtype)
        q = torch.from_numpy(q_np).to(dtype)
        q.requires_grad = True
        pred = net(q, train=False)["dist_pred"]
        g = mg.gradient(q, pred)
        out[f"d_{tag}"] = pred.detach().numpy()
        out[f"dq_{tag}"] = g.detach().numpy()
        if tag == "f32":
            go = torch.from_numpy(np.random.default_rng(6).normal(size=(q.shape[0], 1)).astype(np.float32))
            q2 = torch.from_numpy(q_np).clone().requires_grad_(True)
            (net(q2, train=False)["dist_pred"] * go).sum().backward()
            out["grad_out"] = go.numpy()
            out["grad_pose_f32"] = q2.grad.numpy()
        snaps, trace = mg.project_ref(net, torch.from_numpy(q_np).to(dtype), 10, snap_at=(1, 10))
        for k, v in snaps.items():
            out[f"q{k}_{tag}"] = v.numpy()
        out[f"dtrace_{tag}"] = trace.numpy()
    hidden, act, use_enc, regime = CASES[name]
    out["hidden"] = np.array(hidden)
    out["act"] = np.array(act)
    out["encoder"] = np.array(use_enc)
    out["weights"] = np.array(regime)
    out["torch_version"] = np.array(torch.__version__)
    return out


if __name__ == "__main__":
    for name in (sys.argv[1:] or CASES):
        o = one(name)
        np.savez_compressed(os.path.join(HERE, f"depth_{name}.npz"), **o)
        d = o["d_f32"][:, 0]
        print(f"{name:16s} d: min {d.min():.4f} median {np.median(d):.4f} max {d.max():.4f} zeros {(d == 0).mean():.2f}  "
              f"|dq| median {np.median(np.abs(o['dq_f32']).max(axis=(1, 2))):.3e}  fp32-vs-fp64 d {np.abs(o['d_f32'] - o['d_f64']).max():.1e}")
