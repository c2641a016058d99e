#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REAL REFERENCE.

Dev-container only: imports /root/reference/model/posendf.py (PoseNDF, gradient) with two stub
modules for the absent, arithmetic-free imports (`ipdb`, `torch.utils.tensorboard`), loads the
deterministic weights of posendf_amd.synth into it with load_state_dict and records outputs.
The hot loops of experiments/sample_poses.py:67-74 and experiments/motion_denoise.py:81-83 cannot be
imported (pytorch3d/smplx at module level), so they are re-stated here around the imported
PoseNDF + gradient, line for line in semantics.

Nothing of the reference is copied into the repository: only inputs and outputs (data).
Usage:  python tests/golden/make_golden.py        (writes tests/golden/posendf_<act>_<regime>.npz)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("POSENDF_REFERENCE", "/root/reference")

# --- stubs for arithmetic-free imports (reference model/posendf.py:5,9) ---------------------------
ipdb = types.ModuleType("ipdb")
ipdb.set_trace = lambda *a, **k: None
sys.modules["ipdb"] = ipdb
tb = types.ModuleType("torch.utils.tensorboard")
tb.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None})
sys.modules["torch.utils.tensorboard"] = tb
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

from configs.config import load_config          # noqa: E402  (reference)
from model.posendf import PoseNDF, gradient     # noqa: E402  (reference)

from posendf_amd import synth                   # noqa: E402  (this repo: weights + poses only)

REGIMES = {"live": dict(seed=0, gain=2.0, out_bias=0.1), "mixed": dict(seed=0, gain=2.5, out_bias=0.05),
           # further weight sets (round 2): the regimes in which tools/gpu_sweep.py found the parity gates fraying --
           # deep cancellation before the output activation (s2g3: mean d 0.005..0.9), a softplus net with ~2 % of
           # badly conditioned poses (s4g25), and a small-gain net whose d is dominated by lin6.bias (s1g1)
           "s2g3": dict(seed=2, gain=3.0, out_bias=0.05), "s4g25": dict(seed=4, gain=2.5, out_bias=0.05),
           "s1g1": dict(seed=1, gain=1.0, out_bias=0.2)}
LITE = ("s2g3", "s4g25", "s1g1")      # single step, autograd contract and 1/10-step projections only (smaller files)
NPOSE = 48


def make_inputs():
    q = np.concatenate([synth.make_poses(NPOSE, seed=11), synth.make_poses(NPOSE, seed=12, signed=True)])
    edge = synth.make_poses(4, seed=13, signed=True)
    edge[0, :, 2] = 0.0            # zero component column -> eps clamp of F.normalize
    edge[1] *= 1e-3                # tiny pose (normalisation makes it scale invariant)
    edge[2, :, :] = edge[2, 0:1, :]  # all joints equal
    edge[3, 5, :] = 0.0            # one zero quaternion
    return np.concatenate([q, edge]).astype(np.float32)


def ref_model(act, regime, dtype):
    opt = load_config(os.path.join(REF, "configs", "amass.yaml"))
    opt["train"]["device"] = "cpu"
    opt["model"]["DFNet"]["act"] = act
    opt["model"]["StrEnc"]["act"] = act
    net = PoseNDF(opt)
    sd = synth.make_weights(**REGIMES[regime])
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    return net.to(dtype)


def project_ref(net, q0, steps, snap_at=(1, 10, 100)):
    """experiments/sample_poses.py:67-74 restated around the imported reference objects."""
    noisy = q0.clone()
    noisy.requires_grad = True
    trace = []
    snaps = {}
    for it in range(steps):
        net_pred = net(noisy, train=False)
        grad = gradient(noisy, net_pred["dist_pred"]).reshape(-1, 84)
        noisy = (noisy - (net_pred["dist_pred"] * grad).reshape(-1, 21, 4)).detach()
        noisy.requires_grad = True          # detached formulation: values identical
        trace.append(net_pred["dist_pred"].detach()[:, 0].clone())
        if it + 1 in snap_at:
            snaps[it + 1] = noisy.detach().clone()
    return snaps, torch.stack(trace)


def one(act, regime):
    lite = regime in LITE
    out = {}
    q_np = make_inputs()
    out["q"] = q_np
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        net = ref_model(act, regime, dtype)
        q = torch.from_numpy(q_np).to(dtype)
        q.requires_grad = True
        pred = net(q, train=False)["dist_pred"]
        g = gradient(q, pred)
        out[f"d_{tag}"] = pred.detach().numpy()
        out[f"dq_{tag}"] = g.detach().numpy()
        if tag == "f32":
            # intermediates (reference submodules called directly)
            n = torch.nn.functional.normalize(q.detach().reshape(-1, 21, 4), dim=1)
            out["n_f32"] = n.numpy()
            out["feat_f32"] = net.enc(n).detach().numpy()
            # autograd contract: arbitrary upstream grad_output (motion_denoise.py:82-83,97-98)
            go = torch.from_numpy(np.random.default_rng(5).normal(size=(q.shape[0], 1)).astype(np.float32))
            q2 = torch.from_numpy(q_np).clone().requires_grad_(True)
            (net(q2, train=False)["dist_pred"] * go).sum().backward()
            out["grad_out"] = go.numpy()
            out["grad_pose_f32"] = q2.grad.numpy()
            # pose-prior objective of motion_denoise.py:81-83 + weight :33, it = 0 and 3
            for it in (() if lite else (0, 3)):
                q3 = torch.from_numpy(q_np).clone().requires_grad_(True)
                c = torch.mean(net(q3, train=False)["dist_pred"])
                obj = 10.0 ** 7 * c * c / (1 + it)
                obj.backward()
                out[f"prior_obj_it{it}"] = np.float32(obj.item())
                out[f"prior_grad_it{it}"] = q3.grad.numpy()
        snaps, trace = project_ref(net, torch.from_numpy(q_np).to(dtype), 10 if lite else 100)
        for k, v in snaps.items():
            out[f"q{k}_{tag}"] = v.numpy()
        out[f"dtrace_{tag}"] = trace.numpy()
    out["torch_version"] = np.array(torch.__version__)
    out["regime"] = np.array(str(REGIMES[regime]))
    return out


def train_smoke(regime="live"):
    """Pin the train=True fall-through (posendf.py:62-99): loss dict for a fixed batch."""
    net = ref_model("lrelu", regime, torch.float32)
    q = torch.from_numpy(synth.make_poses(32, seed=21))
    man = torch.from_numpy(synth.make_poses(32, seed=22))
    dist = torch.from_numpy(np.random.default_rng(23).random(32).astype(np.float32) * 0.2)
    loss, ld = net(q.clone(), dist, man, train=True, eikonal=1.0)
    return {"train_q": q.numpy(), "train_man": man.numpy(), "train_dist": dist.numpy(),
            "train_loss": np.float32(loss.item()),
            "train_man_loss": np.float32(ld["man_loss"].item()),
            "train_eikonal": np.float32(ld["eikonal"].item())}


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = sys.argv[1:]                      # regime names; default: all
    for act in ("lrelu", "relu", "softplus"):
        for regime in (only or REGIMES):
            res = one(act, regime)
            if act == "lrelu" and regime == "live":
                res.update(train_smoke())
            path = os.path.join(HERE, f"posendf_{act}_{regime}.npz")
            np.savez_compressed(path, **res)
            print(path, os.path.getsize(path) // 1024, "KiB",
                  "d range", float(res["d_f32"].min()), float(res["d_f32"].max()),
                  "zeros", int((res["d_f32"] == 0).sum()))
