#!/usr/bin/env python3
"""A TRAINED network as a parity regime, produced with the REAL reference (dev container only).

No published checkpoint is reachable offline, and every other golden file uses random-init weights.  This script trains
the imported reference `PoseNDF` (model/posendf.py, train=True path: L1 distance loss + manifold loss + eikonal term,
combined as model/train_posendf.py:92-96 does) for a few hundred Adam steps on a synthetic pose manifold, with the
reference's own DFNet built NARROWER than configs/amass.yaml (`model.DFNet.dims`, net_modules.py:14-28) so that the
checkpoint is small enough to commit.  The trained weights are rounded to fp16-representable fp32 values (that IS the
checkpoint: 2 bytes per weight on disk) and loaded back into the reference, which then produces the vectors: d, dd/dq,
the autograd contract, 1- and 10-step projections with the per-step distance trace, in fp32 and fp64.

It pins two things the random-init fixtures cannot: a network whose distances mean something (d -> 0 on the manifold,
growing with the distance from it, unit-ish gradient), and the narrower-architecture path of the engine against the
reference itself (the other test of that path uses the numpy oracle).

Nothing of the reference is copied: only weights it trained and inputs / outputs (data).
Usage:  python tests/golden/make_golden_trained.py      (writes tests/golden/trained_<act>.npz)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("POSENDF_REFERENCE", "/root/reference")

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

from posendf_amd import synth                   # noqa: E402  (this repo: input poses only)

HIDDEN = [96, 160, 256, 160, 96, 32]            # same depth as amass.yaml, narrower (about 130 k parameters)
STEPS, BATCH, LR = 500, 256, 1e-3
LOSS_W = {"dist": 1.0, "man_loss": 0.3, "eikonal": 0.02}    # keys of amass.yaml:57-59; the two regularisers down-weighted


def unit(q):
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


class Manifold:
    """Synthetic 'valid pose' set: unit quaternions q_j = normalise(base_j + sum_k z_k A_kj), z in R^6."""

    def __init__(self, seed=7, latent=6):
        rng = np.random.default_rng(seed)
        self.base = unit(rng.normal(size=(21, 4)) * 0.3 + np.array([1.0, 0, 0, 0]))
        self.A = rng.normal(size=(latent, 21, 4)) * 0.35
        self.latent = latent

    def sample(self, n, rng):
        z = rng.normal(size=(n, self.latent, 1, 1))
        return unit(self.base[None] + (z * self.A[None]).sum(1)).astype(np.float32)


def geo_dist_to_set(q, cand):
    """mean over joints of 1 - |<q, q'>| to the nearest candidate (the geodesic form of data/dist_utils.py:42-50, k = 1)"""
    dots = np.abs(np.einsum("njc,mjc->nmj", q.astype(np.float64), cand.astype(np.float64)))
    return (1.0 - dots).mean(-1).min(1).astype(np.float32)


def make_batch(man, cand, n, rng):
    clean = man.sample(n, rng)
    sigma = rng.uniform(0.0, 0.6, size=(n, 1, 1)) ** 2 * 1.5           # mostly near the manifold, some far
    noisy = unit(clean + sigma * rng.normal(size=clean.shape)).astype(np.float32)
    return noisy, geo_dist_to_set(noisy, cand), clean


def ref_model(act, dtype=torch.float32):
    opt = load_config(os.path.join(REF, "configs", "amass.yaml"))
    opt["train"]["device"] = "cpu"
    opt["model"]["DFNet"]["act"] = act
    opt["model"]["StrEnc"]["act"] = act
    opt["model"]["DFNet"]["dims"] = list(HIDDEN)
    return PoseNDF(opt).to(dtype)


def train(act, seed=0):
    torch.manual_seed(seed)
    rng = np.random.default_rng(100 + seed)
    man = Manifold()
    cand = man.sample(2000, np.random.default_rng(5))
    net = ref_model(act)
    with torch.no_grad():
        net.dfnet.lin6.bias.fill_(0.1)       # start with a live output ReLU (nn.Linear's default init leaves half of them dead)
    net.train()
    optim = torch.optim.Adam(net.parameters(), lr=LR)
    for it in range(STEPS):
        noisy, dist, clean = make_batch(man, cand, BATCH, rng)
        optim.zero_grad()
        _, ld = net(torch.from_numpy(noisy), torch.from_numpy(dist), torch.from_numpy(clean), eikonal=LOSS_W["eikonal"])
        loss = sum(LOSS_W[k] * ld[k] for k in ld)            # train_posendf.py:92-96
        loss.backward()
        optim.step()
        if it % 100 == 0 or it == STEPS - 1:
            print(f"  [{act}] step {it:4d} loss {float(loss.detach()):.5f} " + " ".join(f"{k} {float(v.detach()):.5f}" for k, v in ld.items()))
    # the checkpoint: weights rounded to fp16-representable values (2 bytes per weight on disk)
    sd = {k: v.detach().to(torch.float16) for k, v in net.state_dict().items()}
    return sd, man, cand


def project_ref(net, q0, steps, snap_at=(1, 10)):
    """experiments/sample_poses.py:67-74 restated around the imported reference objects."""
    noisy = q0.clone()
    noisy.requires_grad = True
    trace, snaps = [], {}
    for it in range(steps):
        pred = net(noisy, train=False)
        grad = gradient(noisy, pred["dist_pred"]).reshape(-1, 84)
        noisy = (noisy - (pred["dist_pred"] * grad).reshape(-1, 21, 4)).detach()
        noisy.requires_grad = True
        trace.append(pred["dist_pred"].detach()[:, 0].clone())
        if it + 1 in snap_at:
            snaps[it + 1] = noisy.detach().clone()
    return snaps, torch.stack(trace)


def one(act):
    sd16, man, cand = train(act)
    out = {"w::" + k: v.numpy() for k, v in sd16.items()}          # fp16 arrays; the tests widen them to fp32
    out["hidden"] = np.array(HIDDEN, dtype=np.int32)
    rng = np.random.default_rng(42)
    near, near_dist, _ = make_batch(man, cand, 48, rng)            # the regime the network was trained for
    rand = synth.make_poses(44, seed=31)                           # the benchmark's input distribution (far from the manifold)
    edge = synth.make_poses(4, seed=13, signed=True)
    edge[0, :, 2] = 0.0
    edge[1] *= 1e-3
    edge[2, :, :] = edge[2, 0:1, :]
    edge[3, 5, :] = 0.0
    q_np = np.concatenate([near, rand, edge]).astype(np.float32)
    out["q"] = q_np
    out["label_near"] = near_dist
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        net = ref_model(act, dtype)
        net.load_state_dict({k: v.to(dtype) for k, v in sd16.items()})
        net.eval()
        q = torch.from_numpy(q_np).to(dtype)
        q.requires_grad = True
        pred = net(q, train=False)["dist_pred"]
        g = gradient(q, pred)
        out[f"d_{tag}"] = pred.detach().numpy()
        out[f"dq_{tag}"] = g.detach().numpy()
        if tag == "f32":
            go = torch.from_numpy(np.random.default_rng(5).normal(size=(q.shape[0], 1)).astype(np.float32))
            q2 = torch.from_numpy(q_np).clone().requires_grad_(True)
            (net(q2, train=False)["dist_pred"] * go).sum().backward()
            out["grad_out"] = go.numpy()
            out["grad_pose_f32"] = q2.grad.numpy()
        snaps, trace = project_ref(net, torch.from_numpy(q_np).to(dtype), 10)
        for k, v in snaps.items():
            out[f"q{k}_{tag}"] = v.numpy()
        out[f"dtrace_{tag}"] = trace.numpy()
    out["torch_version"] = np.array(torch.__version__)
    d = out["d_f32"][:48, 0]
    print(f"  [{act}] near-manifold poses: label mean {near_dist.mean():.4f}, predicted mean {d.mean():.4f}, "
          f"corr {np.corrcoef(near_dist, d)[0, 1]:.3f}; |grad| median "
          f"{np.median(np.linalg.norm(out['dq_f32'][:48].reshape(48, -1), axis=1)):.3f}")
    return out


if __name__ == "__main__":
    torch.set_num_threads(8)
    for act in (sys.argv[1:] or ("lrelu", "softplus")):
        res = one(act)
        path = os.path.join(HERE, f"trained_{act}.npz")
        np.savez_compressed(path, **res)
        print(path, os.path.getsize(path) // 1024, "KiB")
