"""Encoder-less configuration (model.StrEnc.use = False, DFNet in_dim = 84; reference model/posendf.py:40-42,73-74):
oracle against vectors produced by the reference itself; HIP engine against the oracle / the vectors."""
import os

import numpy as np
import pytest

from conftest import d_err, fp32_noise, outlier_gate, rel_err_rows, traj_envelope, traj_margin
from posendf_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-4


def golden(act):
    return np.load(os.path.join(HERE, "golden", f"posendf_noenc_{act}_live.npz"))


def weights():
    return synth.make_weights(seed=0, gain=2.0, out_bias=0.1, dims=synth.DFNET_DIMS_NOENC)


@pytest.mark.parametrize("act", ["lrelu", "softplus"])
def test_oracle_matches_reference_vectors(act):
    from oracle import posendf_np as onp
    g, sd = golden(act), weights()
    assert len(sd) == 14 and not onp.has_encoder(sd)
    d, dq = onp.forward_grad(g["q"], sd, act)
    assert d_err(d, g["d_f32"]) < 2e-5               # fp32 vs fp32: summation order of the BLAS differs
    assert np.median(rel_err_rows(dq, g["dq_f32"])) < 2e-5
    d64, dq64 = onp.forward_grad(g["q"], sd, act, dtype=np.float64)
    assert d_err(d64, g["d_f64"]) < 1e-12
    assert np.max(rel_err_rows(dq64, g["dq_f64"])) < 1e-9
    q10, _ = onp.project(g["q"], sd, steps=10, act=act, dtype=np.float64)
    assert np.median(rel_err_rows(q10, g["q10_f64"])) < 1e-10


def make_net(torch, act, precision):
    from posendf_amd import PoseNDF, amass_config
    cfg = amass_config(act, "cuda:0")
    cfg["model"]["StrEnc"]["use"] = False
    cfg["model"]["DFNet"]["in_dim"] = 84
    cfg["engine"] = {"precision": precision}
    net = PoseNDF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in weights().items()})
    net.eval()
    return net


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
@pytest.mark.parametrize("act", ["lrelu", "softplus"])
def test_engine_matches_reference_vectors(act, precision):
    import torch
    g = golden(act)
    net = make_net(torch, act, precision)
    assert list(net.state_dict().keys()) == list(weights().keys())
    q = torch.from_numpy(g["q"]).cuda().requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    (dq,) = torch.autograd.grad(d, q, grad_outputs=torch.ones_like(d))
    assert d_err(d.detach().cpu().numpy(), g["d_f32"]) < TOL
    outlier_gate(rel_err_rows(dq.cpu().numpy(), g["dq_f64"]), rel_err_rows(g["dq_f32"], g["dq_f64"]), TOL, "dq",
                 margin=traj_margin(g["q"], weights(), act), sigma=fp32_noise(g["q"], weights(), act)[1])
    for steps in (1, 10, 100):
        qp, _ = net.project(q.detach(), steps=steps)
        truth = g[f"q{steps}_f64"]
        outlier_gate(rel_err_rows(qp.cpu().numpy(), truth), rel_err_rows(g[f"q{steps}_f32"], truth), TOL, f"project{steps}",
                     **traj_envelope(g["q"], weights(), act, steps, truth))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_engine_ragged_batch_matches_oracle(precision):
    import torch
    from oracle import posendf_np as onp
    sd = weights()
    net = make_net(torch, "lrelu", precision)
    q_np = synth.make_poses(333, seed=41, signed=True)
    qp, dl = net.project(torch.from_numpy(q_np).cuda(), steps=3)
    qo, do = onp.project(q_np, sd, steps=3, dtype=np.float64)
    assert np.median(rel_err_rows(qp.cpu().numpy(), qo)) < TOL / 10
    assert d_err(dl.cpu().numpy(), do) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("act,precision", [("lrelu", "f16x3"), ("softplus", "f16x3"), ("softplus", "fp32")])
def test_noenc_narrower_combination(act, precision):
    """encoder-less AND narrower than amass.yaml at once (in_dim 84, hidden 100-300-520-77-130-33): both are runtime
    variations of the same kernels; against the fp64 oracle with the per-pose gates"""
    import torch
    from conftest import d_rows, fp32_noise, pose_gate
    from oracle import posendf_np as onp
    from posendf_amd import PoseNDF, amass_config
    hidden = [100, 300, 520, 77, 130, 33]
    sd = synth.make_weights(seed=8, gain=2.0, out_bias=0.1, dims=(84, *hidden, 1))
    cfg = amass_config(act, "cuda:0")
    cfg["model"]["StrEnc"]["use"] = False
    cfg["model"]["DFNet"]["in_dim"] = 84
    cfg["model"]["DFNet"]["dims"] = hidden
    cfg["engine"] = {"precision": precision}
    net = PoseNDF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    qn = synth.make_poses(300, seed=49, signed=True)
    q = torch.from_numpy(qn).cuda().requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    (dq,) = torch.autograd.grad(d.sum(), q)
    sig_d, sig_g, d64, g64 = fp32_noise(qn, sd, act)
    ex = None if act == "softplus" else onp.kink_margin(qn, sd, act) < 1e-5
    pose_gate(d_rows(d.detach().cpu().numpy(), d64), sig_d, "d")
    pose_gate(rel_err_rows(dq.cpu().numpy(), g64), sig_g, "dq", exempt=ex)
    qp, _ = net.project(q.detach(), steps=5)
    q64, _ = onp.project(qn, sd, steps=5, act=act, dtype=np.float64)
    q32, _ = onp.project(qn, sd, steps=5, act=act)
    outlier_gate(rel_err_rows(qp.cpu().numpy(), q64), rel_err_rows(q32, q64), TOL, "project5", **traj_envelope(qn, sd, act, 5, q64))
