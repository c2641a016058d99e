"""The host twins `pndf_*_cpu` (posendf_amd/csrc/pndf_cpu.cpp) through the C ABI and through the facade with
`train.device: cpu`, against the vectors the reference itself produced (tests/golden) and the same per-pose gates as the HIP
kernels.  Runs without a GPU: this is the part of the C ABI's arithmetic the CPU suite can check end to end."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import (ALL_REGIMES, REGIMES, d_err, d_rows, fp32_noise, golden_weights, load_golden, outlier_gate, pose_gate,
                      rel_err_rows, traj_envelope, traj_margin)
from posendf_amd import synth

TOL = 1e-4


def make_net(act, sd=None, regime="live", noenc=False, hidden=None):
    from posendf_amd import PoseNDF, amass_config
    cfg = amass_config(act, "cpu")
    if noenc:
        cfg["model"]["StrEnc"]["use"] = False
        cfg["model"]["DFNet"]["in_dim"] = 84
    if hidden is not None:
        cfg["model"]["DFNet"]["dims"] = list(hidden)
    net = PoseNDF(cfg)
    sd = sd if sd is not None else golden_weights(regime)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    return net


@pytest.mark.parametrize("regime", list(ALL_REGIMES))
@pytest.mark.parametrize("act", ["lrelu", "relu", "softplus"])
def test_golden_single_step_on_the_host(act, regime):
    from oracle import posendf_np as onp
    g, sd = load_golden(act, regime), golden_weights(regime)
    net = make_net(act, sd)
    q = torch.from_numpy(g["q"]).requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    assert d.shape == (len(g["q"]), 1) and d.device.type == "cpu"
    assert net._engine_for(q.device).kernel_name() == "pndf_cpu (host twin)"
    (dq,) = torch.autograd.grad(d, q, grad_outputs=torch.ones_like(d))
    d_np, dq_np = d.detach().numpy(), dq.numpy()
    if regime in REGIMES:
        assert d_err(d_np, g["d_f32"]) < TOL
    sig_d, sig_g, _, _ = fp32_noise(g["q"], sd, act, extra_d=[d_rows(g["d_f32"], g["d_f64"])],
                                    extra_g=[rel_err_rows(g["dq_f32"], g["dq_f64"])])
    ex = None if act == "softplus" else onp.kink_margin(g["q"], sd, act) < 1e-5
    pose_gate(d_rows(d_np, g["d_f64"]), sig_d, "d host")
    pose_gate(rel_err_rows(dq_np, g["dq_f64"]), sig_g, "dq host", exempt=ex)
    with torch.no_grad():      # forward-only entry point: the same distances
        assert torch.equal(net(torch.from_numpy(g["q"]), train=False)["dist_pred"], d.detach())
    # arbitrary upstream gradient (motion_denoise.py:82-83,97-98)
    if "grad_pose_f32" in g:
        q2 = torch.from_numpy(g["q"]).requires_grad_(True)
        (net(q2, train=False)["dist_pred"] * torch.from_numpy(g["grad_out"])).sum().backward()
        truth = g["dq_f64"] * g["grad_out"].reshape(-1, 1, 1)
        ref_rows = rel_err_rows(g["grad_pose_f32"], truth)
        outlier_gate(rel_err_rows(q2.grad.numpy(), truth), ref_rows, TOL, "grad_out host", margin=traj_margin(g["q"], sd, act), sigma=sig_g)


@pytest.mark.parametrize("act", ["lrelu", "softplus"])
def test_golden_projection_on_the_host(act):
    g, sd = load_golden(act, "live"), golden_weights("live")
    net = make_net(act, sd)
    for steps in (1, 10):
        qp, dl = net.project(torch.from_numpy(g["q"]), steps=steps)
        env = traj_envelope(g["q"], sd, act, steps, g[f"q{steps}_f64"])
        outlier_gate(rel_err_rows(qp.numpy(), g[f"q{steps}_f64"]), rel_err_rows(g[f"q{steps}_f32"], g[f"q{steps}_f64"]), TOL,
                     f"q{steps} host", **env)
    # the fused loop equals the caller's own loop around forward + gradient bit for bit (two roundings of q - d * grad)
    q = torch.from_numpy(g["q"][:40])
    mine, _ = net.project(q, steps=3)
    cur = q.clone()
    for _ in range(3):
        cur = cur.detach().requires_grad_(True)
        d = net(cur, train=False)["dist_pred"]
        (gr,) = torch.autograd.grad(d, cur, grad_outputs=torch.ones_like(d))
        cur = cur - (d * gr.reshape(-1, 84)).reshape(-1, 21, 4)
    assert torch.equal(mine, cur.detach())
    same, dl0 = net.project(q, steps=0)
    assert torch.equal(same, q) and torch.all(dl0 == 0)


def test_host_twin_configurations_and_refusals():
    from oracle import posendf_np as onp
    from posendf_amd.engine import CpuEngine, PndfError, load_library
    # encoder-less
    sd = synth.make_weights(seed=0, gain=2.0, out_bias=0.1, dims=synth.DFNET_DIMS_NOENC)
    q = synth.make_poses(70, seed=3)            # ragged: 2 blocks of 32 + 6
    net = make_net("lrelu", sd, noenc=True)
    t = torch.from_numpy(q).requires_grad_(True)
    d = net(t, train=False)["dist_pred"]
    (g,) = torch.autograd.grad(d.sum(), t)
    d64, g64 = onp.forward_grad(q, sd, "lrelu", dtype=np.float64)
    assert d_err(d.detach().numpy(), d64) < 2e-5 and np.median(rel_err_rows(g.numpy(), g64)) < 1e-5
    # narrower hidden layers (model.DFNet.dims)
    dims = (126, 192, 384, 700, 300, 200, 48, 1)
    sd = synth.make_weights(5, 1.5, 0.1, dims=dims)
    net = make_net("softplus", sd, hidden=dims[1:-1])
    d = net(torch.from_numpy(q), train=False)["dist_pred"]
    d64, _ = onp.forward_grad(q, sd, "softplus", dtype=np.float64)
    assert d_err(d.detach().numpy(), d64) < 2e-5
    # thread count does not change a bit (poses are independent, blocks are fixed)
    import os
    ref = net.project(torch.from_numpy(q), steps=2)[0]
    for n in ("1", "3"):
        os.environ["PNDF_CPU_THREADS"] = n
        try:
            assert torch.equal(net.project(torch.from_numpy(q), steps=2)[0], ref)
        finally:
            del os.environ["PNDF_CPU_THREADS"]
    # NaN stays in its row, relu(NaN) = NaN
    bad = torch.from_numpy(q.copy())
    bad[37, 5, 2] = float("nan")
    dn = make_net("lrelu")(bad, train=False)["dist_pred"]
    assert torch.isnan(dn[37]).all() and torch.isfinite(dn[torch.arange(70) != 37]).all()
    # refusals: compute before weights, wrong tensor count, empty batch is a no-op
    eng = CpuEngine("lrelu")
    buf = np.zeros((2, 84), np.float32)
    out = np.zeros(2, np.float32)
    with pytest.raises(PndfError):
        eng.forward(buf.ctypes.data, out.ctypes.data, 2)
    with pytest.raises(PndfError):
        eng.load_weights(synth.make_weights(0, 2.0, 0.1, dims=synth.DFNET_DIMS_NOENC))
    eng.load_weights(golden_weights("live"))
    eng.forward(buf.ctypes.data, out.ctypes.data, 0)
    # a load that fails half way (one tensor of the wrong size) leaves no half-loaded engine behind
    bad_sd = dict(golden_weights("live"))
    bad_sd["dfnet.lin3.bias"] = bad_sd["dfnet.lin3.bias"][:-1]
    with pytest.raises(PndfError):
        eng.load_weights(bad_sd)
    with pytest.raises(PndfError):
        eng.forward(buf.ctypes.data, out.ctypes.data, 2)
    eng.load_weights(golden_weights("live"))
    lib = load_library()
    assert lib.pndf_forward_cpu(eng.handle, None, out.ctypes.data, 2) == -1
    with pytest.raises(PndfError):
        CpuEngine("lrelu", hidden=[256, 512, 2048, 512, 256, 64])       # wider than 1024: refused like pndf_create


def test_reference_projection_loop_runs_unchanged_on_cpu():
    """experiments/sample_poses.py:67-74 verbatim around the facade on a CPU config (the reference's class runs there)."""
    from posendf_amd import gradient
    g, sd = load_golden("lrelu", "live"), golden_weights("live")
    net = make_net("lrelu", sd)
    noisy_poses = torch.from_numpy(g["q"])
    for _ in range(10):
        noisy_poses = noisy_poses.detach()
        noisy_poses.requires_grad = True
        net_pred = net(noisy_poses, train=False)
        grad_val = gradient(noisy_poses, net_pred["dist_pred"]).reshape(-1, 84)
        noisy_poses = noisy_poses.detach()
        noisy_poses = noisy_poses.reshape(-1, 84) - (net_pred["dist_pred"] * grad_val)
        noisy_poses = noisy_poses.reshape(-1, 21, 4)
    env = traj_envelope(g["q"], sd, "lrelu", 10, g["q10_f64"])
    outlier_gate(rel_err_rows(noisy_poses.detach().numpy(), g["q10_f64"]), rel_err_rows(g["q10_f32"], g["q10_f64"]), TOL, "loop host", **env)


def test_motion_denoise_loop_on_a_cpu_config():
    """experiments/motion_denoise.py's loop (autograd driver) around a `train.device: cpu` model: the pose prior runs on the host
    twins; same poses as the loop around the PyTorch restatement of the reference; the fused HIP driver refuses host poses."""
    from test_motion_denoise import _OraclePrior, _noisy_sequences
    from posendf_amd.motion_denoise import MotionDenoise
    sd = golden_weights("live")
    noisy = _noisy_sequences(2, 8, seed=5)
    got, h = MotionDenoise(make_net("lrelu", sd), device="cpu").denoise(noisy, iterations=2, steps_per_iter=4)
    ref, h_ref = MotionDenoise(_OraclePrior("lrelu", sd), device="cpu").denoise(noisy, iterations=2, steps_per_iter=4)
    assert abs(h[0]["pose_pr"] - h_ref[0]["pose_pr"]) < 1e-5 * abs(h_ref[0]["pose_pr"])
    err = (got - ref).abs()
    assert err.median().item() < 1e-5 and err.max().item() < 5e-3, (err.median().item(), err.max().item())
    with pytest.raises(ValueError):
        MotionDenoise(make_net("lrelu", sd), device="cpu").denoise(noisy, fused=True)
