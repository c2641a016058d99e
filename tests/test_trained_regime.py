"""A TRAINED network as a parity regime (tests/golden/trained_<act>.npz, make_golden_trained.py): the imported reference
trained its own DFNet -- built narrower than configs/amass.yaml through `model.DFNet.dims`, net_modules.py:14-28 -- on a
synthetic pose manifold, and then produced d, dd/dq, the autograd contract and 1/10-step projections in fp32 and fp64.
Every other golden file holds random-init weights of the amass.yaml widths.

CPU part: the numpy oracle against those vectors (pins the oracle on a narrower, trained network).
GPU part: both kernels through the facade / C ABI against them, with the per-pose gates of conftest.py."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, d_rows, fp32_noise, outlier_gate, pose_gate, rel_err_rows, traj_envelope, traj_margin

ACTS = ("lrelu", "softplus")
TOL = 1e-4
EDGE = {92: "zero component column (eps clamp of F.normalize)", 93: "tiny pose (scale invariance)",
        94: "all joints equal", 95: "one zero quaternion"}            # make_golden_trained.py:one


def load(act):
    g = dict(np.load(os.path.join(GOLDEN, f"trained_{act}.npz")))
    sd = {k[3:]: v.astype(np.float32) for k, v in g.items() if k.startswith("w::")}      # fp16 on disk = the checkpoint
    return g, sd, [int(h) for h in g["hidden"]]


@pytest.mark.parametrize("act", ACTS)
def test_trained_network_means_something(act):
    """the fixture is a trained distance field, not noise: near the manifold the prediction follows the label"""
    g, sd, hidden = load(act)
    assert [sd[f"dfnet.lin{l}.weight"].shape[0] for l in range(6)] == hidden and len(sd) == 98
    d = g["d_f32"][:48, 0]
    assert np.corrcoef(g["label_near"], d)[0, 1] > 0.7
    assert abs(d.mean() - g["label_near"].mean()) < 0.03
    assert g["dtrace_f32"][-1].mean() < 0.9 * g["dtrace_f32"][0].mean()      # projection steps walk towards the manifold


@pytest.mark.parametrize("act", ACTS)
def test_oracle_matches_reference_on_trained_network(act):
    from oracle import posendf_np as onp
    g, sd, _ = load(act)
    d64, g64 = onp.forward_grad(g["q"], sd, act, dtype=np.float64)
    assert np.abs(d64 - g["d_f64"].reshape(d64.shape)).max() < 1e-12
    ok = np.ones(len(g["q"]), bool)
    ok[92] = False                                  # eps-clamp pose: gradient ~1e10, compared relatively below
    assert np.abs(g64.reshape(len(ok), -1)[ok] - g["dq_f64"].reshape(len(ok), -1)[ok]).max() < 1e-10
    assert rel_err_rows(g64, g["dq_f64"]).max() < 1e-9
    d32, g32 = onp.forward_grad(g["q"], sd, act, dtype=np.float32)
    # fp32 oracle vs the reference's fp32 run: both within fp32 rounding of the fp64 truth
    assert np.median(d_rows(d32, g["d_f64"])) < 2e-6 and np.median(d_rows(g["d_f32"], g["d_f64"])) < 2e-6
    q10, _, trace = onp.project(g["q"], sd, steps=10, act=act, dtype=np.float64, trace=True)
    assert rel_err_rows(q10, g["q10_f64"]).max() < 1e-6          # kink-free in fp64 on both sides
    assert np.abs(np.asarray(trace) - g["dtrace_f64"]).max() < 1e-9


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    return torch


def make_net(torch, act, sd, hidden, precision):
    from posendf_amd import PoseNDF, amass_config
    cfg = amass_config(act, "cuda:0")
    cfg["model"]["DFNet"]["dims"] = hidden
    cfg["engine"] = {"precision": precision}
    net = PoseNDF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    return net


def kink_exempt(q, sd, act):
    from oracle import posendf_np as onp
    return None if act == "softplus" else onp.kink_margin(q, sd, act) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
@pytest.mark.parametrize("act", ACTS)
def test_trained_single_step(torch_cuda, act, precision):
    torch = torch_cuda
    g, sd, hidden = load(act)
    net = make_net(torch, act, sd, hidden, precision)
    q = torch.from_numpy(g["q"]).cuda().requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    (dq,) = torch.autograd.grad(d, q, grad_outputs=torch.ones_like(d))
    d_np, dq_np = d.detach().cpu().numpy(), dq.cpu().numpy()
    sig_d, sig_g, _, _ = fp32_noise(g["q"], sd, act, extra_d=[d_rows(g["d_f32"], g["d_f64"])],
                                    extra_g=[rel_err_rows(g["dq_f32"], g["dq_f64"])])
    e_d, e_g = d_rows(d_np, g["d_f64"]), rel_err_rows(dq_np, g["dq_f64"])
    ex = kink_exempt(g["q"], sd, act)
    pose_gate(e_d, sig_d, "d")
    pose_gate(e_g, sig_g, "dq", exempt=ex)
    for i, name in EDGE.items():
        assert e_d[i] <= 8 * sig_d[i] + 8e-6, (name, "d", e_d[i], sig_d[i])
        assert e_g[i] <= 8 * sig_g[i] + 8e-6 or (ex is not None and ex[i]), (name, "dq", e_g[i], sig_g[i])
        assert np.isfinite(d_np[i]).all() and np.isfinite(dq_np[i]).all(), name
    if act != "softplus":        # poses the reference clips in fp32 AND fp64 (on the manifold): exactly zero d and gradient
        z = (g["d_f32"][:, 0] == 0) & (g["d_f64"][:, 0] == 0)
        assert np.all(d_np[z, 0] == 0) and np.all(dq_np[z] == 0)
    # autograd contract with an arbitrary upstream gradient (motion_denoise.py:82-83,97-98)
    q2 = torch.from_numpy(g["q"]).cuda().requires_grad_(True)
    (net(q2, train=False)["dist_pred"] * torch.from_numpy(g["grad_out"]).cuda()).sum().backward()
    truth = g["dq_f64"] * g["grad_out"].reshape(-1, 1, 1)
    pose_gate(rel_err_rows(q2.grad.cpu().numpy(), truth), sig_g, "grad_pose", exempt=ex)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
@pytest.mark.parametrize("act", ACTS)
def test_trained_projection(torch_cuda, act, precision):
    """sample_poses.py:67-74 on the trained network: 1 and 10 steps against the reference's fp64 trajectory, within the
    envelope of the reference's own fp32 trajectory; distances after ten steps against the reference's trace"""
    torch = torch_cuda
    g, sd, hidden = load(act)
    net = make_net(torch, act, sd, hidden, precision)
    q0 = torch.from_numpy(g["q"]).cuda()
    for steps in (1, 10):
        qp, d_last = net.project(q0, steps=steps)
        ref = g["dtrace_f64"][steps - 1]
        dm = lambda a, t=None: np.abs(np.asarray(a, np.float64).reshape(-1) - ref) / np.maximum(np.abs(ref), 0.05 * np.abs(ref).max())
        env, env_d = traj_envelope(g["q"], sd, act, steps, g[f"q{steps}_f64"], truth_d=ref, d_metric=dm)
        outlier_gate(rel_err_rows(qp.cpu().numpy(), g[f"q{steps}_f64"]),
                     rel_err_rows(g[f"q{steps}_f32"], g[f"q{steps}_f64"]), TOL, f"q{steps}", **env)
        err = np.abs(d_last.cpu().numpy().reshape(-1) - ref) / np.maximum(np.abs(ref), 0.05 * np.abs(ref).max())
        ref32 = np.abs(g["dtrace_f32"][steps - 1] - ref) / np.maximum(np.abs(ref), 0.05 * np.abs(ref).max())
        outlier_gate(err, ref32, TOL, f"d_last{steps}", **env_d)


@pytest.mark.gpu
@pytest.mark.parametrize("act", ACTS)
def test_fp16_checkpoint_runs_two_term_kernel_bit_identically(torch_cuda, act, monkeypatch):
    """The trained fixture is a half-precision checkpoint: every weight is exactly representable in fp16, so the lo halves
    of the packed weights are all zero, the lo*hi term of the split arithmetic vanishes identically, and pndf_load_weights
    selects the two-term kernels.  Same results bit for bit as the three-term kernels (PNDF_THREE_TERMS=1); a network
    with ordinary fp32 weights stays on three terms."""
    torch = torch_cuda
    from posendf_amd import synth
    g, sd, hidden = load(act)
    q0 = torch.from_numpy(np.concatenate([g["q"], synth.make_poses(1000, seed=3, signed=True)])).cuda()

    def run(net):
        q = q0.clone().requires_grad_(True)
        d = net(q, train=False)["dist_pred"]
        (dq,) = torch.autograd.grad(d.sum(), q)
        qp, dl = net.project(q0, steps=10)
        return net._engine_for(q0.device).kernel_name(), d.detach(), dq, qp, dl

    name2, *out2 = run(make_net(torch, act, sd, hidden, "f16x3"))
    monkeypatch.setenv("PNDF_THREE_TERMS", "1")
    name3, *out3 = run(make_net(torch, act, sd, hidden, "f16x3"))
    monkeypatch.delenv("PNDF_THREE_TERMS")
    fam = "softplus" if act == "softplus" else "relu"
    assert name2 == f"pndf_fused_split2_{fam}_kernel" and name3 == f"pndf_fused_split_{fam}_kernel"
    for a, b in zip(out2, out3):
        assert torch.equal(a, b)
    # ordinary fp32 weights (lo halves non-zero): three terms; one weight off the fp16 grid is enough
    sd_off = {k: v.copy() for k, v in sd.items()}
    sd_off["dfnet.lin3.weight"][5, 7] += np.float32(2.0 ** -20)
    assert run(make_net(torch, act, sd_off, hidden, "f16x3"))[0] == f"pndf_fused_split_{fam}_kernel"
    # the choice follows the weights of the SAME module through reloads (engine re-packs on a changed fingerprint)
    net = make_net(torch, act, sd, hidden, "f16x3")
    assert run(net)[0] == f"pndf_fused_split2_{fam}_kernel"
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_off.items()})
    assert run(net)[0] == f"pndf_fused_split_{fam}_kernel"
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    assert run(net)[0] == f"pndf_fused_split2_{fam}_kernel"
    assert run(make_net(torch, act, sd, hidden, "fp32"))[0] == ("pndf_fused_softplus_kernel" if act == "softplus" else "pndf_fused_relu_kernel")


@pytest.mark.gpu
@pytest.mark.parametrize("act", ["lrelu", "relu", "softplus"])
def test_two_term_kernel_full_width_bit_identity(torch_cuda, act, monkeypatch):
    """amass.yaml widths, benchmark weights rounded to fp16, a ragged batch: two-term and three-term kernels agree bit for
    bit on d, dd/dq (with grad_outputs) and a 20-step projection, and both meet the oracle gate."""
    torch = torch_cuda
    from oracle import posendf_np as onp
    from posendf_amd import PoseNDF, amass_config, synth
    sd = {k: v.astype(np.float16).astype(np.float32) for k, v in synth.make_weights(2, 2.5, 0.05).items()}
    qn = synth.make_poses(4096 + 37, seed=9, signed=True)
    q0 = torch.from_numpy(qn).cuda()
    go = torch.from_numpy(np.random.default_rng(3).normal(size=(len(qn), 1)).astype(np.float32)).cuda()

    def run():
        cfg = amass_config(act, "cuda:0")
        cfg["engine"] = {"precision": "f16x3"}
        net = PoseNDF(cfg)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net.eval()
        q = q0.clone().requires_grad_(True)
        d = net(q, train=False)["dist_pred"]
        (dq,) = torch.autograd.grad(d, q, grad_outputs=go)
        qp, dl = net.project(q0, steps=20)
        q1 = q0[:512].clone().requires_grad_(True)
        (g1,) = torch.autograd.grad(net(q1, train=False)["dist_pred"].sum(), q1)
        return net._engine_for(q0.device).kernel_name(), d.detach(), dq, qp, dl, g1

    two = run()
    monkeypatch.setenv("PNDF_THREE_TERMS", "1")
    three = run()
    assert "split2" in two[0] and "split2" not in three[0]
    for a, b in zip(two[1:], three[1:]):
        assert torch.equal(a, b)
    sig_d, sig_g, d64, g64 = fp32_noise(qn[:512], sd, act)
    pose_gate(d_rows(two[1][:512].cpu().numpy(), d64), sig_d, "d")
    ex = None if act == "softplus" else onp.kink_margin(qn[:512], sd, act) < 1e-5
    pose_gate(rel_err_rows(two[5].cpu().numpy(), g64), sig_g, "dq", exempt=ex)


@pytest.mark.gpu
@pytest.mark.parametrize("encoder", [True, False], ids=["enc", "noenc"])
@pytest.mark.parametrize("act", ["lrelu", "softplus"])
def test_two_term_kernels_fetch_hi_tiles_only_same_bits(torch_cuda, act, encoder, monkeypatch):
    """Round 5: the two-term kernels do not FETCH the lo tiles of the trunk either (csrc/pndf_device.h PNDF_RING_PIECES), and
    complete the slots their look-ahead ran into the encoder's backward section / the next step's first slot once per step
    (ring_complete_lookahead).  The full-width network as a half-precision checkpoint, with and without the structure
    encoder, a batch of several workgroup rounds (the softplus kernels walk their blocks with a persistent grid: the ring
    restarts per block), 25 free-running steps: bit-identical to the three-term kernels, which fetch everything."""
    torch = torch_cuda
    from posendf_amd import PoseNDF, amass_config, synth
    sd = {k: v.astype(np.float16).astype(np.float32) for k, v in synth.make_weights(5, 2.0, 0.1).items()}
    if not encoder:
        sd = {k: v for k, v in sd.items() if k.startswith("dfnet.")}
        rng = np.random.default_rng(5)
        sd["dfnet.lin0.weight"] = rng.uniform(-0.2, 0.2, (256, 84)).astype(np.float16).astype(np.float32)
    q0 = torch.from_numpy(synth.make_poses(256 * 64 + 37, seed=9)).cuda()          # > one round of workgroups, ragged tail

    def run():
        cfg = amass_config(act, "cuda:0")
        cfg["model"]["StrEnc"]["use"] = encoder
        if not encoder:
            cfg["model"]["DFNet"]["in_dim"] = 84
        cfg["engine"] = {"precision": "f16x3"}
        net = PoseNDF(cfg)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net.eval()
        qp, dl = net.project(q0, steps=25)
        q = q0[:1000].clone().requires_grad_(True)
        d = net(q, train=False)["dist_pred"]
        (dq,) = torch.autograd.grad(d.sum(), q)
        return net._engine_for(q0.device).kernel_name(), qp, dl, d.detach(), dq

    name2, *out2 = run()
    monkeypatch.setenv("PNDF_THREE_TERMS", "1")
    name3, *out3 = run()
    monkeypatch.delenv("PNDF_THREE_TERMS")
    fam = "softplus" if act == "softplus" else "relu"
    assert name2 == f"pndf_fused_split2_{fam}_kernel" and name3 == f"pndf_fused_split_{fam}_kernel"
    for a, b in zip(out2, out3):
        assert torch.isfinite(a).all() and torch.equal(a, b)
