"""GPU parity tests (run by the driver with -m gpu on a real MI355X).  Everything goes through the C ABI
(posendf_amd.engine -> libposendf_amd.so); the checker is the numpy oracle and the committed golden
vectors generated from the reference.  Tolerance: 1e-4 relative (BASELINE.json north_star), with the
distance floor of conftest.d_err."""
import numpy as np
import pytest

from conftest import (ALL_REGIMES, LITE_REGIMES, REGIMES, d_err, d_rows, fp32_noise, golden_weights, load_golden, outlier_gate,
                      pose_gate, rel_err, rel_err_rows, traj_envelope, traj_margin)

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    return torch


PRECISIONS = ["fp32", "f16x3"]      # f16x3: fp16 hi/lo split operands, fp32 accumulate


def cases(acts):
    return [(a, p) for a in acts for p in PRECISIONS]


def make_net(torch, act, regime=None, sd=None, precision="fp32"):
    from posendf_amd import PoseNDF, amass_config
    cfg = amass_config(act, "cuda:0")
    cfg["engine"] = {"precision": precision}
    net = PoseNDF(cfg)
    sd = sd if sd is not None else golden_weights(regime)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    return net


ALL_ACTS = ["lrelu", "relu", "softplus"]


EDGE_POSES = {96: "zero component column (eps clamp of F.normalize)", 97: "tiny pose (scale invariance)",
              98: "all joints equal", 99: "one zero quaternion"}      # tests/golden/make_golden.py:make_inputs


def kink_exempt(q, sd, act):
    """relu family: poses with a pre-activation within 1e-5 (relative) of a kink in the fp64 run may flip a derivative;
    everything else -- and every softplus pose -- is held to the per-pose fp32 sensitivity (conftest.pose_gate)."""
    from oracle import posendf_np as onp
    return None if act == "softplus" else onp.kink_margin(q, sd, act) < 1e-5


@pytest.mark.parametrize("act,precision", cases(ALL_ACTS))
@pytest.mark.parametrize("regime", list(ALL_REGIMES))
def test_golden_single_step(torch_cuda, act, regime, precision):
    torch = torch_cuda
    g = load_golden(act, regime)
    sd = golden_weights(regime)
    net = make_net(torch, act, regime, precision=precision)
    q = torch.from_numpy(g["q"]).cuda().requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    assert d.shape == (len(g["q"]), 1)
    (dq,) = torch.autograd.grad(d, q, grad_outputs=torch.ones_like(d))
    d_np, dq_np = d.detach().cpu().numpy(), dq.cpu().numpy()
    if regime in REGIMES:
        assert d_err(d_np, g["d_f32"]) < TOL         # the headline bar against the reference's fp32 output
    # every pose individually against the reference's fp64 run, within the fp32 sensitivity of the reference arithmetic
    # at that pose (the reference's own fp32 output is one of the samples that define it)
    sig_d, sig_g, _, _ = fp32_noise(g["q"], sd, act, extra_d=[d_rows(g["d_f32"], g["d_f64"])],
                                    extra_g=[rel_err_rows(g["dq_f32"], g["dq_f64"])])
    e_d, e_g = d_rows(d_np, g["d_f64"]), rel_err_rows(dq_np, g["dq_f64"])
    ex = kink_exempt(g["q"], sd, act)
    pose_gate(e_d, sig_d, "d")
    pose_gate(e_g, sig_g, "dq", exempt=ex)
    for i, name in EDGE_POSES.items():               # the edge poses, by name
        assert e_d[i] <= 8 * sig_d[i] + 8e-6, (name, "d", e_d[i], sig_d[i])
        assert e_g[i] <= 8 * sig_g[i] + 8e-6 or (ex is not None and ex[i]), (name, "dq", e_g[i], sig_g[i])
        assert np.isfinite(d_np[i]).all() and np.isfinite(dq_np[i]).all(), name
    outlier_gate(e_g, rel_err_rows(g["dq_f32"], g["dq_f64"]), TOL, "dq", margin=traj_margin(g["q"], sd, act), sigma=sig_g)
    # forward-only launch gives the same distances as the forward+grad launch
    with torch.no_grad():
        d2 = net(torch.from_numpy(g["q"]), train=False)["dist_pred"]      # CPU tensor is moved (posendf.py:64)
    assert torch.equal(d2, d.detach())
    if act != "softplus":
        # clipped poses: exactly zero distance and exactly zero gradient -- wherever the reference's fp32 AND fp64 runs
        # agree that the pose is clipped (a pre-activation of lin6 within rounding of 0 may go either way)
        z = (g["d_f32"][:, 0] == 0) & (g["d_f64"][:, 0] == 0)
        nz = (g["d_f32"][:, 0] > 0) & (g["d_f64"][:, 0] > 0)
        assert np.all(d_np[z, 0] == 0) and np.all(dq_np[z] == 0)
        if regime in REGIMES:
            assert np.all(d_np[nz, 0] > 0)


@pytest.mark.parametrize("act,precision", cases(ALL_ACTS))
def test_live_regime_regression_tripwires(torch_cuda, act, precision):
    """Plain absolute assertions where the achieved error sits far inside the bar.  The envelope gates
    above are calibrated on the engine's own sweep and would let a 10x regression on well-conditioned poses through; in the
    benchmark's `live` regime a single step is good to ~2e-6 (the reference arithmetic's own fp32 run: max d 1.7e-6, median
    d d/d q 0.6 - 1.4e-6, max 1.5 - 4.4e-6), so: max d error <= 2e-5, median d d/d q error <= 5e-6, largest d d/d q error away
    from a kink <= 5e-5, and no pose closer than 0.9 to its per-pose gate -- for both kernels and all three activations."""
    torch = torch_cuda
    g = load_golden(act, "live")
    sd = golden_weights("live")
    net = make_net(torch, act, "live", precision=precision)
    q = torch.from_numpy(g["q"]).cuda().requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    (dq,) = torch.autograd.grad(d, q, grad_outputs=torch.ones_like(d))
    e_d, e_g = d_rows(d.detach().cpu().numpy(), g["d_f64"]), rel_err_rows(dq.cpu().numpy(), g["dq_f64"])
    ex = kink_exempt(g["q"], sd, act)
    smooth = np.ones(len(e_g), bool) if ex is None else ~ex
    print(f"[tripwire {act} {precision}] max d {e_d.max():.2e} (<= 2e-5)  median dq {np.median(e_g):.2e} (<= 5e-6)  "
          f"max dq off-kink {e_g[smooth].max():.2e} (<= 5e-5)")
    assert e_d.max() <= 2e-5, ("d", float(e_d.max()), int(e_d.argmax()))
    assert np.median(e_g) <= 5e-6, ("median dq", float(np.median(e_g)))
    assert e_g[smooth].max() <= 5e-5, ("dq off-kink", float(e_g[smooth].max()), int(np.flatnonzero(smooth)[e_g[smooth].argmax()]))
    sig_d, sig_g, _, _ = fp32_noise(g["q"], sd, act, extra_d=[d_rows(g["d_f32"], g["d_f64"])],
                                    extra_g=[rel_err_rows(g["dq_f32"], g["dq_f64"])])
    assert pose_gate(e_d, sig_d, "d tripwire") <= 0.9
    assert pose_gate(e_g, sig_g, "dq tripwire", exempt=ex) <= 0.9


@pytest.mark.parametrize("act,precision", cases(ALL_ACTS))
@pytest.mark.parametrize("regime", ["mixed", "s2g3", "s4g25", "s1g1"])
def test_golden_autograd_contract(torch_cuda, act, precision, regime):
    """backward with an arbitrary upstream gradient (motion_denoise.py:82-83,97-98) and the pose-prior
    objective 1e7 c^2 / (1 + it) of motion_denoise.py:33."""
    torch = torch_cuda
    g = load_golden(act, regime)
    sd = golden_weights(regime)
    net = make_net(torch, act, regime, precision=precision)
    q = torch.from_numpy(g["q"]).cuda().requires_grad_(True)
    (net(q, train=False)["dist_pred"] * torch.from_numpy(g["grad_out"]).cuda()).sum().backward()
    truth = g["dq_f64"] * g["grad_out"].reshape(-1, 1, 1)
    ref_rows = rel_err_rows(g["grad_pose_f32"], truth)
    margin = traj_margin(g["q"], sd, act)
    _, sig_g, _, _ = fp32_noise(g["q"], sd, act, extra_g=[ref_rows])
    outlier_gate(rel_err_rows(q.grad.cpu().numpy(), truth), ref_rows, TOL, "grad_out", margin=margin, sigma=sig_g)
    pose_gate(rel_err_rows(q.grad.cpu().numpy(), truth), sig_g, "grad_out", exempt=kink_exempt(g["q"], sd, act))
    for it in (() if regime in LITE_REGIMES else (0, 3)):
        q = torch.from_numpy(g["q"]).cuda().requires_grad_(True)
        c = torch.mean(net(q, train=False)["dist_pred"])
        obj = 10.0 ** 7 * c * c / (1 + it)
        obj.backward()
        assert abs(obj.item() - g[f"prior_obj_it{it}"]) <= TOL * abs(g[f"prior_obj_it{it}"])
        scale = 2e7 * g["d_f64"].mean() / ((1 + it) * len(g["q"]))
        truth = g["dq_f64"] * scale
        outlier_gate(rel_err_rows(q.grad.cpu().numpy(), truth), rel_err_rows(g[f"prior_grad_it{it}"], truth),
                     2 * TOL, "prior", margin=margin, sigma=sig_g)


@pytest.mark.parametrize("act,precision", cases(ALL_ACTS))
@pytest.mark.parametrize("regime", list(ALL_REGIMES))
def test_golden_projection(torch_cuda, act, regime, precision):
    """1/10/100-step projection vs the reference.  Single step: 1e-4.  Free-running: measured against the
    reference's fp64 trajectory with the reference's own fp32 run as the envelope (LeakyReLU/ReLU kinks make
    a per-pose 1e-4 gate fail for the reference against itself)."""
    torch = torch_cuda
    g = load_golden(act, regime)
    net = make_net(torch, act, regime, precision=precision)
    q0 = torch.from_numpy(g["q"]).cuda()
    for steps in ((1, 10) if regime in LITE_REGIMES else (1, 10, 100)):
        qp, dl = net.project(q0, steps=steps)
        qp = qp.cpu().numpy()
        truth = g[f"q{steps}_f64"]
        mine = rel_err_rows(qp, truth)
        ref = rel_err_rows(g[f"q{steps}_f32"], truth)
        dref64 = g["dtrace_f64"][steps - 1]
        floor = max(0.05 * np.abs(dref64).max(), 1e-30)       # s2g3: every pose is clipped (d == 0) after a few steps
        derr = lambda a, t=None: np.abs(np.asarray(a, np.float64).reshape(-1) - dref64) / np.maximum(np.abs(dref64), floor)
        env, env_d = traj_envelope(g["q"], golden_weights(regime), act, steps, truth, truth_d=dref64, d_metric=derr)
        outlier_gate(mine, ref, TOL, f"project{steps}", **env)       # includes BASELINE.md section 5's p95 gate
        if regime in REGIMES:      # the original headline statement: 90 % of the poses inside the bar, unconditionally
            assert np.percentile(mine, 90) < TOL, (steps, float(np.percentile(mine, 90)))
        # d_last is dist_pred of the last iteration (before its update); along a free-running trajectory it
        # is subject to the same kink divergence as q, so it gets the same outlier gate
        if steps == 1 and regime in REGIMES:
            assert d_err(dl.cpu().numpy()[:, 0], g["dtrace_f32"][0]) < TOL
        elif steps == 1:
            sig_d, _, _, _ = fp32_noise(g["q"], golden_weights(regime), act, extra_d=[d_rows(g["dtrace_f32"][0], dref64)])
            pose_gate(d_rows(dl.cpu().numpy()[:, 0], dref64), sig_d, "d_last1")
        else:
            outlier_gate(derr(dl.cpu().numpy()[:, 0]), derr(g["dtrace_f32"][steps - 1]), 20 * TOL, f"d_last{steps}", **env_d)


@pytest.mark.parametrize("act,precision", cases(["lrelu", "softplus"]))
@pytest.mark.parametrize("B", [1, 15, 63, 64, 65, 257, 1000])
def test_ragged_batches_match_oracle(torch_cuda, B, act, precision):
    torch = torch_cuda
    from oracle import posendf_np as onp
    from posendf_amd import synth
    sd = golden_weights("mixed")
    net = make_net(torch, act, sd=sd, precision=precision)
    qn = synth.make_poses(B, seed=100 + B, signed=True)
    q = torch.from_numpy(qn).cuda().requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    (dq,) = torch.autograd.grad(d.sum(), q)
    do, go = onp.forward_grad(qn, sd, act)
    _, g64 = onp.forward_grad(qn, sd, act, dtype=np.float64)
    assert d_err(d.detach().cpu().numpy(), do) < TOL
    outlier_gate(rel_err_rows(dq.cpu().numpy(), g64), rel_err_rows(go, g64), TOL, "dq", margin=traj_margin(qn, sd, act),
                 sigma=fp32_noise(qn, sd, act)[1])
    qp, _ = net.project(q.detach(), steps=4)
    q64, _ = onp.project(qn, sd, steps=4, act=act, dtype=np.float64)
    q32, _ = onp.project(qn, sd, steps=4, act=act)
    outlier_gate(rel_err_rows(qp.cpu().numpy(), q64), rel_err_rows(q32, q64), TOL, "project4", **traj_envelope(qn, sd, act, 4, q64))


@pytest.mark.parametrize("precision", PRECISIONS)
def test_teacher_forced_steps(torch_cuda, precision):
    """Per-step parity along the reference's own fp32 trajectory (BASELINE.md gate 2): feed q_k of the
    oracle trajectory, compare one engine step with the oracle's next iterate."""
    torch = torch_cuda
    from oracle import posendf_np as onp
    from posendf_amd import synth
    sd = golden_weights("live")
    net = make_net(torch, "lrelu", sd=sd, precision=precision)
    q = synth.make_poses(128, seed=9)
    for k in range(12):
        d, dq = onp.forward_grad(q, sd, "lrelu")
        nxt = q - (d * dq.reshape(-1, 84)).reshape(-1, 21, 4)
        d64, dq64 = onp.forward_grad(q, sd, "lrelu", dtype=np.float64)
        nxt64 = q.astype(np.float64) - (d64 * dq64.reshape(-1, 84)).reshape(-1, 21, 4)
        got, dl = net.project(torch.from_numpy(q), steps=1)
        outlier_gate(rel_err_rows(got.cpu().numpy(), nxt64), rel_err_rows(nxt, nxt64), TOL, f"step{k}", margin=traj_margin(q, sd, "lrelu"))
        assert d_err(dl.cpu().numpy(), d) < TOL, k
        q = nxt.astype(np.float32)


@pytest.mark.parametrize("act,precision", cases(["lrelu", "softplus"]))
def test_full_size_properties(torch_cuda, act, precision):
    """BASELINE.json configs 2-3 size (B = 65,536): size-independent properties instead of an oracle run.  Softplus runs
    as a persistent grid (one workgroup per CU walks four 64-pose blocks and re-uses its derivative scratch)."""
    torch = torch_cuda
    from oracle import posendf_np as onp
    from posendf_amd import synth
    sd = golden_weights("live")
    net = make_net(torch, act, sd=sd, precision=precision)
    B = 65536
    qn = synth.make_poses(B, seed=1234)
    q = torch.from_numpy(qn).cuda()
    # (1) per-pose independence: a permuted batch gives the permuted result, bit for bit
    perm = torch.randperm(B, device="cuda", generator=torch.Generator("cuda").manual_seed(0))
    q10, d10 = net.project(q, steps=10)
    q10p, d10p = net.project(q[perm].contiguous(), steps=10)
    assert torch.equal(q10[perm], q10p) and torch.equal(d10[perm], d10p)
    # (2) determinism and in-place operation
    q10b, _ = net.project(q, steps=10)
    assert torch.equal(q10, q10b)
    buf = q.clone()
    eng = net._engine_for(q.device)
    eng.project(buf.data_ptr(), buf.data_ptr(), 0, B, 10, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(buf, q10)
    # (3) composition: 10 steps == 4 steps followed by 6 steps
    q4, _ = net.project(q, steps=4)
    q46, _ = net.project(q4, steps=6)
    assert torch.equal(q46, q10)
    # (4) a random sample of the full-size result against the oracle
    idx = np.random.default_rng(0).choice(B, 256, replace=False)
    q64, _ = onp.project(qn[idx], sd, steps=10, act=act, dtype=np.float64)
    q32, _ = onp.project(qn[idx], sd, steps=10, act=act)
    outlier_gate(rel_err_rows(q10[idx].cpu().numpy(), q64), rel_err_rows(q32, q64), TOL, "project10", **traj_envelope(qn[idx], sd, act, 10, q64))
    # (5) the projection decreases the predicted distance on average (it is a descent on d^2 / 2)
    d0 = net(q, train=False)["dist_pred"]
    q100, dl100 = net.project(q, steps=100)
    d100 = net(q100, train=False)["dist_pred"]
    assert d100.mean() < d0.mean()
    # (5b) BASELINE.json configs[2] itself -- the FULL 100 steps at B = 65,536 -- against the oracle on a 256-pose sample:
    # fp64 trajectory as truth, the reference arithmetic's own fp32 trajectory as the envelope, every outlier explained
    idx100 = np.random.default_rng(1).choice(B, 256, replace=False)
    q64, d64 = onp.project(qn[idx100], sd, steps=100, act=act, dtype=np.float64)
    q32, d32 = onp.project(qn[idx100], sd, steps=100, act=act)
    env100, envd100 = traj_envelope(qn[idx100], sd, act, 100, q64, truth_d=d64.reshape(-1))
    outlier_gate(rel_err_rows(q100[idx100].cpu().numpy(), q64), rel_err_rows(q32, q64), TOL, "project100 at B = 65,536", **env100)
    outlier_gate(d_rows(dl100[idx100].cpu().numpy(), d64), d_rows(d32, d64), 20 * TOL, "d_last100 at B = 65,536", **envd100)
    # (6) steps = 0 is the identity
    q0, _ = net.project(q, steps=0)
    assert torch.equal(q0, q)


def test_weight_reload_and_errors(torch_cuda):
    torch = torch_cuda
    from posendf_amd import PoseNDF, amass_config, synth
    from posendf_amd.engine import PndfError
    net = make_net(torch, "lrelu", "mixed")
    q = torch.from_numpy(synth.make_poses(64, seed=2)).cuda()
    d_a = net(q, train=False)["dist_pred"].clone()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in golden_weights("live").items()})
    d_b = net(q, train=False)["dist_pred"]
    assert not torch.equal(d_a, d_b)                       # re-packed after load_state_dict
    assert (d_b > 0).all()
    with torch.no_grad():
        net.dfnet.lin6.bias.add_(1.0)                      # in-place parameter update is seen too
    d_c = net(q, train=False)["dist_pred"]
    assert torch.allclose(d_c, d_b + 1.0, atol=1e-5)
    cfg = amass_config("lrelu", "cuda:0")
    cfg["model"]["DFNet"]["dims"] = [256, 512, 2048, 512, 256, 64]     # wider than 1024: loud failure
    with pytest.raises(PndfError):
        PoseNDF(cfg)(q, train=False)
    cfg["model"]["DFNet"]["dims"] = [64] * 8                           # deeper than seven hidden layers: loud failure
    with pytest.raises(PndfError):
        PoseNDF(cfg)(q, train=False)
    cfg["model"]["DFNet"]["dims"] = [256, 512, 1024, 512, 256]         # another depth RUNS (runtime-planned kernels, tests/test_depth.py)
    other = PoseNDF(cfg)
    assert other(q, train=False)["dist_pred"].shape == (64, 1) and other._engine_for(q.device).kernel_name() == "pndf_generic_split_relu_kernel"      # (default precision f16x3)
    with pytest.raises(RuntimeError):                      # no double backward on the engine path
        qq = q.clone().requires_grad_(True)
        dd = net(qq, train=False)["dist_pred"]
        (g1,) = torch.autograd.grad(dd.sum(), qq, create_graph=True)
        g1.sum().backward()


def test_debug_stages(torch_cuda):
    """Stage-by-stage register dumps of workgroup 0 against the oracle's intermediates."""
    import subprocess
    import sys
    import os
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "gpu_selfcheck.py"), "lrelu"],
                       capture_output=True, text=True, timeout=900)
    print(r.stdout[-4000:])
    assert "MISMATCH" not in r.stdout and r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("act,precision", cases(["lrelu", "softplus"]))
def test_side_stream_and_graph_capture(torch_cuda, act, precision):
    """The C ABI enqueues on the caller's stream and allocates nothing per call (include/posendf_amd.h conventions):
    a launch on a side stream and a hipGraph capture + replay give the same bits as the default-stream call."""
    torch = torch_cuda
    from posendf_amd import synth
    net = make_net(torch, act, "live", precision=precision)      # softplus: the derivative scratch is allocated at create
    q = torch.from_numpy(synth.make_poses(300, seed=9)).cuda()
    ref_q, ref_d = net.project(q, steps=7)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        s_q, s_d = net.project(q, steps=7)
    side.synchronize()
    assert torch.equal(s_q, ref_q) and torch.equal(s_d, ref_d)
    # graph capture: static input/output buffers, replay after changing the input in place
    q_static = q.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        g_q, g_d = net.project(q_static, steps=7)
    q2 = torch.from_numpy(synth.make_poses(300, seed=10)).cuda()
    q_static.copy_(q2)
    g.replay()
    torch.cuda.synchronize()
    want_q, want_d = net.project(q2, steps=7)
    assert torch.equal(g_q, want_q) and torch.equal(g_d, want_d)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("act", ["lrelu", "softplus"])
def test_nonfinite_pose_stays_in_its_row(torch_cuda, precision, act):
    """Poses are independent rows (model/posendf.py:64 reshapes to [B,21,4]): a NaN pose poisons only its own
    distance and gradient, as in the reference, and its 63 workgroup neighbours are bit-identical.  Softplus too (round 4:
    the hardware min / max of its evaluation drop a NaN operand; the kernels carry it past them, pndf_device.h
    joint_axis_norms), for a NaN and for an infinity (F.normalize turns inf into inf / inf)."""
    torch = torch_cuda
    from posendf_amd import synth
    net = make_net(torch, act, "live", precision=precision)
    q = torch.from_numpy(synth.make_poses(128, seed=11)).cuda()
    clean_q, clean_d = net.project(q, steps=3)
    bad = q.clone()
    bad[37, 5, 2] = float("nan")
    got_q, got_d = net.project(bad, steps=3)
    torch.cuda.synchronize()
    keep = torch.ones(128, dtype=torch.bool, device="cuda")
    keep[37] = False
    assert torch.equal(got_q[keep], clean_q[keep]) and torch.equal(got_d[keep], clean_d[keep])
    assert not torch.isfinite(got_q[37]).all()
    for poison in (float("nan"), float("inf")):
        bad = q.clone()
        bad[90, 11, 0] = poison
        bad.requires_grad_(True)
        d = net(bad, train=False)["dist_pred"]
        (g,) = torch.autograd.grad(d.sum(), bad)
        assert torch.isnan(d[90]).all() and torch.isnan(g[90]).any(), (act, poison, d[90], g[90].isnan().sum())
        ok = torch.ones(128, dtype=torch.bool, device="cuda")
        ok[90] = False
        assert torch.isfinite(d[ok]).all() and torch.isfinite(g[ok]).all()


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("scale", [1e-9, 3e5])
def test_grad_outputs_scale_is_harmless(torch_cuda, precision, scale):
    """dist.backward(gradient=s): the input gradient is s times the unit one for any s -- also for scales that
    would leave the fp16 operand range of the split kernel if they seeded the backward pass
    (motion_denoise.py:31 weights the prior by 1e7 * c^2)."""
    torch = torch_cuda
    from posendf_amd import synth
    net = make_net(torch, "lrelu", "live", precision=precision)
    q = torch.from_numpy(synth.make_poses(200, seed=12)).cuda().requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    (unit,) = torch.autograd.grad(d, q, grad_outputs=torch.ones_like(d), retain_graph=True)
    (scaled,) = torch.autograd.grad(d, q, grad_outputs=torch.full_like(d, scale))
    assert torch.isfinite(scaled).all()
    assert rel_err((scaled / scale).cpu().numpy(), unit.cpu().numpy()) < 1e-6


def test_f16_single_is_a_bounded_approximation(torch_cuda):
    """precision="f16" (operands rounded to fp16, one MFMA per block) is a speed/accuracy comparison point, NOT a
    parity-grade mode: it must stay a sane approximation of the oracle (1e-2 relative here; the parity modes are
    held to 1e-4) and it must actually differ from f16x3 (i.e. the single-term kernel is the one that ran)."""
    torch = torch_cuda
    g = load_golden("lrelu", "live")
    net = make_net(torch, "lrelu", "live", precision="f16")
    q = torch.from_numpy(g["q"]).cuda().requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    (dq,) = torch.autograd.grad(d, q, grad_outputs=torch.ones_like(d))
    e_d = d_err(d.detach().cpu().numpy().ravel(), g["d_f64"].ravel())
    e_g = np.median(rel_err_rows(dq.cpu().numpy(), g["dq_f64"]))
    print(f"f16 single: d err {e_d:.2e}, median grad err {e_g:.2e}")
    assert e_d < 1e-2 and e_g < 1e-2
    ref = make_net(torch, "lrelu", "live", precision="f16x3")
    d3 = ref(q.detach(), train=False)["dist_pred"]
    assert not torch.equal(d3, d.detach())
    qp, dl = net.project(q.detach(), steps=100)
    assert torch.isfinite(qp).all() and (dl >= 0).all()


@pytest.mark.parametrize("precision", ["f16", "bf16"])
def test_one_term_kernels_compute_their_stated_arithmetic(torch_cuda, precision):
    """BASELINE.json configs[2] "fp32 vs bf16": precision="bf16" (v_mfma_f32_16x16x32_bf16, one MFMA per product block, operands rounded
    to bfloat16) and its fp16 sibling are measured comparison points, not parity modes.  What they ARE held to: (i) the kernel computes
    exactly the arithmetic it states -- every trunk operand rounded once to the 16-bit format, fp32 accumulate, everything else as in the
    parity kernels -- i.e. it agrees with a numpy emulation of that arithmetic (tests/fp8_cross_model.py) two orders of magnitude more
    closely than with the truth; (ii) the error against the fp64 oracle is the format's: bf16 (8 significant bits) worse than f16 (11)
    and both outside the 1e-4 bar that fp32 / f16x3 meet on the same inputs; (iii) the engine never selects them implicitly."""
    torch = torch_cuda
    import fp8_cross_model as model
    g = load_golden("lrelu", "live")
    sd = golden_weights("live")
    net = make_net(torch, "lrelu", "live", precision=precision)
    assert net._engine_for(torch.device("cuda:0")).kernel_name() == {"f16": "pndf_fused_half_relu_kernel", "bf16": "pndf_fused_bf16_relu_kernel"}[precision]
    q = torch.from_numpy(g["q"]).cuda().requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    (dq,) = torch.autograd.grad(d, q, grad_outputs=torch.ones_like(d))
    d, dq = d.detach().cpu().numpy().ravel(), dq.cpu().numpy()
    d_m, dq_m = model.forward_grad(g["q"], sd, "lrelu", precision)
    live = np.isfinite(g["dq_f64"]).all(axis=(1, 2))
    to_model = np.median(rel_err_rows(dq[live], dq_m[live])), np.median(np.abs(d[live] - d_m.ravel()[live]) / np.abs(g["d_f64"].ravel()[live]))
    to_truth = np.median(rel_err_rows(dq[live], g["dq_f64"][live])), np.median(np.abs(d[live] - g["d_f64"].ravel()[live]) / np.abs(g["d_f64"].ravel()[live]))
    print(f"{precision}: against its emulation dq {to_model[0]:.2e} d {to_model[1]:.2e}; against fp64 dq {to_truth[0]:.2e} d {to_truth[1]:.2e}")
    assert to_model[0] < to_truth[0] / 30 and to_model[0] < 3e-5, "the kernel does not compute the arithmetic it states"
    assert to_model[1] < 3e-5
    lo, hi = {"f16": (1e-4, 1e-2), "bf16": (1e-2, 0.5)}[precision]      # (the emulation: 8e-4 and 1e-1 -- bf16 flips LeakyReLU derivatives)
    assert lo < to_truth[0] < hi, f"{precision}: the error against fp64 is not the format's"
    # never implicit: the default engine of the same configuration runs the split-precision kernel
    from posendf_amd import PoseNDF, amass_config
    default = PoseNDF(amass_config("lrelu", "cuda:0"))
    default.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    default.eval()
    default(q.detach(), train=False)
    assert default._engine_for(torch.device("cuda:0")).kernel_name() == "pndf_fused_split_relu_kernel"
    qp, dl = net.project(q.detach(), steps=100)
    assert torch.isfinite(qp[torch.from_numpy(live).cuda()]).all()


def test_bf16_refusals(torch_cuda):
    """precision bf16 exists for amass.yaml-shaped relu-family networks: softplus and the runtime-planned kernels refuse it loudly."""
    from posendf_amd import engine
    with pytest.raises(engine.PndfError, match="relu / lrelu only"):
        engine.Engine("softplus", precision="bf16")
    with pytest.raises(engine.PndfError, match="amass.yaml-shaped"):
        engine.Engine("lrelu", precision="bf16", hidden=[96, 200, 40])


@pytest.mark.parametrize("precision", PRECISIONS)
def test_large_batch_indexing(torch_cuda, precision):
    """1,000,003 poses (15,626 workgroups, ragged tail): 64-bit indexing of poses, outputs and the softplus scratch;
    spot-checked against the oracle at the start, in the middle and at the very end."""
    torch = torch_cuda
    from oracle import posendf_np as onp
    from posendf_amd import synth
    B = 1_000_003
    sd = golden_weights("live")
    q = torch.from_numpy(synth.make_poses(B, seed=31)).cuda()
    for act in ("lrelu", "softplus"):
        net = make_net(torch, act, sd=sd, precision=precision)
        with torch.no_grad():
            d = net(q, train=False)["dist_pred"]
        qp, dl = net.project(q, steps=2)
        assert d.shape == (B, 1) and torch.isfinite(d).all() and torch.isfinite(qp).all()
        for lo in (0, B // 2 - 3, B - 70):
            sl = slice(lo, lo + 70 if lo + 70 <= B else B)
            qs = q[sl].cpu().numpy()
            d_o, _ = onp.forward_grad(qs, sd, act)
            assert d_err(d[sl, 0].cpu().numpy(), d_o) < TOL
            qp_o, _ = onp.project(qs, sd, steps=2, act=act)
            assert np.median(rel_err_rows(qp[sl].cpu().numpy(), qp_o)) < TOL / 10


@pytest.mark.parametrize("act,precision", [("lrelu", "f16x3"), ("softplus", "f16x3"), ("lrelu", "fp32"), ("lrelu", "generic")])
def test_maximum_batch_crosses_every_32_bit_boundary(torch_cuda, act, precision):
    """26,000,003 poses in ONE launch: 2.18e9 floats (past 2^31 elements) and 8.7 GB (past 2^32 bytes) per pose tensor, 406,251
    workgroups with a ragged tail -- sized for the 288 GB of one MI355X, where the reference's own loop (sample_poses.py:67-74) is
    bounded by HBM alone.  Poses are generated on the device (sample_poses.py:96-97's distribution).  Every window that straddles a
    boundary (2^31 bytes, 2^32 bytes, 2^31 elements) and both ends are (i) bit-identical to the same rows launched alone -- a
    pose's result depends on nothing but the pose -- and (ii) inside the 1e-4 bar against the oracle."""
    torch = torch_cuda
    from oracle import posendf_np as onp
    from posendf_amd import PoseNDF, amass_config, synth
    B = 26_000_003
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * 2**30:
        pytest.skip(f"needs 40 GB of free HBM, the device has {free / 2**30:.0f}")
    if precision == "generic":          # a runtime-planned network (net_modules.py:14-28: dims is a free list), split form
        hidden = [96, 200, 40]
        sd = synth.make_weights(5, 2.0, 0.1, dims=(126, *hidden, 1))
        cfg = amass_config(act, "cuda:0")
        cfg["model"]["DFNet"]["dims"] = hidden
        net = PoseNDF(cfg)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net.eval()
    else:
        sd = golden_weights("live")
        net = make_net(torch, act, sd=sd, precision=precision)
    gen = torch.Generator(device="cuda").manual_seed(77)
    q = torch.rand((B, 21, 4), device="cuda", generator=gen)
    q /= q.norm(dim=2, keepdim=True)
    with torch.no_grad():
        d = net(q, train=False)["dist_pred"]
    qp, dl = net.project(q, steps=2)
    assert d.shape == (B, 1) and qp.shape == (B, 21, 4) and dl.shape == (B, 1)
    row_bytes = 84 * 4
    starts = {0, B - 70, 2**31 // row_bytes - 35, 2**32 // row_bytes - 35, 2**31 // 84 - 35, 2**33 // row_bytes - 35}
    for lo in sorted(starts):
        sl = slice(lo, min(lo + 70, B))
        qs = q[sl].clone()
        with torch.no_grad():
            d_alone = net(qs, train=False)["dist_pred"]
        qp_alone, dl_alone = net.project(qs, steps=2)
        assert torch.equal(d[sl], d_alone), f"rows {lo}..: forward differs from the same rows launched alone"
        assert torch.equal(qp[sl], qp_alone) and torch.equal(dl[sl], dl_alone), f"rows {lo}..: projection differs"
        qn = qs.cpu().numpy()
        d_o, _ = onp.forward_grad(qn, sd, act)
        assert d_err(d[sl, 0].cpu().numpy(), d_o) < TOL
        qp_o, _ = onp.project(qn, sd, steps=2, act=act)
        assert np.median(rel_err_rows(qp[sl].cpu().numpy(), qp_o)) < TOL / 10
    # nothing in between was skipped or written twice: the same poses projected in four launches give the same bits everywhere
    assert bool(torch.isfinite(d).all())
    for i in range(0, B, 6_500_001):
        j = min(i + 6_500_001, B)
        qp_part, dl_part = net.project(q[i:j], steps=2)
        assert torch.equal(qp_part, qp[i:j]) and torch.equal(dl_part, dl[i:j]), f"rows {i}..{j}"
        del qp_part, dl_part


def test_precision_auto_selects_by_weight_range(torch_cuda):
    """Default precision 'auto': the split kernel when every trunk layer is inside its operating range, the exact
    fp32 kernel (with a warning) when not -- both are HIP kernels, there is no fallback off the engine."""
    import warnings
    torch = torch_cuda
    from posendf_amd import PoseNDF, amass_config, synth
    sd = golden_weights("live")
    q = torch.from_numpy(synth.make_poses(64, seed=3)).cuda()
    net = PoseNDF(amass_config("lrelu", "cuda:0"))                 # no engine key, no env override -> auto
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    d = net(q, train=False)["dist_pred"]
    assert net._engine_for(q.device).precision == "f16x3"
    tiny = {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    tiny["dfnet.lin3.weight"] *= 1e-5                              # any magnitude packs: the scale is per layer
    net1 = PoseNDF(amass_config("lrelu", "cuda:0"))
    net1.load_state_dict(tiny)
    d1 = net1(q, train=False)["dist_pred"]
    assert net1._engine_for(q.device).precision == "f16x3"
    from oracle import posendf_np as onp
    d_o1, _ = onp.forward_grad(q.cpu().numpy(), {k: v.numpy() for k, v in tiny.items()}, "lrelu", dtype=np.float64)
    assert d_err(d1.detach().cpu().numpy().ravel(), d_o1.ravel()) < TOL
    tiny["dfnet.lin3.weight"] *= 0.0                               # a layer that cannot be scaled at all
    net2 = PoseNDF(amass_config("lrelu", "cuda:0"))
    net2.load_state_dict(tiny)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        d2 = net2(q, train=False)["dist_pred"]
    assert net2._engine_for(q.device).precision == "fp32" and any("operating range" in str(x.message) for x in w)
    d_o, _ = onp.forward_grad(q.cpu().numpy(), {k: v.numpy() for k, v in tiny.items()}, "lrelu")
    assert d_err(d2.detach().cpu().numpy().ravel(), d_o.ravel()) < TOL
    assert torch.isfinite(d).all()


@pytest.mark.parametrize("act,precision", cases(["lrelu", "softplus"]))
def test_narrower_architecture_runs_zero_padded(torch_cuda, act, precision):
    """reference net_modules.py:14-28 builds DFNet from `dims`: a network of the amass.yaml depth with NARROWER hidden layers
    runs on the same kernels zero padded (the padded units have zero outgoing weights: they reach neither d nor its
    gradient -- also for softplus, whose padded units output ln 2 / beta)."""
    torch = torch_cuda
    from oracle import posendf_np as onp
    from posendf_amd import PoseNDF, amass_config, synth
    hidden = [192, 384, 700, 300, 200, 48]
    sd = synth.make_weights(5, 2.0, 0.1, dims=(126, *hidden, 1))
    cfg = amass_config(act, "cuda:0")
    cfg["model"]["DFNet"]["dims"] = hidden
    cfg["engine"] = {"precision": precision}
    net = PoseNDF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    qn = synth.make_poses(300, seed=41, signed=True)
    q = torch.from_numpy(qn).cuda().requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    (dq,) = torch.autograd.grad(d.sum(), q)
    sig_d, sig_g, d64, g64 = fp32_noise(qn, sd, act)
    pose_gate(d_rows(d.detach().cpu().numpy(), d64), sig_d, "d")
    pose_gate(rel_err_rows(dq.cpu().numpy(), g64), sig_g, "dq", exempt=kink_exempt(qn, sd, act))
    qp, _ = net.project(q.detach(), steps=5)
    q64, _ = onp.project(qn, sd, steps=5, act=act, dtype=np.float64)
    q32, _ = onp.project(qn, sd, steps=5, act=act)
    outlier_gate(rel_err_rows(qp.cpu().numpy(), q64), rel_err_rows(q32, q64), TOL, "project5", **traj_envelope(qn, sd, act, 5, q64))


@pytest.mark.parametrize("hidden", [[1, 1, 1, 1, 1, 1], [17, 33, 65, 31, 15, 1], [256, 512, 1024, 512, 256, 63],
                                    [255, 511, 1023, 511, 255, 64], [16, 32, 64, 32, 16, 16], [3, 500, 7, 300, 2, 40]],
                         ids=lambda h: "x".join(map(str, h)))
@pytest.mark.parametrize("act,precision", [("lrelu", "f16x3"), ("softplus", "f16x3"), ("relu", "fp32")])
def test_narrower_architecture_width_extremes(torch_cuda, act, precision, hidden):
    """`model.DFNet.dims` of the amass.yaml depth with arbitrary widths up to those of amass.yaml (net_modules.py:14-28):
    width 1, widths one short of a tile / of the full layer, wildly unbalanced layers -- against the fp64 oracle."""
    torch = torch_cuda
    from oracle import posendf_np as onp
    from posendf_amd import PoseNDF, amass_config, synth
    sd = synth.make_weights(6, 2.0, 0.1, dims=(126, *hidden, 1))
    cfg = amass_config(act, "cuda:0")
    cfg["model"]["DFNet"]["dims"] = hidden
    cfg["engine"] = {"precision": precision}
    net = PoseNDF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    qn = synth.make_poses(200, seed=43, signed=True)
    q = torch.from_numpy(qn).cuda().requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    (dq,) = torch.autograd.grad(d.sum(), q)
    sig_d, sig_g, d64, g64 = fp32_noise(qn, sd, act)
    pose_gate(d_rows(d.detach().cpu().numpy(), d64), sig_d, "d")
    # the two-unit softplus bottleneck (both units saturated low for every pose: derivative e^(beta z) ~ 1e-7, so every
    # rounding of z is amplified by beta = 100 and the whole gradient is ~1e-9): the eight-draw sensitivity estimate is
    # coarse there, the gate is held at 16 sigma instead of 8
    factor = 16.0 if (act == "softplus" and min(hidden) <= 3) else 8.0
    pose_gate(rel_err_rows(dq.cpu().numpy(), g64), sig_g, "dq", exempt=kink_exempt(qn, sd, act), factor=factor)
    qp, dl = net.project(q.detach(), steps=3)
    q64, _ = onp.project(qn, sd, steps=3, act=act, dtype=np.float64)
    q32, _ = onp.project(qn, sd, steps=3, act=act)
    outlier_gate(rel_err_rows(qp.cpu().numpy(), q64), rel_err_rows(q32, q64), TOL, "project3", **traj_envelope(qn, sd, act, 3, q64))


def _rescaled(sd, c):
    """layer l's outputs scaled by c[l] (W_l' = W_l c_l / c_{l-1}, b_l' = b_l c_l; c_6 = 1): for the positively homogeneous
    relu family the SAME function, with activations ten thousand times smaller or larger from layer to layer"""
    out = {k: v.copy() for k, v in sd.items()}
    prev = 1.0
    for l in range(7):
        cl = c[l] if l < 6 else 1.0
        out[f"dfnet.lin{l}.weight"] = (sd[f"dfnet.lin{l}.weight"].astype(np.float64) * (cl / prev)).astype(np.float32)
        out[f"dfnet.lin{l}.bias"] = (sd[f"dfnet.lin{l}.bias"].astype(np.float64) * cl).astype(np.float32)
        prev = cl
    return out


@pytest.mark.parametrize("scales", [(1e-4, 1e3, 1e-3, 1e4, 1e-2, 1e2), (1e4, 1e-3, 1e3, 1e-4, 1e2, 1e-2),
                                    (1e-6, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6), (1e5, 1e5, 1e5, 1e5, 1e5, 1e5)],
                         ids=["zigzag", "zagzig", "tiny", "huge"])
@pytest.mark.parametrize("act,precision", cases(["lrelu", "relu"]))
def test_layer_scale_extremes(torch_cuda, act, precision, scales):
    """activations of 1e-6 .. 1e+5 times the usual size, changing by up to 1e7 from one layer to the next: the per-layer
    weight scale and the per-pose operand scale of the split kernel must keep every operand inside fp16"""
    torch = torch_cuda
    from posendf_amd import synth
    sd = _rescaled(synth.make_weights(0, 2.0, 0.1), scales)
    net = make_net(torch, act, sd=sd, precision=precision)
    qn = synth.make_poses(256, seed=47, signed=True)
    q = torch.from_numpy(qn).cuda().requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    (dq,) = torch.autograd.grad(d.sum(), q)
    sig_d, sig_g, d64, g64 = fp32_noise(qn, sd, act)
    pose_gate(d_rows(d.detach().cpu().numpy(), d64), sig_d, "d")
    pose_gate(rel_err_rows(dq.cpu().numpy(), g64), sig_g, "dq", exempt=kink_exempt(qn, sd, act))


@pytest.mark.parametrize("beta", [1.0, 10.0, 1000.0])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_softplus_beta_range(torch_cuda, precision, beta):
    """model.DFNet.beta other than amass.yaml's 100 (net_modules.py:39-41): smooth (beta 1: every unit in the transition
    region, softplus(0) = 0.69) to almost-ReLU (beta 1000)"""
    torch = torch_cuda
    from oracle import posendf_np as onp
    from posendf_amd import PoseNDF, amass_config, synth
    sd = synth.make_weights(1, 2.0, 0.1)
    cfg = amass_config("softplus", "cuda:0")
    cfg["model"]["DFNet"]["beta"] = beta
    cfg["model"]["StrEnc"]["beta"] = beta
    cfg["engine"] = {"precision": precision}
    net = PoseNDF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    qn = synth.make_poses(256, seed=48, signed=True)
    q = torch.from_numpy(qn).cuda().requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    (dq,) = torch.autograd.grad(d.sum(), q)
    d64, g64 = onp.forward_grad(qn, sd, "softplus", beta=beta, dtype=np.float64)
    sig_d, sig_g = [], []
    rng = np.random.default_rng(1)
    for _ in range(8):              # conftest.fp32_noise with this beta
        qk = (qn * (1 + rng.uniform(-2.0 ** -23, 2.0 ** -23, qn.shape))).astype(np.float32)
        d32, g32 = onp.forward_grad(qk, sd, "softplus", beta=beta, dtype=np.float32)
        sig_d.append(d_rows(d32, d64))
        sig_g.append(rel_err_rows(g32, g64))
    pose_gate(d_rows(d.detach().cpu().numpy(), d64), np.max(sig_d, axis=0), "d")
    pose_gate(rel_err_rows(dq.cpu().numpy(), g64), np.max(sig_g, axis=0), "dq")


@pytest.mark.parametrize("go", [1e-12, 1e-7, 1e7, -3e9])
@pytest.mark.parametrize("act,precision", cases(["lrelu", "softplus"]))
def test_grad_outputs_of_any_magnitude(torch_cuda, act, precision, go):
    """motion_denoise.py weights the prior by 1e7 c^2 / (1 + it): the upstream gradient reaching forward()'s backward spans
    many orders of magnitude.  It multiplies the RESULT of the backward pass (never its fp16 operands), so dq(go) equals
    go * dq(1) to one rounding"""
    torch = torch_cuda
    from posendf_amd import synth
    net = make_net(torch, act, "live", precision=precision)
    qn = synth.make_poses(500, seed=50, signed=True)
    q = torch.from_numpy(qn).cuda().requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    (g1,) = torch.autograd.grad(d.sum(), q)
    q2 = torch.from_numpy(qn).cuda().requires_grad_(True)
    (g2,) = torch.autograd.grad(net(q2, train=False)["dist_pred"], q2, grad_outputs=torch.full_like(d, go))
    ref = g1.double() * go
    err = (g2.double() - ref).abs().amax(dim=(1, 2)) / ref.abs().amax(dim=(1, 2)).clamp_min(1e-300)
    assert torch.isfinite(g2).all() and float(err.max()) < 3e-7, float(err.max())
