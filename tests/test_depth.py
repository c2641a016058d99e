"""DFNets of other depths and widths than configs/amass.yaml.

reference model/network/net_modules.py:14-28 builds DFNet from a free list of hidden widths; tests/golden/make_golden_depth.py
runs the REAL reference on seven such networks (one to seven hidden layers, widths up to 1024, with and without the structure
encoder, all three activations) and records inputs and outputs.  Here:
  * CPU: the numpy oracle is pinned on those vectors (so it is a valid checker at these depths too), and the library's host twins
    (`train.device: cpu`) reproduce them;
  * GPU: `posendf_amd.PoseNDF` built from the same config runs on the runtime-planned HIP kernels (csrc/pndf_generic.hip) through
    the C ABI and is held to the same per-pose gates as the fused kernels; what the engine still refuses is asserted too.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, d_rows, escalated_noise, fp32_noise, outlier_gate, pose_gate, rel_err_rows, traj_envelope, traj_margin

CASES = ["d1_lrelu", "d4_lrelu", "d4_softplus", "d7_lrelu", "d7_softplus", "wide_relu", "noenc_d3_lrelu",
         "mix_softplus_lreluenc", "mix_relu_softplusenc"]      # the last two: model.StrEnc.act != model.DFNet.act ("trunk/encoder")
KERNELS = {("relu", "relu"): "pndf_generic_relu_kernel", ("softplus", "softplus"): "pndf_generic_softplus_kernel",
           ("softplus", "relu"): "pndf_generic_softplus_reluenc_kernel", ("relu", "softplus"): "pndf_generic_relu_spenc_kernel"}
TOL = 1e-4


def load_case(name):
    g = dict(np.load(os.path.join(GOLDEN, f"depth_{name}.npz")))
    hidden = [int(w) for w in g["hidden"]]
    act, enc = str(g["act"]), bool(g["encoder"])
    seed, gain, ob = (float(x) for x in g["weights"])
    from posendf_amd import synth
    sd = synth.make_weights(int(seed), gain, ob, dims=(126 if enc else 84, *hidden, 1))
    return g, hidden, act, enc, sd


def config_for(hidden, act, enc, device):
    from posendf_amd import amass_config
    trunk_act, _, enc_act = act.partition("/")
    cfg = amass_config(trunk_act, device)
    cfg["model"]["StrEnc"]["act"] = enc_act or trunk_act
    cfg["model"]["DFNet"]["dims"] = list(hidden)
    cfg["model"]["StrEnc"]["use"] = enc
    if not enc:
        cfg["model"]["DFNet"]["in_dim"] = 84
    return cfg


def kink_exempt(q, sd, act):
    from oracle import posendf_np as onp
    return None if act == "softplus" else onp.kink_margin(q, sd, act) < 1e-5


@pytest.mark.parametrize("name", CASES)
def test_oracle_is_pinned_at_other_depths(name):
    """fp64: the oracle IS the reference to rounding; fp32: within the reference arithmetic's own fp32 noise."""
    from oracle import posendf_np as onp
    g, hidden, act, enc, sd = load_case(name)
    d64, g64 = onp.forward_grad(g["q"], sd, act, dtype=np.float64)
    assert np.abs(d64 - g["d_f64"]).max() <= 1e-11 * max(np.abs(g["d_f64"]).max(), 1.0)
    assert rel_err_rows(g64, g["dq_f64"]).max() <= 1e-9
    q10, _ = onp.project(g["q"], sd, steps=10, act=act, dtype=np.float64)
    live = np.isfinite(g["q10_f64"]).all(axis=(1, 2))
    assert live.sum() >= len(live) - 1 and rel_err_rows(q10[live], g["q10_f64"][live]).max() <= 1e-7
    d32, g32 = onp.forward_grad(g["q"], sd, act)
    sig_d, sig_g, _, _ = fp32_noise(g["q"], sd, act)
    pose_gate(d_rows(g["d_f32"], g["d_f64"]), np.maximum(sig_d, d_rows(d32, g["d_f64"])), f"{name} reference fp32 d")
    pose_gate(rel_err_rows(g["dq_f32"], g["dq_f64"]), np.maximum(sig_g, rel_err_rows(g32, g["dq_f64"])), f"{name} reference fp32 dq",
              exempt=kink_exempt(g["q"], sd, act))


def _check_network(net, torch, g, act, sd, name, device):
    """single step (d, d d/d q), the autograd contract with an arbitrary grad_output, and the 1 / 10-step projections of a
    PoseNDF built from the case's config, against the reference's vectors through the gates of the fused kernels"""
    q = torch.from_numpy(g["q"]).to(device).requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    assert d.shape == (len(g["q"]), 1)
    (dq,) = torch.autograd.grad(d, q, grad_outputs=torch.ones_like(d))
    d_np, dq_np = d.detach().cpu().numpy(), dq.cpu().numpy()
    ref_d, ref_g = d_rows(g["d_f32"], g["d_f64"]), rel_err_rows(g["dq_f32"], g["dq_f64"])
    sig_d, sig_g, _, _ = fp32_noise(g["q"], sd, act, extra_d=[ref_d], extra_g=[ref_g])
    ex = kink_exempt(g["q"], sd, act)
    e_d, e_g = d_rows(d_np, g["d_f64"]), rel_err_rows(dq_np, g["dq_f64"])
    pose_gate(e_d, sig_d, f"{name} d")
    pose_gate(e_g, sig_g, f"{name} dq", exempt=ex)
    outlier_gate(e_g, ref_g, TOL, f"{name} dq", margin=traj_margin(g["q"], sd, act), sigma=sig_g)
    assert np.isfinite(d_np).all() and np.isfinite(dq_np).all()
    if act != "softplus":      # clipped poses: exactly zero distance and gradient where the reference's two runs agree on it
        z = (g["d_f32"][:, 0] == 0) & (g["d_f64"][:, 0] == 0)
        assert np.all(d_np[z, 0] == 0) and np.all(dq_np[z] == 0)
    # autograd contract (motion_denoise.py:82-83,97-98)
    q2 = torch.from_numpy(g["q"]).to(device).requires_grad_(True)
    (net(q2, train=False)["dist_pred"] * torch.from_numpy(g["grad_out"]).to(device)).sum().backward()
    truth = g["dq_f64"] * g["grad_out"].reshape(-1, 1, 1)
    ref_rows = rel_err_rows(g["grad_pose_f32"], truth)
    _, sig_go, _, _ = fp32_noise(g["q"], sd, act, extra_g=[ref_rows])
    outlier_gate(rel_err_rows(q2.grad.cpu().numpy(), truth), ref_rows, TOL, f"{name} grad_out", margin=traj_margin(g["q"], sd, act), sigma=sig_go)
    # projection loop (sample_poses.py:67-74): 1 and 10 steps, free running
    for steps in (1, 10):
        qp, dl = net.project(torch.from_numpy(g["q"]).to(device), steps=steps)
        truth_q = g[f"q{steps}_f64"]
        live = np.isfinite(truth_q).all(axis=(1, 2)) & np.isfinite(g[f"q{steps}_f32"]).all(axis=(1, 2))      # (the eps-clamp edge pose may overflow in both runs)
        env = traj_envelope(g["q"], sd, act, steps, truth_q)
        outlier_gate(rel_err_rows(qp.cpu().numpy()[live], truth_q[live]), rel_err_rows(g[f"q{steps}_f32"][live], truth_q[live]), TOL,
                     f"{name} project {steps}", margin=None if env["margin"] is None else env["margin"][live], sigma=env["sigma"][live])
        assert np.isfinite(dl.cpu().numpy()[live]).all()


@pytest.mark.parametrize("name", CASES)
def test_host_twins_at_other_depths(name):
    """`train.device: cpu`: the library's plain-C++ host twins (csrc/pndf_cpu.cpp) take the same configurations."""
    import torch
    from posendf_amd import PoseNDF
    g, hidden, act, enc, sd = load_case(name)
    net = PoseNDF(config_for(hidden, act, enc, "cpu"))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    _check_network(net, torch, g, act, sd, name + " host twin", "cpu")
    assert net._engine_for(torch.device("cpu")).kernel_name() == "pndf_cpu (host twin)"


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
@pytest.mark.parametrize("name", CASES)
def test_runtime_planned_kernels(name, precision):
    """The HIP path: every depth / width runs on pndf_generic_{relu,softplus}_kernel through the C ABI -- exact fp32 whatever
    `precision` the config asks for -- and meets the gates of the fused kernels."""
    import torch
    from posendf_amd import PoseNDF
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X (no CPU fallback exists)"
    g, hidden, act, enc, sd = load_case(name)
    cfg = config_for(hidden, act, enc, "cuda:0")
    cfg["engine"] = {"precision": precision}
    net = PoseNDF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    _check_network(net, torch, g, act, sd, f"{name} {precision}", "cuda:0")
    fam = lambda a: "softplus" if a == "softplus" else "relu"      # noqa: E731
    trunk_act, _, enc_act = act.partition("/")
    want = KERNELS[(fam(trunk_act), fam(enc_act or trunk_act) if enc else fam(trunk_act))]
    if precision != "fp32":      # the trunk on split-precision fp16 MFMAs (csrc/pndf_generic.hip gen_layer_split): its own four kernels
        want = want.replace("pndf_generic_", "pndf_generic_split_")
    assert net._engine_for(torch.device("cuda:0")).kernel_name() == want


@pytest.mark.gpu
def test_runtime_planned_kernels_full_size_properties():
    """B = 16,411 (ragged: 256 whole blocks + 27 poses, more blocks than compute units: the persistent grid walks them) on the
    four-hidden-layer network: bit-exact permutation equivariance and determinism, step composition (10 = 4 + 6 steps,
    bit-exact), in-place operation, a sample against the oracle's fp64 trajectory, steps = 0 identity."""
    import torch
    from oracle import posendf_np as onp
    from posendf_amd import PoseNDF, synth
    g, hidden, act, enc, sd = load_case("d4_lrelu")
    net = PoseNDF(config_for(hidden, act, enc, "cuda:0"))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    B = 16411
    q_np = synth.make_poses(B, seed=91)
    q = torch.from_numpy(q_np).cuda()
    a, da = net.project(q, steps=10)
    b, db = net.project(q, steps=10)
    assert torch.equal(a, b) and torch.equal(da, db)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
    c, dc = net.project(q[perm], steps=10)
    assert torch.equal(c, a[perm]) and torch.equal(dc, da[perm])
    m, _ = net.project(q, steps=4)
    e, de = net.project(m, steps=6)
    assert torch.equal(e, a) and torch.equal(de, da)
    z, dz = net.project(q, steps=0)
    assert torch.equal(z, q) and float(dz.abs().max()) == 0.0
    idx = np.random.default_rng(0).choice(B, 64, replace=False)
    q64, _ = onp.project(q_np[idx], sd, steps=10, act=act, dtype=np.float64)
    q32, _ = onp.project(q_np[idx], sd, steps=10, act=act)
    env = traj_envelope(q_np[idx], sd, act, 10, q64)
    outlier_gate(rel_err_rows(a[idx].cpu().numpy(), q64), rel_err_rows(q32, q64), TOL, "generic full size", margin=env["margin"], sigma=env["sigma"])
    # forward + gradient with grad_outputs at the same size, against the forward-only launch
    qg = q.clone().requires_grad_(True)
    d = net(qg, train=False)["dist_pred"]
    go = torch.from_numpy(np.random.default_rng(1).normal(size=(B, 1)).astype(np.float32)).cuda()
    (gq,) = torch.autograd.grad(d, qg, grad_outputs=go)
    (g1,) = torch.autograd.grad(net(qg, train=False)["dist_pred"].sum(), qg)
    assert torch.allclose(gq, go.view(-1, 1, 1) * g1, rtol=2e-6, atol=0)
    with torch.no_grad():
        assert torch.equal(net(q, train=False)["dist_pred"], d.detach())


def test_what_the_engine_still_refuses():
    """pndf_create accepts n_dims 3 .. 9 with hidden widths 1 .. 1024; beyond that it fails loudly (no silent truncation)."""
    import ctypes
    from posendf_amd import engine
    lib = engine.load_library()
    h = ctypes.c_void_p()
    cfg = engine.PndfConfig()
    for n_dims, dims, why in ((2, [126, 1], b"depth"), (10, [126] + [64] * 8 + [1], b"depth"), (5, [126, 64, 1025, 64, 1], b"widths"),
                              (5, [126, 64, 0, 64, 1], b"widths"), (5, [100, 64, 64, 64, 1], b"in_dim"), (5, [126, 64, 64, 64, 2], b"in_dim")):
        lib.pndf_default_config(ctypes.byref(cfg), 1, 100.0)
        cfg.n_dims = n_dims
        for i in range(16):
            cfg.dims[i] = dims[i] if i < len(dims) else 0
        assert lib.pndf_create(ctypes.byref(h), ctypes.byref(cfg), 0) == -4, (n_dims, dims)
        assert why in lib.pndf_last_error(None), (why, lib.pndf_last_error(None))
    for hidden in ([], [8] * 8):
        with pytest.raises(engine.PndfError, match="hidden layers"):
            engine.CpuEngine("lrelu", hidden=hidden)
    # accepted shapes get as far as the device (no HIP device here -> -6, on a GPU box 0)
    lib.pndf_default_config(ctypes.byref(cfg), 2, 100.0)
    cfg.n_dims = 4
    for i, w in enumerate([84, 1024, 1, 1]):
        cfg.dims[i] = w
    rc = lib.pndf_create(ctypes.byref(h), ctypes.byref(cfg), 0)
    assert rc in (0, -6), (rc, lib.pndf_last_error(None))
    if h.value:
        lib.pndf_destroy(h)


# width / depth extremes that no fixture of the reference covers: the oracle -- pinned at 1, 3, 4, 6 and 7 hidden layers above -- is the checker
EXTREMES = [([1], "lrelu", True), ([16], "relu", True), ([17], "softplus", True), ([1024], "lrelu", True), ([1, 1], "lrelu", True),
            ([15, 17, 33], "softplus", True), ([1024, 1024], "relu", True), ([64] * 7, "lrelu", True), ([1000, 3, 900], "lrelu", True),
            ([7, 1024, 5, 1024, 3, 1024, 9], "softplus", False), ([512, 513], "lrelu", False), ([256, 512, 1024, 512, 256, 64, 32], "relu", True),
            # three groups per pass (the k steps of such a pass alternate between the halves of a ring slot), passes of 4 + 1 and 4 + 3
            # groups, an even and an odd number of groups in the whole stream (a slot of padding or half a slot behind the trunk)
            ([384], "lrelu", True), ([384, 384], "softplus", True), ([640, 896], "relu", True), ([130, 384, 48], "lrelu", False),
            ([128], "relu", False), ([128, 128], "lrelu", True),
            # found by tools/sweep_generic.py: behind a ONE-unit Softplus layer the gradient of a pose is e^(beta z) ~ 1e-20 -- the split
            # form's gradient scale must reach that far (gen_pose_scale_grad)
            ([65, 347, 385, 256, 81, 1, 2], "softplus", True)]


def live_weights(dims, act):
    """the first seed (deterministic) whose network is not clipped to d = 0 on most poses: a dead network tests nothing"""
    from oracle import posendf_np as onp
    from posendf_amd import synth
    probe = synth.make_poses(48, seed=60)
    for seed in range(50, 90):
        sd = synth.make_weights(seed, 2.5, 0.3, dims=dims)
        d = onp.forward(probe, sd, act)
        if (d > 1e-3).mean() > 0.8:
            return sd
    raise AssertionError(f"no live weight set for {dims} {act}")


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])      # the exact and the split-precision form of the runtime-planned kernels
@pytest.mark.parametrize("hidden,act,enc", EXTREMES, ids=lambda v: "-".join(map(str, v)) if isinstance(v, list) else str(v))
def test_runtime_planned_kernels_width_and_depth_extremes(hidden, act, enc, precision):
    """Hidden widths 1, 15 - 17, one past a tile / a pass / 512, 1024; one to seven hidden layers; bottlenecks of a few units between
    1024-wide layers: single step and a 3-step projection against the numpy oracle (fp64 truth, the fp32 oracle's own noise as the
    envelope), through the per-pose gates."""
    import torch
    from oracle import posendf_np as onp
    from posendf_amd import PoseNDF, synth
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X (no CPU fallback exists)"
    dims = (126 if enc else 84, *hidden, 1)
    sd = live_weights(dims, act)
    cfg = config_for(hidden, act, enc, "cuda:0")
    cfg["engine"] = {"precision": precision}
    net = PoseNDF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    q_np = np.concatenate([synth.make_poses(100, seed=61), synth.make_poses(100, seed=62, signed=True)])
    q = torch.from_numpy(q_np).cuda().requires_grad_(True)
    d = net(q, train=False)["dist_pred"]
    (dq,) = torch.autograd.grad(d.sum(), q)
    assert net._engine_for(q.device).kernel_name().startswith("pndf_generic_split_" if precision == "f16x3" else "pndf_generic_")
    assert ("_split_" in net._engine_for(q.device).kernel_name()) == (precision == "f16x3")
    sig_d, sig_g, d64, g64 = fp32_noise(q_np, sd, act)
    what = f"{hidden} {act} enc={enc} {precision}"
    # (`escalate`, as in the held-out sweep of tests/test_gpu_sweep.py: a pose over the cheap envelope is held to twice the reference
    # arithmetic's own variability there, measured properly -- the 3-unit bottlenecks of these networks make single poses ill-conditioned:
    # pose 20 of [7, 1024, 5, 1024, 3, 1024, 9]: cheap sigma 5e-6, 32 perturbed fp32 evaluations of the oracle 4.7e-5)
    pose_gate(d_rows(d.detach().cpu().numpy(), d64), sig_d, what + " d", escalate=lambda i: escalated_noise(q_np, sd, act, i, d64, kind="d"))
    pose_gate(rel_err_rows(dq.cpu().numpy(), g64), sig_g, what + " dq", exempt=kink_exempt(q_np, sd, act),
              escalate=lambda i: escalated_noise(q_np, sd, act, i, g64, kind="g"))
    qp, _ = net.project(torch.from_numpy(q_np).cuda(), steps=3)
    q64, _ = onp.project(q_np, sd, steps=3, act=act, dtype=np.float64)
    q32, _ = onp.project(q_np, sd, steps=3, act=act)
    env = traj_envelope(q_np, sd, act, 3, q64)
    outlier_gate(rel_err_rows(qp.cpu().numpy(), q64), rel_err_rows(q32, q64), TOL, what + " project 3", margin=env["margin"], sigma=env["sigma"])


@pytest.mark.gpu
def test_runtime_planned_reload_switches_between_the_two_forms():
    """precision f16x3 on a runtime-planned network: the split-precision form -- unless a layer has no finite non-zero weight (it cannot
    be scaled into the fp16 range), in which case the SAME handle runs the exact fp32 form after that load, and the split form again
    after the next one (csrc/pndf_generic.hip pndf_generic_load re-plans the stream per load)."""
    import torch
    from oracle import posendf_np as onp
    from posendf_amd import PoseNDF, synth
    hidden, act = [96, 200, 40], "lrelu"
    dims = (126, *hidden, 1)
    sd = live_weights(dims, act)
    net = PoseNDF(config_for(hidden, act, True, "cuda:0"))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    q_np = synth.make_poses(96, seed=63)
    q = torch.from_numpy(q_np).cuda()
    eng = net._engine_for(q.device)
    p1, d1 = net.project(q, steps=3)
    assert eng.kernel_name() == "pndf_generic_split_relu_kernel"
    dead = {k: v.copy() for k, v in sd.items()}
    dead["dfnet.lin1.weight"][:] = 0.0                       # a layer of zeros: d = act(b) downstream, finite, but unscalable
    net.load_state_dict({k: torch.from_numpy(v) for k, v in dead.items()})
    p2, d2 = net.project(q, steps=3)
    eng = net._engine_for(q.device)
    assert eng.kernel_name() == "pndf_generic_relu_kernel"
    q64, d64 = onp.project(q_np, dead, steps=3, act=act, dtype=np.float64)
    assert np.isfinite(p2.cpu().numpy()).all()
    assert np.abs(p2.cpu().numpy().reshape(q64.shape) - q64).max() < 1e-5 and np.abs(d2.cpu().numpy().reshape(-1) - d64.reshape(-1)).max() < 1e-5
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    p3, d3 = net.project(q, steps=3)
    assert net._engine_for(q.device).kernel_name() == "pndf_generic_split_relu_kernel"
    assert torch.equal(p1, p3) and torch.equal(d1, d3)


@pytest.mark.gpu
def test_softplus_bottleneck_network_takes_the_runtime_planned_split_kernels():
    """Six hidden widths within amass.yaml's normally run zero-padded on the fused kernels.  A Softplus network with a layer of a few
    units is the exception under split precision: behind such a layer a pose's whole gradient is e^(beta z) ~ 1e-20, below what the fused
    split kernels' per-pose scale (2^-40 .. 2^40) can lift into fp16's range -- 49 of these 200 poses came back beyond 1 % from
    pndf_fused_split_softplus_kernel (tools/r6/fused_narrow.py); the runtime-planned split kernels scale gradients down to 2^-80."""
    import torch
    from oracle import posendf_np as onp
    from posendf_amd import PoseNDF, synth
    hidden, act = [256, 2, 1024, 512, 2, 64], "softplus"
    sd = live_weights((126, *hidden, 1), act)
    q_np = np.concatenate([synth.make_poses(100, seed=61), synth.make_poses(100, seed=62, signed=True)])
    sig_d, sig_g, d64, g64 = fp32_noise(q_np, sd, act)
    for precision, kernel in (("f16x3", "pndf_generic_split_softplus_kernel"), ("fp32", "pndf_fused_softplus_kernel")):
        cfg = config_for(hidden, act, True, "cuda:0")
        cfg["engine"] = {"precision": precision}
        net = PoseNDF(cfg)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net.eval()
        q = torch.from_numpy(q_np).cuda().requires_grad_(True)
        d = net(q, train=False)["dist_pred"]
        (dq,) = torch.autograd.grad(d.sum(), q)
        assert net._engine_for(q.device).kernel_name() == kernel
        what = f"{hidden} {act} {precision}"
        pose_gate(d_rows(d.detach().cpu().numpy(), d64), sig_d, what + " d", escalate=lambda i: escalated_noise(q_np, sd, act, i, d64, kind="d"))
        pose_gate(rel_err_rows(dq.cpu().numpy(), g64), sig_g, what + " dq", escalate=lambda i: escalated_noise(q_np, sd, act, i, g64, kind="g"))
