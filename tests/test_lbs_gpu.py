"""GPU tests of the HIP linear-blend-skinning path (posendf_amd/csrc/pndf_lbs.hip, C ABI pndf_lbs_*) against
the numpy oracle oracle/lbs_np.py -- parity UNPINNED (smplx and the SMPL files are third-party and absent): the oracle
restates the published algorithm, model parameters are synthetic with SMPL's shapes (6,890 vertices, 24 joints, 21 picked
joints).  Both arithmetics of the forward / fused-terms passes run every test: "f16x3" (the default: fp16 MFMAs on operands
split into hi + lo halves, fp32 accumulate) and "fp32" (fp32 MFMAs).  Tolerances are the same for both: errors are measured
against the fp64 oracle and held to 1e-4 relative (north_star's bar) and to a multiple of the fp32 oracle's own error."""
import numpy as np
import pytest
import torch

from conftest import golden_weights
from oracle import denoise_np, lbs_np

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module", params=["f16x3", "fp32"])
def smpl_like(request):
    """both arithmetics of the forward / fused-terms passes (split fp16 MFMAs, the default, and fp32 MFMAs): same tolerances"""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    from posendf_amd import BodyModel
    m = lbs_np.synthetic_model(seed=11)                       # V = 6890, SMPL tree, 21 vertex-picked joints
    bm = BodyModel(m, device="cuda:0", extra_joint_vertex=m["extra_joint_vertex"], precision=request.param)
    assert bm.precision == request.param and bm.lib.pndf_lbs_precision(bm.handle) == BodyModel.PRECISIONS[request.param]
    return m, bm


def _theta(S, T, seed=0):
    rng = np.random.default_rng(seed)
    th = np.cumsum(rng.normal(size=(S, T, 69)) * 0.03, axis=1) + rng.normal(size=(S, 1, 69)) * 0.3
    th[0, min(2, T - 1), 6:9] = 0.0                           # a zero rotation
    return th.astype(np.float32)


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("N", [1, 16, 37])
def test_forward_vertices_and_joints(smpl_like, N):
    m, bm = smpl_like
    th = _theta(1, N, seed=N)[0]
    out = bm(pose_body=torch.from_numpy(th))
    assert out.vertices.shape == (N, 6890, 3) and out.Jtr.shape == (N, 45, 3)
    V64, J64 = lbs_np.lbs(th, m)
    V32, J32 = lbs_np.lbs(th, m, np.float32)
    ev, ej = _rel(out.vertices.cpu().numpy(), V64), _rel(out.Jtr.cpu().numpy(), J64)
    print(f"LBS forward N={N}: verts {ev:.2e} (fp32 oracle {_rel(V32, V64):.2e})  joints {ej:.2e} (fp32 oracle {_rel(J32, J64):.2e})")
    assert ev < 1e-5 and ej < 1e-5
    # joints without the vertex output: the chain + the 21 picked vertices alone on the VALU (pndf_lbs_joints_only_*),
    # another arithmetic than the vertex kernels' MFMAs -- equal to rounding, and held to the fp64 oracle by itself
    jo = bm.joints_of(torch.from_numpy(th)).cpu().numpy()
    assert jo.shape == (N, 45, 3) and _rel(jo, J64) < 1e-5 and _rel(jo, out.Jtr.cpu().numpy().astype(np.float64)) < 2e-6


@pytest.mark.parametrize("S,T,it", [(1, 1, 2), (2, 16, 0), (3, 33, 0), (3, 33, 2), (1, 31, 3), (2, 46, 1)])
def test_fused_terms_gradient(smpl_like, S, T, it):
    """d (10 (1+it) temp + [it>0] 100/(1+it) data) / d theta from ONE fused pass (vertices never in HBM), chunks of 15 pairs
    with shared frames, sequences independent."""
    m, bm = smpl_like
    th = _theta(S, T, seed=5 + T)
    th0 = th + np.random.default_rng(1).normal(size=th.shape).astype(np.float32) * 0.05
    j0 = bm.joints_of(torch.from_numpy(th0))
    g = bm.terms_grad(torch.from_numpy(th).cuda(), j0, it).cpu().numpy()
    for s in range(S):
        _, J0 = lbs_np.lbs(th0[s], m)
        g64, _ = lbs_np.body_terms(th[s], J0, m, it)
        g32, _ = lbs_np.body_terms(th[s], J0.astype(np.float32), m, it, np.float32)
        scale = max(np.abs(g64).max(), 1e-30)
        err, ref = np.abs(g[s] - g64).max() / scale, np.abs(g32 - g64).max() / scale
        print(f"LBS terms grad S={S} T={T} it={it} seq {s}: err {err:.2e} (fp32 oracle {ref:.2e}) |g| {scale:.2e}")
        if T == 1 and it == 0:
            assert np.all(g[s] == 0)
        else:
            assert np.isfinite(g[s]).all() and err < max(TOL, 8 * ref)


def test_general_reverse_pass_through_autograd(smpl_like):
    """BodyModel.forward is differentiable w.r.t. pose_body (the reference backpropagates through smplx the same way)."""
    m, bm = smpl_like
    th = _theta(1, 21, seed=9)[0]
    rng = np.random.default_rng(2)
    gv = rng.normal(size=(21, 6890, 3)).astype(np.float32)
    gj = rng.normal(size=(21, 45, 3)).astype(np.float32)
    t = torch.from_numpy(th).cuda().requires_grad_(True)
    out = bm(pose_body=t)
    ((out.vertices * torch.from_numpy(gv).cuda()).sum() + (out.Jtr * torch.from_numpy(gj).cuda()).sum()).backward()
    _, _, cache = lbs_np.lbs(th, m, keep=True)
    g64 = lbs_np.lbs_vjp(m, cache, gv, gj)
    assert _rel(t.grad.cpu().numpy(), g64) < 1e-5
    # vertices only / joints only
    t2 = torch.from_numpy(th).cuda().requires_grad_(True)
    (bm(pose_body=t2).Jtr * torch.from_numpy(gj).cuda()).sum().backward()
    assert _rel(t2.grad.cpu().numpy(), lbs_np.lbs_vjp(m, cache, np.zeros_like(gv), gj)) < 1e-5


def test_large_batch_matches_small_batch(smpl_like):
    """128 x 300 frames in one call (no vertex split) against the same sequences alone (vertex range split over 8
    workgroups): only the summation order differs."""
    m, bm = smpl_like
    S, T = 128, 300
    th = torch.from_numpy(_theta(S, T, seed=3)).cuda()
    j0 = bm.joints_of(th + 0.03)
    g = bm.terms_grad(th, j0, 2)
    assert torch.isfinite(g).all()
    for s in (0, 77, 127):
        gs = bm.terms_grad(th[s:s + 1].contiguous(), j0.reshape(S, T, 45, 3)[s].contiguous(), 2)
        scale = gs.abs().max().item()
        assert (g[s] - gs[0]).abs().max().item() < 1e-5 * scale
    g2 = bm.terms_grad(th, j0, 2)
    assert torch.equal(g, g2)                                   # deterministic


@pytest.mark.parametrize("precision,schedule", [("f16x3", "motion_denoise"), ("fp32", "motion_denoise"),
                                                ("f16x3", "partial_observation")])
def test_fused_denoise_with_body_model_matches_oracle_loop(smpl_like, precision, schedule):
    """optimize(fused=True) with the reference's objective (pose prior + SMPL vertex temporal term + joint data term,
    motion_denoise.py:74-99): engine launch + fused LBS pass + Adam kernel per step, against the numpy oracle of the loop."""
    from posendf_amd import PoseNDF, amass_config
    from posendf_amd.motion_denoise import MotionDenoise
    m, bm = smpl_like
    sd = golden_weights("live")
    cfg = amass_config("lrelu", "cuda:0")
    cfg["engine"] = {"precision": precision}
    net = PoseNDF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    S, T, iters, per = 2, 20, 2, 3
    th0 = _theta(S, T, seed=21) * 0.5
    md = MotionDenoise(net, body_model=bm, device="cuda:0", schedule=schedule)      # partial_observation.py:29-35: other weights
    got, _ = md.denoise(torch.from_numpy(th0), iterations=iters, steps_per_iter=per, fused=True)
    ref = denoise_np.optimize(th0, sd, iterations=iters, steps_per_iter=per, body_model=m, schedule=schedule)
    diff = np.abs(got.cpu().numpy() - ref)
    moved = np.abs(ref - th0).max()
    print(f"fused denoise + LBS vs oracle loop: median {np.median(diff):.2e} p99 {np.percentile(diff, 99):.2e} max {diff.max():.2e} moved {moved:.3f}")
    assert np.isfinite(got.cpu().numpy()).all()
    assert np.median(diff) < 1e-5 and (diff > 1e-3).mean() < 0.01 and diff.max() < 0.5 * moved
    assert np.abs(got.cpu().numpy()[..., 63:] - th0[..., 63:]).max() > 1e-3      # the hand joints are optimised too
    # the autograd driver around the same engine and body model takes the same steps
    auto, hist = md.denoise(torch.from_numpy(th0), iterations=iters, steps_per_iter=per)
    d2 = (auto - got).abs().flatten()
    assert d2.median().item() < 1e-5 and (d2 > 1e-3).float().mean().item() < 0.01
    assert {"pose_pr", "temp"} <= set(hist[0]) and "data" in hist[-1]


@pytest.mark.parametrize("lbs_precision", ["f16x3", "fp32"])
def test_small_model_short_sequences_and_reference_nan_semantics(lbs_precision):
    """A 41-vertex model (three vertex groups, the last one padded), sequences of 2, 15, 16 and 17 frames (one pair; one
    chunk exactly; a chunk boundary with and without a shared frame), and the reference's behaviour for two IDENTICAL
    consecutive frames: no epsilon under the root (motion_denoise.py:89), so the temporal gradient of those frames is NaN."""
    from posendf_amd import BodyModel
    m = lbs_np.synthetic_model(V=41, seed=5, extra=(3, 17, 40))
    bm = BodyModel(m, device="cuda:0", precision=lbs_precision)
    assert bm.num_joints == 27 and bm.num_vertices == 41
    for T in (2, 15, 16, 17):
        th = _theta(2, T, seed=T)
        th0 = th + 0.04
        j0 = bm.joints_of(torch.from_numpy(th0))
        g = bm.terms_grad(torch.from_numpy(th).cuda(), j0, 1).cpu().numpy()
        for s in range(2):
            _, J0 = lbs_np.lbs(th0[s], m)
            g64, _ = lbs_np.body_terms(th[s], J0, m, 1)
            assert np.abs(g[s] - g64).max() < TOL * np.abs(g64).max(), (T, s)
    th = _theta(1, 6, seed=1)
    th[0, 3] = th[0, 2]                                      # frames 2 and 3 identical: |V2 - V3| = 0
    g = bm.terms_grad(torch.from_numpy(th).cuda(), None, 0).cpu().numpy()[0]
    with np.errstate(invalid="ignore", divide="ignore"):
        g64, _ = lbs_np.body_terms(th[0], None, m, 0)
    assert np.isnan(g64[2]).all() and np.isnan(g64[3]).all() and np.isfinite(g64[[0, 1, 4, 5]]).all()      # the reference's NaN
    assert np.isnan(g[2]).all() and np.isnan(g[3]).all() and np.isfinite(g[[0, 1, 4, 5]]).all()
    assert np.abs(g[[0, 1, 4, 5]] - g64[[0, 1, 4, 5]]).max() < TOL * np.abs(g64[[0, 1, 4, 5]]).max()


def test_sample_pose_project_with_body_model(smpl_like):
    """experiments/sample_poses.py:57-83 mirrored: project random poses (one persistent launch) and mesh them before / after."""
    from posendf_amd import PoseNDF, amass_config
    from posendf_amd.sample_poses import SamplePose, quaternion_to_axis_angle, random_poses
    m, bm = smpl_like
    sd = golden_weights("live")
    net = PoseNDF(amass_config("lrelu", "cuda:0"))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    q0 = random_poses(10, "cuda:0", torch.Generator().manual_seed(3))
    poses, dist, meshes = SamplePose(net, body_model=bm).project(q0, steps=10)
    want, dwant = net.project(q0, steps=10)
    assert torch.equal(poses, want) and torch.equal(dist, dwant)
    assert meshes["vertices"].shape == (10, 6890, 3) and meshes["vertices_init"].shape == (10, 6890, 3)
    aa = torch.zeros(10, 23, 3)
    aa[:, :21] = quaternion_to_axis_angle(poses.cpu())
    V64, _ = lbs_np.lbs(aa.reshape(10, 69).numpy(), m)
    assert _rel(meshes["vertices"].cpu().numpy(), V64) < 1e-5
    assert dist.mean() < net(q0, train=False)["dist_pred"].mean()


def test_denoise_motion_file_script_level(smpl_like, tmp_path):
    """experiments/motion_denoise.py:124-153 mirrored: noisy motion .npz in, denoised poses and the v2v error (cm) out."""
    from posendf_amd import PoseNDF, amass_config
    from posendf_amd.motion_denoise import denoise_motion_file, load_motion_npz, v2v_error_cm
    m, bm = smpl_like
    net = PoseNDF(amass_config("lrelu", "cuda:0"))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in golden_weights("live").items()})
    gt = _theta(1, 24, seed=31)[0] * 0.5
    gt[:, 63:] = 0
    noisy = gt.copy()
    noisy[:, :63] += np.random.default_rng(5).normal(size=(24, 63)).astype(np.float32) * 0.05
    np.savez(tmp_path / "noisy.npz", pose_body=noisy[:, :63])
    np.savez(tmp_path / "gt.npz", pose_body=gt[:, :63])
    out, err = denoise_motion_file(net, bm, tmp_path / "noisy.npz", gt_file=tmp_path / "gt.npz", iterations=2, steps_per_iter=3)
    assert out.shape == (24, 69) and torch.isfinite(out).all() and np.isfinite(err) and err > 0
    V64, _ = lbs_np.lbs(out.cpu().numpy(), m)
    G64, _ = lbs_np.lbs(gt, m)
    want = np.sqrt(((V64 - G64) ** 2).sum(-1)).mean() * 100.0
    assert abs(err - want) < 1e-3 * want
    assert abs(v2v_error_cm(bm, load_motion_npz(tmp_path / "gt.npz"), load_motion_npz(tmp_path / "gt.npz"))) < 1e-6


@pytest.mark.parametrize("lbs_precision", ["f16x3", "fp32"])
def test_other_kinematic_tree_takes_the_generic_per_frame_kernels(lbs_precision):
    """SMPL's own tree runs on per-frame kernels that hold it as a compile-time table (everything in registers); any other valid
    tree (parents before children) on the generic ones.  A chain with two side branches, against the oracle."""
    from posendf_amd import BodyModel
    parents = (-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 3, 20, 7, 22)
    m = lbs_np.synthetic_model(V=137, seed=9, parents=parents, extra=(5, 60, 136))
    bm = BodyModel(m, device="cuda:0", precision=lbs_precision)
    th = _theta(2, 19, seed=4) * 0.4
    out = bm(pose_body=torch.from_numpy(th.reshape(-1, 69)))
    V64, J64 = lbs_np.lbs(th.reshape(-1, 69), m)
    assert _rel(out.vertices.cpu().numpy(), V64) < 1e-5 and _rel(out.Jtr.cpu().numpy(), J64) < 1e-5
    th0 = th + 0.03
    j0 = bm.joints_of(torch.from_numpy(th0))
    g = bm.terms_grad(torch.from_numpy(th).cuda(), j0, 2).cpu().numpy()
    for s_ in range(2):
        _, J0 = lbs_np.lbs(th0[s_], m)
        g64, _ = lbs_np.body_terms(th[s_], J0, m, 2)
        assert np.abs(g[s_] - g64).max() < TOL * np.abs(g64).max()


def test_constructor_and_argument_hardening(tmp_path):
    """what the constructor and the raw-pointer entry points must not accept silently."""
    from posendf_amd import BodyModel
    from posendf_amd.engine import PndfError
    m = lbs_np.synthetic_model(V=137, seed=9, extra=(5, 60, 136))
    parents_before = np.array(m["parents"], copy=True)
    m_i32 = dict(m, parents=np.ascontiguousarray(m["parents"], dtype=np.int32))
    with pytest.raises(PndfError):                                      # validated before anything is created
        BodyModel(m, device="cuda:0", precision="bf16")
    with pytest.raises(PndfError):                                      # a vertex named twice
        BodyModel(m, device="cuda:0", extra_joint_vertex=(5, 5, 136))
    bm = BodyModel(m_i32, device="cuda:0")
    assert np.array_equal(m_i32["parents"], parents_before)             # the caller's table keeps its root entry
    th = _theta(2, 9, seed=3) * 0.4
    # betas shorter than num_betas are zero padded (the host read used to run past the array); the model's own betas are the default
    betas = np.array([0.7, -0.4, 0.2], np.float32)
    b_short = BodyModel(m, device="cuda:0", betas=betas)
    full = np.zeros(10, np.float32)
    full[:3] = betas
    assert np.array_equal(b_short.betas.cpu().numpy(), full)
    V64, J64 = lbs_np.lbs(th.reshape(-1, 69), dict(m, betas=full))
    out = b_short(pose_body=torch.from_numpy(th.reshape(-1, 69)))
    assert _rel(out.vertices.cpu().numpy(), V64) < 1e-5 and _rel(out.Jtr.cpu().numpy(), J64) < 1e-5
    b_own = BodyModel(dict(m, betas=full), device="cuda:0")              # params['betas'] is what rest_shape() of the oracle uses
    assert torch.equal(b_own(pose_body=torch.from_numpy(th.reshape(-1, 69))).vertices, out.vertices)
    # from_arrays / from_npz(faces=...)
    b_arr = BodyModel.from_arrays(m["v_template"], m["shapedirs"], m["posedirs"], m["J_regressor"], m["parents"], m["lbs_weights"],
                                  extra_joint_vertex=(5, 60, 136), device="cuda:0")
    assert torch.equal(b_arr.joints_of(torch.from_numpy(th)), bm.joints_of(torch.from_numpy(th)))
    faces = np.arange(12, dtype=np.int64).reshape(4, 3)
    np.savez(tmp_path / "model.npz", **{k: v for k, v in m.items()}, f=faces + 1)
    assert np.array_equal(BodyModel.from_npz(tmp_path / "model.npz", device="cuda:0").faces_tensor.cpu().numpy(), faces + 1)
    assert np.array_equal(BodyModel.from_npz(tmp_path / "model.npz", device="cuda:0", faces=faces).faces_tensor.cpu().numpy(), faces)
    # forward hands back the caller's own [N,69] tensor (smplx does: the reference feeds body_pose into the next step)
    pose = torch.from_numpy(th.reshape(-1, 69)).cuda().requires_grad_(True)
    assert bm(pose_body=pose).body_pose is pose
    zeros = torch.zeros(1, 10, device="cuda")
    for _ in range(3):                                                  # the reference passes the same zero betas every step
        bm(pose_body=pose, betas=zeros)
    with pytest.raises(PndfError):
        bm(pose_body=pose, betas=torch.ones(1, 10, device="cuda"))
    # a TEMPORARY zero tensor, then a fresh tensor that the caching allocator puts at the same address (same shape, version 0):
    # the validated tensor is remembered by identity, not by address
    bm(pose_body=pose, betas=torch.zeros(1, 10, device="cuda"))
    with pytest.raises(PndfError):
        bm(pose_body=pose, betas=torch.ones(1, 10, device="cuda"))
    # terms_grad: inputs are coerced, a wrong output buffer or shape is refused
    th_t = torch.from_numpy(th).cuda()
    j0 = bm.joints_of(th_t + 0.02)
    want = bm.terms_grad(th_t, j0, 2)
    strided = torch.empty(2, 9, 138, device="cuda")[:, :, ::2]
    strided.copy_(th_t)
    assert not strided.is_contiguous() and torch.equal(bm.terms_grad(strided, j0.double(), 2), want)
    assert torch.equal(bm.terms_grad(th_t.cpu(), j0.cpu(), 2), want)
    with pytest.raises(PndfError):
        bm.terms_grad(th_t, j0, 2, out=torch.empty(2, 9, 69, device="cuda", dtype=torch.float64))
    with pytest.raises(PndfError):
        bm.terms_grad(th_t, j0[:-1], 2)
    with pytest.raises(PndfError):
        bm.terms_grad(th_t.reshape(-1, 69), j0, 2)
