"""The rotation conversions around the hot path against an INDEPENDENT implementation: scipy.spatial.transform.Rotation.

pytorch3d (`axis_angle_to_quaternion`, `quaternion_to_axis_angle`: experiments/motion_denoise.py:81, sample_poses.py:60,80) and
smplx (`batch_rodrigues` inside `lbs()`: experiments/body_model.py:33-40) are third-party and absent, so the restatements here
cannot be pinned on the reference's own dependencies.  scipy implements the same mathematics independently:
Rodrigues' formula, rotation vector <-> unit quaternion (scalar first on request).  These tests hold every restatement -- the
torch ones the callers use, the numpy oracle's, and the HIP kernels' (`pndf_aa2quat`, the per-frame kernels of csrc/pndf_lbs.hip)
-- to it, including |theta| -> 0 (the series branches) and |theta| -> pi (where w changes sign)."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

PI = np.pi


def _rotvecs(dtype=np.float64):
    """random rotation vectors with angles spread over (0, 2 pi) plus the edges named in the module docstring"""
    rng = np.random.default_rng(12)
    axes = rng.normal(size=(64, 3))
    axes /= np.linalg.norm(axes, axis=1, keepdims=True)
    angles = np.concatenate([rng.uniform(0.0, PI, 40), rng.uniform(PI, 2 * PI - 0.05, 8),
                             [0.0, 1e-12, 1e-9, 3e-7, 0.99e-6, 1.01e-6, 1e-5, 1e-3, PI - 1e-4, PI - 1e-7, PI, PI + 1e-7, PI + 1e-4,
                              0.5 * PI, 1.5 * PI, 2 * PI - 1e-3]])
    return (axes * angles[:, None]).astype(dtype), angles


def _same_rotation(R_a, R_b, tol):
    assert np.abs(R_a - R_b).max() < tol, np.abs(R_a - R_b).max()


# ---------------------------------------------------------------------------------------------- CPU: torch restatements, oracle
def test_axis_angle_to_quaternion_matches_scipy():
    from posendf_amd.motion_denoise import axis_angle_to_quaternion
    r, ang = _rotvecs()
    q = axis_angle_to_quaternion(torch.from_numpy(r)).numpy()
    want = Rotation.from_rotvec(r).as_quat(scalar_first=True)           # [cos(t/2), axis sin(t/2)], not canonicalised
    assert np.abs(q - want).max() < 1e-14
    assert np.abs(np.linalg.norm(q, axis=1) - 1).max() < 1e-14
    # fp32, as the callers run it: the series branch below 1e-6 must not lose the vector part
    q32 = axis_angle_to_quaternion(torch.from_numpy(r.astype(np.float32))).numpy()
    assert np.abs(q32 - want).max() < 3e-7
    tiny = ang < 1e-5
    assert np.abs(q32[tiny, 1:] - 0.5 * r[tiny]).max() < 1e-12             # q_v = r / 2 to first order


def test_quaternion_to_axis_angle_matches_scipy():
    from posendf_amd.sample_poses import quaternion_to_axis_angle
    r, ang = _rotvecs()
    q = Rotation.from_rotvec(r).as_quat(scalar_first=True)
    back = quaternion_to_axis_angle(torch.from_numpy(q)).numpy()
    # pytorch3d's convention: angle = 2 atan2(|v|, w) in [0, 2 pi) -- the rotation vector itself comes back, also beyond pi
    assert np.abs(back - r).max() < 1e-9
    # scipy returns the short way round ([0, pi]); for w >= 0 the two agree as vectors, in general as rotations
    rv = Rotation.from_quat(q, scalar_first=True).as_rotvec()
    short = q[:, 0] >= 1e-9
    assert np.abs(back[short] - rv[short]).max() < 1e-9
    _same_rotation(Rotation.from_rotvec(back).as_matrix(), Rotation.from_quat(q, scalar_first=True).as_matrix(), 1e-12)
    # unnormalised input (the projected poses are not renormalised, sample_poses.py:74): the angle only depends on the direction
    scaled = quaternion_to_axis_angle(torch.from_numpy(q * 1.7)).numpy()
    _same_rotation(Rotation.from_rotvec(scaled * (np.linalg.norm(back, axis=1, keepdims=True)
                                                  / np.maximum(np.linalg.norm(scaled, axis=1, keepdims=True), 1e-300))).as_matrix(),
                   Rotation.from_rotvec(back).as_matrix(), 1e-9)


def test_round_trip_through_both_conversions():
    from posendf_amd.motion_denoise import axis_angle_to_quaternion
    from posendf_amd.sample_poses import quaternion_to_axis_angle
    r, _ = _rotvecs()
    back = quaternion_to_axis_angle(axis_angle_to_quaternion(torch.from_numpy(r))).numpy()
    assert np.abs(back - r).max() < 1e-9


def test_oracle_rodrigues_matches_scipy():
    from oracle import lbs_np
    r, ang = _rotvecs()
    R = lbs_np.batch_rodrigues(r)
    want = Rotation.from_rotvec(r).as_matrix()
    # smplx's `angle = ||r + 1e-8||` (lbs.py batch_rodrigues) is the formula's only deviation: ~1e-8 absolute
    _same_rotation(R, want, 5e-8)
    assert np.abs(R[ang == 0.0] - np.eye(3)).max() < 1e-15
    R32 = lbs_np.batch_rodrigues(r.astype(np.float32))
    _same_rotation(R32, want, 2e-6)


# ---------------------------------------------------------------------------------------------- GPU: the HIP kernels
@pytest.mark.gpu
def test_pndf_aa2quat_kernel_matches_scipy():
    import ctypes
    from posendf_amd.engine import load_library
    lib = load_library()
    r, ang = _rotvecs(np.float32)
    N = 7
    theta = np.zeros((N, 69), np.float32)
    rr = np.resize(r, (N * 21, 3))
    theta.reshape(N, 23, 3)[:, :21] = rr.reshape(N, 21, 3)
    th = torch.from_numpy(theta).cuda()
    q = torch.empty(N, 21, 4, device="cuda")
    assert lib.pndf_aa2quat(th.data_ptr(), q.data_ptr(), N, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    want = Rotation.from_rotvec(rr.astype(np.float64)).as_quat(scalar_first=True).reshape(N, 21, 4)
    got = q.cpu().numpy()
    assert np.abs(got - want).max() < 4e-7, np.abs(got - want).max()
    tiny = np.linalg.norm(rr, axis=1) < 1e-5
    assert np.abs(got.reshape(-1, 4)[tiny, 1:] - 0.5 * rr[tiny]).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("lbs_precision", ["f16x3", "fp32"])
def test_lbs_kernels_rodrigues_matches_scipy(lbs_precision):
    """Rodrigues inside the per-frame kernels, read back through skinned vertices: a body whose pose correctives are zero and
    whose vertices hang on ONE joint each (one-hot skinning weights) places vertex v at G_j (v - J_j) + t_j.  For joint 1 (a
    child of the fixed root) that is R(r_1) (v - J_1) + J_1 with everything but R known -- R comes from scipy."""
    from posendf_amd import BodyModel, synth
    V = 64
    m = synth.make_body_model(V=V, seed=8, extra=())
    m["posedirs"] = np.zeros_like(m["posedirs"])
    w = np.zeros((V, 24), np.float32)
    w[:, 1] = 1.0
    m["lbs_weights"] = w
    bm = BodyModel(m, device="cuda:0", precision=lbs_precision)
    r, _ = _rotvecs(np.float32)
    theta = np.zeros((len(r), 69), np.float32)
    theta[:, 0:3] = r                                  # body_pose[0:3] = SMPL joint 1 (the root's orientation is a constant)
    out = bm(pose_body=torch.from_numpy(theta))
    J = m["J_regressor"].astype(np.float64) @ m["v_template"].astype(np.float64)
    R = Rotation.from_rotvec(r.astype(np.float64)).as_matrix()
    want = np.einsum("nij,vj->nvi", R, m["v_template"].astype(np.float64) - J[1]) + J[1]
    got = out.vertices.cpu().numpy()
    scale = np.abs(want).max()
    assert np.abs(got - want).max() < 3e-6 * scale, np.abs(got - want).max() / scale
    # and the chain: joint 4 = child of joint 1 (SMPL tree) carries t_4 = R_1 (J_4 - J_1) + J_1
    want_j4 = np.einsum("nij,j->ni", R, J[4] - J[1]) + J[1]
    assert np.abs(out.Jtr.cpu().numpy()[:, 4] - want_j4).max() < 3e-6 * scale
