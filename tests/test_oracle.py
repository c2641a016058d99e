"""Pin the numpy oracle (oracle/posendf_np.py) against vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only.  Tolerances: single-step quantities 2e-5 relative in fp32
(both sides are fp32 with different summation orders), 1e-9 in fp64."""
import numpy as np
import pytest

from conftest import LITE_REGIMES, d_err, d_rows, fp32_noise, pose_gate, rel_err, rel_err_rows
from oracle import posendf_np as onp


def kink_exempt(g, sd, act):
    """relu family: poses with a pre-activation within 1e-5 (relative) of a kink may flip a derivative"""
    return None if act == "softplus" else onp.kink_margin(g["q"], sd, act) < 1e-5


def test_single_step_fp32(golden_case):
    act, regime, g, sd = golden_case
    dbg = {}
    d, dq = onp.forward_grad(g["q"], sd, act, debug=dbg)
    assert rel_err(dbg["n"], g["n_f32"]) < 1e-5
    assert rel_err(dbg["feat"], g["feat_f32"]) < 1e-5
    if regime in LITE_REGIMES:
        # ill-conditioned weight sets: two fp32 evaluations (numpy here, torch in the fixture) legitimately differ by more
        # than 2e-5 on the poses whose d is a small difference of large terms -- each is held, pose by pose, to the fp32
        # sensitivity of the arithmetic (conftest.fp32_noise), both against the reference's fp64 run
        sig_d, sig_g, _, _ = fp32_noise(g["q"], sd, act)
        pose_gate(d_rows(d, g["d_f64"]), sig_d, "oracle d")
        pose_gate(d_rows(g["d_f32"], g["d_f64"]), sig_d, "reference d")
        ex = kink_exempt(g, sd, act)
        pose_gate(rel_err_rows(dq, g["dq_f64"]), sig_g, "oracle dq", exempt=ex)
        pose_gate(rel_err_rows(g["dq_f32"], g["dq_f64"]), sig_g, "reference dq", exempt=ex)
    else:
        assert d_err(d, g["d_f32"]) < 2e-5
        assert rel_err(dq, g["dq_f32"]) < 5e-5
        assert d_err(onp.forward(g["q"], sd, act), g["d_f32"]) < 2e-5
    # clipped poses are clipped on both sides, and their gradient is exactly zero
    if act != "softplus":
        z = g["d_f32"][:, 0] == 0
        assert np.array_equal(d[:, 0] == 0, z)
        assert np.all(dq[z] == 0)


def test_single_step_fp64(golden_case):
    act, regime, g, sd = golden_case
    d, dq = onp.forward_grad(g["q"], sd, act, dtype=np.float64)
    assert d_err(d, g["d_f64"]) < 1e-9
    assert rel_err(dq, g["dq_f64"]) < 1e-9


def test_autograd_contract(golden_case):
    act, regime, g, sd = golden_case
    _, gp = onp.forward_grad(g["q"], sd, act, grad_out=g["grad_out"])
    if regime in LITE_REGIMES:
        truth = g["dq_f64"] * g["grad_out"].reshape(-1, 1, 1)
        _, sig_g, _, _ = fp32_noise(g["q"], sd, act)
        ex = kink_exempt(g, sd, act)
        pose_gate(rel_err_rows(gp, truth), sig_g, "oracle grad_pose", exempt=ex)
        pose_gate(rel_err_rows(g["grad_pose_f32"], truth), sig_g, "reference grad_pose", exempt=ex)
    else:
        assert rel_err(gp, g["grad_pose_f32"]) < 5e-5


def test_pose_prior_objective(golden_case):
    act, regime, g, sd = golden_case
    if regime in LITE_REGIMES:
        pytest.skip("lite fixture: no pose-prior vectors")
    for it in (0, 3):
        obj, grad = onp.pose_prior_objective(g["q"], sd, it, act)
        assert abs(obj - g[f"prior_obj_it{it}"]) <= 2e-5 * abs(g[f"prior_obj_it{it}"])
        assert rel_err(grad, g[f"prior_grad_it{it}"]) < 1e-4


def test_projection_fp64(golden_case):
    """fp64 trajectories agree to ~1e-9 unless a ReLU/LeakyReLU kink is crossed within rounding."""
    act, regime, g, sd = golden_case
    last = 10 if regime in LITE_REGIMES else 100
    q, d, tr = onp.project(g["q"], sd, steps=last, act=act, dtype=np.float64, trace=True)
    e = np.abs(q - g[f"q{last}_f64"]).reshape(len(q), -1).max(1) / np.abs(g[f"q{last}_f64"]).max()
    assert np.median(e) < 1e-9
    assert (e > 1e-6).mean() <= 0.03
    assert rel_err(tr[0], g["dtrace_f64"][0]) < 1e-9


def test_projection_fp32_envelope(golden_case):
    """Free-running fp32 vs the reference's fp64 truth: the reference's own fp32 run sets the envelope
    (SURVEY.md section 7 'hard parts'); the oracle must be no worse than 2x + slack."""
    act, regime, g, sd = golden_case
    for steps in ((1, 10) if regime in LITE_REGIMES else (1, 10, 100)):
        q, _ = onp.project(g["q"], sd, steps=steps, act=act)
        truth = g[f"q{steps}_f64"]
        scale = np.abs(truth).max()
        mine = np.abs(q - truth).reshape(len(q), -1).max(1) / scale
        ref = np.abs(g[f"q{steps}_f32"] - truth).reshape(len(q), -1).max(1) / scale
        assert np.median(mine) < 1e-5
        assert (mine > 1e-4).mean() <= 2 * (ref > 1e-4).mean() + 0.03, (steps, mine.max(), ref.max())


def test_per_pose_independence():
    from conftest import golden_weights
    from posendf_amd import synth
    sd = golden_weights("mixed")
    q = synth.make_poses(64, seed=3, signed=True)
    d, dq = onp.forward_grad(q, sd)
    perm = np.random.default_rng(0).permutation(64)
    d2, dq2 = onp.forward_grad(q[perm], sd)
    assert np.allclose(d[perm], d2, rtol=1e-6, atol=1e-7)
    assert np.allclose(dq[perm], dq2, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("act", ["lrelu", "softplus"])
def test_gradient_finite_difference(act):
    from conftest import golden_weights
    from posendf_amd import synth
    sd = golden_weights("live")
    q = synth.make_poses(4, seed=4, signed=True).astype(np.float64)
    d, dq = onp.forward_grad(q, sd, act, dtype=np.float64)
    rng = np.random.default_rng(1)
    for _ in range(6):
        v = rng.normal(size=q.shape)
        h = 1e-6
        fd = (onp.forward(q + h * v, sd, act, dtype=np.float64) - onp.forward(q - h * v, sd, act, dtype=np.float64)) / (2 * h)
        an = (dq * v).reshape(4, -1).sum(1)
        assert np.allclose(fd[:, 0], an, rtol=2e-4, atol=1e-9)
