"""Pin oracle/denoise_np.py (the checker of the fused motion-denoise step) on vectors produced by the imported reference
network inside the reference's optimisation loop (tests/golden/make_golden_denoise.py).  CPU only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_weights
from oracle import denoise_np as dn


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(GOLDEN, "denoise_live.npz")))


@pytest.mark.parametrize("act", ["lrelu", "softplus"])
def test_loop_matches_reference_fp64(gold, act):
    sd = golden_weights("live")
    _, hist = dn.optimize(gold["theta0"].astype(np.float64), sd, iterations=2, steps_per_iter=4, act=act, trace=True)
    want = gold[f"{act}_theta_f64"]
    for k, th in enumerate(hist):
        err = np.abs(th - want[k]).max()
        assert err < 1e-9, (k, err)
    assert np.array_equal(hist[-1][:, 63:], gold["theta0"][:, 63:].astype(np.float64))     # hand joints never move


@pytest.mark.parametrize("act", ["lrelu", "softplus"])
def test_weighted_terms_match_reference(gold, act):
    sd = golden_weights("live")
    th = gold["theta0"].astype(np.float64)
    for it, prev in [(0, None), (1, 3)]:
        cur = th if prev is None else gold[f"{act}_theta_f64"][prev]
        _, terms = dn.step_gradient(cur, th, sd, it, act)
        ref = gold[f"{act}_terms_f64"][0 if prev is None else prev + 1]      # sorted keys: data, pose_pr, temp
        for name, val in zip(sorted(terms), ref):
            assert abs(terms[name] - val) <= 1e-9 * abs(val), (name, terms[name], val)


def test_fp32_loop_stays_close_to_fp64(gold):
    """Adam turns a gradient into a step of ~lr regardless of its size, so fp32 rounding is not amplified: the fp32
    loop tracks the fp64 one (the envelope the GPU tests use for the engine-backed loops)."""
    sd = golden_weights("live")
    out = dn.optimize(gold["theta0"], sd, iterations=2, steps_per_iter=4, dtype=np.float32)
    ref32, ref64 = gold["lrelu_theta_f32"][-1], gold["lrelu_theta_f64"][-1]
    assert np.abs(out - ref64).max() < 5e-4 and np.abs(ref32 - ref64).max() < 5e-4


def test_aa2quat_jacobian_finite_difference():
    rng = np.random.default_rng(0)
    a = rng.normal(size=(50, 3)) * 0.7
    a[0] = 0.0
    gq = rng.normal(size=(50, 4))
    an = dn.aa2quat_vjp(a, gq)
    h = 1e-6
    for e in range(3):
        da = np.zeros_like(a)
        da[:, e] = h
        fd = ((dn.axis_angle_to_quaternion(a + da)[0] - dn.axis_angle_to_quaternion(a - da)[0]) / (2 * h) * gq).sum(-1)
        assert np.allclose(fd[1:], an[1:, e], rtol=1e-5, atol=1e-8)
