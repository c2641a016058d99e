import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")
ACTS = ("lrelu", "relu", "softplus")
REGIMES = {"live": dict(seed=0, gain=2.0, out_bias=0.1), "mixed": dict(seed=0, gain=2.5, out_bias=0.05)}
# further reference-generated weight sets (tests/golden/make_golden.py LITE: single step, autograd contract, 1/10 steps)
LITE_REGIMES = {"s2g3": dict(seed=2, gain=3.0, out_bias=0.05), "s4g25": dict(seed=4, gain=2.5, out_bias=0.05),
                "s1g1": dict(seed=1, gain=1.0, out_bias=0.2)}
ALL_REGIMES = {**REGIMES, **LITE_REGIMES}
# the weight sets of the robustness sweep (tools/gpu_sweep.py, tests/test_gpu_sweep.py): (seed, gain, lin6.bias)
SWEEP_WEIGHTS = ((0, 2.0, 0.1), (0, 2.5, 0.05), (1, 1.0, 0.2), (2, 3.0, 0.05), (3, 0.5, 0.3), (4, 2.5, 0.05))
# HELD-OUT weight sets that never took part in calibrating a gate: PNDF_SWEEP_HELDOUT=1 (round 2's six) or =2 (six more, first
# run in round 3 after the outlier gate was tightened) swaps them into tests/test_gpu_sweep.py; results under profiles/.
_HELDOUT = {"1": ((5, 0.7, 0.3), (6, 1.5, 0.1), (7, 2.0, 0.0), (8, 3.5, 0.02), (9, 4.0, 0.1), (10, 2.2, -0.05)),
            "2": ((11, 0.6, 0.25), (12, 1.2, 0.15), (13, 1.8, 0.0), (14, 2.8, 0.03), (15, 3.2, 0.08), (16, 2.4, -0.03)),
            # =3: six more, first run in round 4 after the softplus activation was rewritten (packed fp32, clamp instead of selects)
            "3": ((21, 0.8, 0.2), (22, 1.4, 0.1), (23, 2.1, 0.0), (24, 3.0, 0.04), (25, 3.6, 0.06), (26, 1.7, -0.02)),
            # =4: six more, first run at the end of round 6 (no gate changed in round 6: a pure re-check on weights never seen)
            "4": ((31, 0.9, 0.15), (32, 1.6, 0.05), (33, 2.3, 0.0), (34, 2.7, 0.07), (35, 3.3, 0.03), (36, 1.1, -0.04))}
SWEEP_WEIGHTS = _HELDOUT.get(os.environ.get("PNDF_SWEEP_HELDOUT", ""), SWEEP_WEIGHTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    # The GPU box shows 256 hardware threads; numpy's BLAS would spread every small oracle matmul ([100..256] x 1024) over
    # all of them and spend its time in thread hand-offs (the -m gpu suite took 13 minutes of billed box time that way).
    try:
        from threadpoolctl import threadpool_limits
        config._blas_limit = threadpool_limits(limits=min(16, os.cpu_count() or 1))
    except ImportError:
        pass


def load_golden(act, regime):
    return dict(np.load(os.path.join(GOLDEN, f"posendf_{act}_{regime}.npz")))


def golden_weights(regime):
    from posendf_amd import synth
    return synth.make_weights(**ALL_REGIMES[regime])


def rel_err(a, b):
    """Per-pose relative error, the parity metric used throughout (DESIGN.md 'Parity metric'):
    for every pose (row) max_i |a_i - b_i| / max(max_i |b_i|, tiny); returns the max over poses.
    For [B,1] distances this is the plain elementwise relative error (exact zeros must match
    exactly or within 1e-30)."""
    return float(rel_err_rows(a, b).max())


def rel_err_rows(a, b, floor_frac=1e-3):
    """Per-pose max|a-b| / max(max|b_row|, floor_frac * typical row scale), typical = the batch MEDIAN of max|b_row|.
    The floor (0.1 % of the typical scale) matters only for poses whose whole row is ~0 relative to the batch --
    e.g. softplus poses with d ~ 1e-6 whose gradient is 1e-4 of the typical one: there the reference's own fp32 run
    is only good to 7e-5 relative (exp() of a large cancelling argument) although its absolute error is negligible.
    The median, not the maximum: the golden batches contain eps-clamp edge poses whose gradient is ~1e10, and a
    floor tied to them would hide every error of the ordinary rows."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    a = a.reshape(a.shape[0], -1)
    b = b.reshape(b.shape[0], -1)
    num = np.abs(a - b).max(axis=1)
    rows = np.abs(b).max(axis=1)
    den = np.maximum(rows, max(floor_frac * float(np.median(rows)), 1e-30))
    return np.where(num == 0, 0.0, num / den)


def d_err(a, b, floor_frac=0.05):
    """Distance parity metric: |a-b| / max(|b|, floor_frac * max|b|), max over poses.  The floor
    exists because d = relu(sum of 64 cancelling terms): for poses whose d is far below the batch
    scale the REFERENCE's own fp32 result is only good to ~1e-4 relative against its fp64 run
    (tests/golden 'mixed' regime: 9.9e-5), so a purely elementwise 1e-4 gate would fail the
    reference against itself.  In the 'live' benchmark regime the floor is inactive."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    den = np.maximum(np.abs(b), floor_frac * max(np.abs(b).max(), 1e-30))
    return float((np.abs(a - b) / den).max())


def d_rows(a, b, floor_frac=0.05):
    """per-pose form of d_err"""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return np.abs(a - b) / np.maximum(np.abs(b), floor_frac * max(np.abs(b).max(), 1e-30))


_MEMO = {}


def _key(*parts):
    """content key of the oracle-side envelopes: the fp32 / f16x3 parametrisations of a test share them (they are pure
    functions of poses, weights and activation), and on the GPU box oracle time is billed like kernel time"""
    import hashlib
    h = hashlib.sha1()
    for p in parts:
        if isinstance(p, dict):
            for k in sorted(p):
                h.update(k.encode())
                h.update(np.ascontiguousarray(p[k]).tobytes())
        elif isinstance(p, np.ndarray):
            h.update(np.ascontiguousarray(p).tobytes())
        else:
            h.update(repr(p).encode())
    return h.hexdigest()


def fp32_noise(q, sd, act, draws=8, extra_d=(), extra_g=(), seed=1):
    """memoised front of _fp32_noise (the `extra_*` samples are folded in afterwards)"""
    k = _key("noise", np.asarray(q, np.float32), sd, act, draws, seed)
    if k not in _MEMO:
        _MEMO[k] = _fp32_noise(q, sd, act, draws, (), (), seed)
    sig_d, sig_g, d64, g64 = _MEMO[k]
    for e in extra_d:
        sig_d = np.maximum(sig_d, np.asarray(e, dtype=np.float64))
    for e in extra_g:
        sig_g = np.maximum(sig_g, np.asarray(e, dtype=np.float64))
    return sig_d, sig_g, d64, g64


def _fp32_noise(q, sd, act, draws=8, extra_d=(), extra_g=(), seed=1):
    """Per-pose fp32 sensitivity of the REFERENCE arithmetic: the largest error against the fp64 run that the fp32 oracle
    makes over `draws` evaluations whose inputs are perturbed by one fp32 rounding (q (1 + e), |e| <= 2^-23), plus any
    further reference-arithmetic samples (`extra_*`: per-pose error vectors, e.g. the reference's own fp32 run from the
    golden file).  A well-conditioned pose gets ~1e-6; a pose whose d is the small difference of large terms, or a
    softplus pose with exp(beta z) of a large cancelling z, gets what fp32 can actually deliver there.  Returns
    (sigma_d [B], sigma_g [B], d64, g64) in the metrics of d_rows / rel_err_rows."""
    from oracle import posendf_np as onp
    q = np.asarray(q, dtype=np.float32)
    d64, g64 = onp.forward_grad(q, sd, act, dtype=np.float64)
    rng = np.random.default_rng(seed)
    sig_d = [np.asarray(e, dtype=np.float64) for e in extra_d]
    sig_g = [np.asarray(e, dtype=np.float64) for e in extra_g]
    for _ in range(draws):
        qk = (q * (1 + rng.uniform(-2.0 ** -23, 2.0 ** -23, q.shape))).astype(np.float32)
        d32, g32 = onp.forward_grad(qk, sd, act, dtype=np.float32)
        sig_d.append(d_rows(d32, d64))
        sig_g.append(rel_err_rows(g32, g64))
    return np.max(sig_d, axis=0), np.max(sig_g, axis=0), d64, g64


def escalated_noise(q, sd, act, idx, truth, draws=32, seed=11, steps=0, kind="g"):
    """Second, better estimate of the fp32 variability of the REFERENCE arithmetic, for the few poses `idx` that exceeded the
    cheap envelope (round 4; found by a fresh held-out sweep, profiles/r04/sweep_heldout.txt).  The cheap envelopes perturb
    the INPUT by one fp32 rounding (8 single-step draws, 3 trajectory draws): that under-samples (i) heavy-tailed trajectories
    -- one softplus pose of a gain-3.6 network: 1.5e-5 over 3 draws, 1.24e-4 over 100 -- and (ii) the freedom of the
    evaluation ORDER: a correct fp32 evaluation may round every product differently (BLAS blocking, MFMA accumulation
    order), which one-rounding perturbations of the WEIGHTS model -- one lrelu pose whose pre-activation sits 7e-5 from a
    kink: 2.5e-6 under 48 input perturbations, 3.3e-5 under 32 weight perturbations, the exact-fp32 kernel 3.2e-5.
    Here: `draws` fp32 oracle evaluations of the poses `idx` with inputs AND weights perturbed by one rounding each;
    returns the per-pose maximum error against `truth` (the fp64 result of the full batch: the metric keeps the full
    batch's floors).  steps = 0: single step (kind "d" or "g"); steps > 0: the projected poses after `steps` steps."""
    from oracle import posendf_np as onp
    idx = np.asarray(idx)
    q = np.asarray(q, np.float32)
    truth = np.asarray(truth, np.float64)
    t_rows = truth.reshape(truth.shape[0], -1)
    if kind == "d":
        floor = 0.05 * max(np.abs(t_rows).max(), 1e-30)
    else:
        floor = max(1e-3 * float(np.median(np.abs(t_rows).max(axis=1))), 1e-30)
    den = np.maximum(np.abs(t_rows[idx]).max(axis=1), floor)
    rng = np.random.default_rng(seed)
    worst = np.zeros(len(idx))
    for _ in range(draws):
        qk = (q[idx] * (1 + rng.uniform(-2.0 ** -23, 2.0 ** -23, q[idx].shape))).astype(np.float32)
        sdk = {k: (v * (1 + rng.uniform(-2.0 ** -23, 2.0 ** -23, v.shape))).astype(np.float32) for k, v in sd.items()}
        if steps > 0:
            out, _ = onp.project(qk, sdk, steps=steps, act=act, dtype=np.float32)      # the reference's arithmetic, explicitly
        else:
            d32, g32 = onp.forward_grad(qk, sdk, act, dtype=np.float32)
            out = d32 if kind == "d" else g32
        num = np.abs(np.asarray(out, np.float64).reshape(len(idx), -1) - t_rows[idx]).max(axis=1)
        worst = np.maximum(worst, num / den)
    return worst


ESCALATIONS = []      # every pose a gate excused through `escalate` (counted, printed at the end of the session)


def pytest_terminal_summary(terminalreporter):
    expl = [e for e in ESCALATIONS if e[3]]
    terminalreporter.write_line(f"[escalations] {len(ESCALATIONS)} pose(s) went through the escalated envelope, {len(expl)} explained by it: "
                                + "; ".join(f"{w} pose {i} err {e:.2e}" for w, i, e, _ in ESCALATIONS[:12]))


def pose_gate(err, sigma, what="", factor=8.0, floor=8e-6, exempt=None, tol=1e-4, escalate=None):
    """Every pose individually: error <= factor x the pose's fp32 sensitivity (fp32_noise) + floor.  Calibrated on
    tools/gpu_sweep.py (6 weight sets x 3 activations x 2 pose distributions x 1,024 poses, both kernels): the largest
    error / (sigma + 1e-6) observed is 6.3, the 99th percentile 2.7.  `exempt`: poses that are allowed to exceed it
    (relu family: poses with a pre-activation within 1e-5 of a kink, where the derivative legitimately flips).
    Also holds the batch to the headline bar where the reference arithmetic itself meets it: median <= tol / 10."""
    err, sigma = np.asarray(err, dtype=np.float64), np.asarray(sigma, dtype=np.float64)
    bad = err > factor * sigma + floor
    if exempt is not None:
        bad &= ~np.asarray(exempt)
    if bad.any() and escalate is not None and bad.sum() <= max(4, len(err) // 100):
        # few poses over the cheap envelope: measure the reference arithmetic's variability there properly (escalated_noise)
        # and hold them to TWICE the worst of those evaluations -- a tighter factor for a better estimate
        idx = np.flatnonzero(bad)
        s2 = np.asarray(escalate(idx), dtype=np.float64)
        still = err[idx] > 2.0 * s2 + floor
        print(f"[pose_gate {what}] escalated {idx.tolist()}: error {err[idx].tolist()} cheap sigma {sigma[idx].tolist()} "
              f"-> worst of the escalated reference evaluations {s2.tolist()}: {'FAIL' if still.any() else 'explained'}")
        bad[idx] = still
        ESCALATIONS.extend((f"pose_gate {what}", int(i), float(err[i]), not bool(st)) for i, st in zip(idx, still))
    assert not bad.any(), (what, int(bad.sum()), np.flatnonzero(bad)[:8].tolist(), err[bad][:8].tolist(),
                           sigma[bad][:8].tolist())
    assert np.median(err) <= max(tol / 10, factor * float(np.median(sigma))), (what, float(np.median(err)))
    keep = np.ones(len(err), bool) if exempt is None else ~np.asarray(exempt)
    ratio = err[keep] / (factor * sigma[keep] + floor)
    worst = float(ratio.max()) if ratio.size else 0.0
    print(f"[pose_gate {what}] n {len(err)} median {np.median(err):.2e} max {err.max():.2e} | largest error / bound "
          f"{worst:.2f} (1.00 = at the gate), exempt {0 if exempt is None else int(np.sum(exempt))}")
    return worst


def outlier_gate(mine_rows, ref_rows, tol=1e-4, what="", ratio=2.0, margin=None, kink_tol=1e-5, cap=10.0, sigma=None, escalate=None):
    """Gate for quantities that are DISCONTINUOUS in the input (d d/d q and everything derived from it):
    a pre-activation within rounding of a ReLU/LeakyReLU kink flips its derivative (1 vs slope), so any two
    fp32 evaluations -- including the reference's own fp32 run against its fp64 run -- disagree by O(1) on a
    few poses per thousand (measured: 1/1000 and 2/4096 poses for the numpy oracle, 3 % after 100 steps).
    Both error vectors are per-pose relative errors against the SAME fp64 truth; `ref_rows` is the
    reference-arithmetic (fp32) run that sets the envelope.  `ratio`: how many times the reference's own outlier
    fraction is tolerated (the kink-crossing probability is proportional to the size of the rounding perturbation;
    both kernels are held to 2).

    Outlier MAGNITUDE (round 3; the fraction alone let single poses be arbitrarily wrong).  Every pose above `tol` must be
    EXPLAINED: either the reference arithmetic's own fp32 run is off at that pose too (ref_rows > tol / 4: an
    ill-conditioned pose or a trajectory that the reference itself cannot hold), or the error is within 8 x the fp32
    sensitivity of the reference arithmetic at that pose (`sigma`: fp32_noise / traj_envelope -- one fp32 run may be lucky
    where eight perturbed ones are not: the s4g25 softplus poses), or -- relu family, when the caller
    supplies `margin` (oracle kink_margin / trajectory_kink_margin of the fp64 run) -- a pre-activation came within
    `kink_tol` of a kink, where the derivative may legitimately flip.  Unexplained outliers fail.  Outliers explained by
    the reference's own error are capped at `cap` x the reference run's largest error; kink poses, whose error after a
    flip is whatever the other branch of the network gives, must stay finite and below 100 % (a diverged pose is not a
    flipped kink); their number is bounded by the fraction gate."""
    mine_rows = np.asarray(mine_rows, dtype=np.float64)
    ref_rows = np.asarray(ref_rows, dtype=np.float64)
    n = len(mine_rows)
    slack = max(0.003, 2.0 / n)
    assert np.isfinite(mine_rows).all(), (what, "non-finite error", int((~np.isfinite(mine_rows)).sum()))
    # batch median: a tenth of the bar -- or, for a network on which the reference arithmetic's own fp32 median is worse than
    # that (held-out set 2, seed 14 gain 2.8 softplus: 1.3e-5 for the fp32 oracle itself), `ratio` x the reference's median
    assert np.median(mine_rows) < max(tol / 10, ratio * float(np.median(ref_rows))), (what, float(np.median(mine_rows)),
                                                                                    float(np.median(ref_rows)))
    frac, ref_frac = float((mine_rows > tol).mean()), float((ref_rows > tol).mean())
    assert frac <= ratio * ref_frac + slack, (what, frac, ref_frac, float(mine_rows.max()), float(ref_rows.max()))
    # p95 inside the bar wherever the reference arithmetic's own p95 is
    if np.percentile(ref_rows, 95) <= tol / 2:
        assert np.percentile(mine_rows, 95) <= tol, (what, float(np.percentile(mine_rows, 95)))
    out = mine_rows > tol
    by_ref = ref_rows > tol / 4
    by_kink = np.zeros(n, bool) if margin is None else (np.asarray(margin) < kink_tol)
    by_sigma = np.zeros(n, bool) if sigma is None else (mine_rows <= 8.0 * np.asarray(sigma, dtype=np.float64) + 8e-6)
    unexplained = out & ~by_ref & ~by_kink & ~by_sigma
    if unexplained.any() and escalate is not None and unexplained.sum() <= max(4, n // 100):
        idx = np.flatnonzero(unexplained)       # (see pose_gate: escalated_noise, factor 2 on the better estimate)
        s2 = np.asarray(escalate(idx), dtype=np.float64)
        still = mine_rows[idx] > 2.0 * s2 + 8e-6
        print(f"[gate {what}] escalated {idx.tolist()}: error {mine_rows[idx].tolist()} -> worst of the escalated reference "
              f"evaluations {s2.tolist()}: {'FAIL' if still.any() else 'explained'}")
        unexplained[idx] = still
        ESCALATIONS.extend((f"gate {what}", int(i), float(mine_rows[i]), not bool(st)) for i, st in zip(idx, still))
        by_sigma = by_sigma.copy()
        by_sigma[idx[~still]] = True
    assert not unexplained.any(), (what, "outliers that neither the reference's fp32 error nor a kink explains",
                                   np.flatnonzero(unexplained)[:8].tolist(), mine_rows[unexplained][:8].tolist(),
                                   ref_rows[unexplained][:8].tolist())
    capped = out & by_ref & ~by_kink & ~by_sigma
    if capped.any():
        assert mine_rows[capped].max() <= cap * max(float(ref_rows.max()), tol), (
            what, "outlier magnitude", float(mine_rows[capped].max()), float(ref_rows.max()))
    kinked = out & by_kink
    if kinked.any():
        # (their NUMBER is bounded by the fraction gate above: at most `ratio` x the reference run's own outliers + slack)
        assert mine_rows[kinked].max() < 1.0, (what, "kink outliers", int(kinked.sum()), float(mine_rows[kinked].max()))
    print(f"[gate {what}] n {n} median {np.median(mine_rows):.2e} p95 {np.percentile(mine_rows, 95):.2e} max {mine_rows.max():.2e}"
          f" | ref p95 {np.percentile(ref_rows, 95):.2e} max {ref_rows.max():.2e} | outliers {int(out.sum())}"
          f" (ref-explained {int((out & by_ref).sum())}, kink {int(kinked.sum())})")


def traj_envelope(q, sd, act, steps, truth_q, draws=3, seed=1, truth_d=None, d_metric=None):
    k = _key("traj", np.asarray(q, np.float32), sd, act, steps, draws, seed, np.asarray(truth_q), truth_d is not None,
             None if truth_d is None else np.asarray(truth_d), None if d_metric is None else d_metric.__code__.co_code)
    if k not in _MEMO:
        _MEMO[k] = _traj_envelope(q, sd, act, steps, truth_q, draws, seed, truth_d, d_metric)
    return _MEMO[k]


def _traj_envelope(q, sd, act, steps, truth_q, draws=3, seed=1, truth_d=None, d_metric=None):
    """`margin` and `sigma` arguments of outlier_gate for a free-running `steps`-step projection from q: the kink margins
    along the fp64 trajectory (traj_margin) and the per-pose fp32 sensitivity of the REFERENCE arithmetic -- the largest
    error against `truth_q` (the fp64 trajectory's end point) over `draws` fp32 oracle trajectories whose inputs are
    perturbed by one fp32 rounding.  With `truth_d` (+ `d_metric(d, truth_d)` -> per-pose errors, default d_rows) a second
    envelope for the distance of the last iteration is returned as well: (env_q, env_d)."""
    from oracle import posendf_np as onp
    q = np.asarray(q, dtype=np.float32)
    rng = np.random.default_rng(seed)
    sig, sig_d = [], []
    for _ in range(draws):
        qk = (q * (1 + rng.uniform(-2.0 ** -23, 2.0 ** -23, q.shape))).astype(np.float32)
        qo, do = onp.project(qk, sd, steps=steps, act=act)
        sig.append(rel_err_rows(qo, truth_q))
        if truth_d is not None:
            sig_d.append((d_metric or d_rows)(do.reshape(-1), truth_d))
    env = dict(margin=traj_margin(q, sd, act, steps), sigma=np.max(sig, axis=0))
    return env if truth_d is None else (env, dict(margin=env["margin"], sigma=np.max(sig_d, axis=0)))


def traj_margin(q, sd, act, steps=1):
    """`margin` argument of outlier_gate: the smallest kink margin of each pose over the `steps` evaluations of its fp64
    projection trajectory (steps = 1: a single forward + gradient at q); None for softplus, which has no kinks."""
    from oracle import posendf_np as onp
    if act == "softplus":
        return None
    k = _key("margin", np.asarray(q, np.float32), sd, act, int(steps))
    if k not in _MEMO:
        _MEMO[k] = onp.trajectory_kink_margin(q, sd, max(int(steps), 1), act)
    return _MEMO[k]


@pytest.fixture(params=[(a, r) for a in ACTS for r in ALL_REGIMES], ids=lambda p: f"{p[0]}-{p[1]}")
def golden_case(request):
    act, regime = request.param
    return act, regime, load_golden(act, regime), golden_weights(regime)
