"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the motion-denoise optimiser step
(reference experiments/motion_denoise.py:58-99) around the numpy Pose-NDF oracle, with ANALYTIC gradients: the checker
of the fused HIP step (posendf_amd/csrc/pndf_denoise.hip, C ABI pndf_aa2quat / pndf_denoise_update) and of the autograd
driver (posendf_amd/motion_denoise.py).

Parity status: PINNED for the loop structure, the weight schedule, the pose-prior term and Adam --
tests/golden/denoise_live.npz was produced by tests/golden/make_golden_denoise.py, which runs the imported reference
PoseNDF inside the restated loop with torch autograd and torch.optim.Adam; tests/test_denoise_oracle.py checks this
module against it (fp64: 1e-9).  UNPINNED by any reference test: pytorch3d's axis_angle_to_quaternion (restated from its
documented convention) and the SMPL vertex / joint terms, which are replaced by pose-space surrogates.

  pose_pr : 1e7 c^2 / (1 + it),  c = mean_t d(q_t)              (:29-35, :81-83)
  temp    : 10 (1 + it) mean_{t, j} ||th_{t, j} - th_{t+1, j}||    (:88-89 on the surrogate "vertices")
  data    : 100 / (1 + it) mean_{t, j} ||th_{t, j} - th0_{t, j}||  (:92-94, only for it > 0)
  Adam(lr = 0.02, betas = (0.9, 0.999), eps = 1e-8), torch.optim.Adam semantics  (:70, :98-99)
"""
from __future__ import annotations

import numpy as np

from . import posendf_np as onp

NJ = 21
# (coefficient, power of the loss, exponent of (1 + it)) per term: experiments/motion_denoise.py:29-35 and the copy of the loop
# in experiments/partial_observation.py:29-35 (pose prior LINEAR in the mean distance)
SCHEDULES = {"motion_denoise": {"temp": (10.0, 1, 1), "data": (100.0, 1, -1), "pose_pr": (1e7, 2, -1)},
             "partial_observation": {"temp": (100.0, 1, 1), "data": (10.0, 1, -1), "pose_pr": (100.0, 1, -1)}}
SURROGATE_EPS = 1e-20      # inside the sqrt of the surrogate norms (a zero difference has a zero, not a NaN, gradient)


def axis_angle_to_quaternion(a):
    """pytorch3d convention: real part first, q = [cos(|a|/2), a sin(|a|/2)/|a|]; series 1/2 - |a|^2/48 below 1e-6."""
    ang = np.sqrt((a * a).sum(-1, keepdims=True))
    small = ang < 1e-6
    k = np.where(small, 0.5 - ang * ang / 48.0, np.sin(0.5 * ang) / np.where(small, 1.0, ang))
    return np.concatenate([np.cos(0.5 * ang), a * k], axis=-1), ang, k


def aa2quat_vjp(a, gq):
    """gradient w.r.t. the axis-angle vector given d L / d q [.., 4]  (reverse pass of axis_angle_to_quaternion)."""
    q, ang, k = axis_angle_to_quaternion(a)
    small = ang < 1e-6
    kp = np.where(small, -1.0 / 24.0, (0.5 * q[..., :1] - k) / np.where(small, 1.0, ang * ang))     # k'(ang) / ang
    gv_dot_a = (gq[..., 1:] * a).sum(-1, keepdims=True)
    common = -0.5 * k * gq[..., :1] + kp * gv_dot_a
    return common * a + k * gq[..., 1:]


def step_gradient(theta, theta0, sd, it, act="lrelu", beta=100.0, dtype=np.float64, body=None, schedule="motion_denoise"):
    """d (total weighted loss) / d theta for ONE sequence theta [T,69]; returns (grad [T,69], weighted terms dict).
    body = (model dict of oracle/lbs_np.py, joints0 [T, n_joints, 3]): the reference's SMPL vertex temporal term and joint
    data term (motion_denoise.py:86-94; oracle/lbs_np.body_terms, parity unpinned) instead of the pose-space surrogates."""
    theta = np.asarray(theta, dtype)
    T = theta.shape[0]
    a = theta.reshape(T, 23, 3)[:, :NJ]
    q, _, _ = axis_angle_to_quaternion(a)
    d, dq = onp.forward_grad(q.astype(dtype), sd, act, beta, dtype)
    c = d.mean(dtype=dtype)
    sch = SCHEDULES[schedule]
    ev = lambda k: dtype(sch[k][0]) * dtype(1 + it) ** sch[k][2]
    pc, pp = ev("pose_pr"), sch["pose_pr"][1]
    w_temp, w_data = ev("temp"), ev("data")
    terms = {"pose_pr": pc * c ** pp}
    g = np.zeros((T, 23, 3), dtype)
    g[:, :NJ] = aa2quat_vjp(a, dq * ((dtype(2) * pc * c if pp == 2 else pc) / dtype(T)))
    if body is not None:
        from . import lbs_np
        gb, bt = lbs_np.body_terms(theta, body[1], body[0], it, dtype, coefs=(w_temp, w_data))
        if "temp" in bt:
            terms["temp"] = w_temp * bt["temp"]
        if "data" in bt:
            terms["data"] = w_data * bt["data"]
        return g.reshape(T, 69) + gb, terms
    if T > 1:
        diff = a[:-1] - a[1:]
        nrm = np.sqrt((diff * diff).sum(-1, keepdims=True) + dtype(SURROGATE_EPS))
        w = w_temp / dtype((T - 1) * NJ)
        terms["temp"] = w_temp * nrm.mean(dtype=dtype)
        g[:-1, :NJ] += w * diff / nrm
        g[1:, :NJ] -= w * diff / nrm
    if it > 0:
        diff = a - np.asarray(theta0, dtype).reshape(T, 23, 3)[:, :NJ]
        nrm = np.sqrt((diff * diff).sum(-1, keepdims=True) + dtype(SURROGATE_EPS))
        terms["data"] = w_data * nrm.mean(dtype=dtype)
        g[:, :NJ] += w_data / dtype(T * NJ) * diff / nrm
    return g.reshape(T, 69), terms


def optimize(theta0, sd, iterations=10, steps_per_iter=50, lr=0.02, act="lrelu", beta=100.0, dtype=np.float64,
             trace=False, body_model=None, schedule="motion_denoise"):
    """The loop of MotionDenoise.optimize (:70-99) for one sequence [T,69] or a batch of independent ones [S,T,69]."""
    theta0 = np.asarray(theta0, dtype)
    single = theta0.ndim == 2
    th0 = theta0[None] if single else theta0
    th = th0.copy()
    m, v = np.zeros_like(th), np.zeros_like(th)
    bodies = [None] * th.shape[0]
    if body_model is not None:          # smpl_init.Jtr of the noisy poses (motion_denoise.py:60,63)
        from . import lbs_np
        bodies = [(body_model, lbs_np.lbs(th0[s], body_model, dtype)[1]) for s in range(th.shape[0])]
    b1, b2, eps = dtype(0.9), dtype(0.999), dtype(1e-8)
    k = 0
    hist = []
    for it in range(iterations):
        for _ in range(steps_per_iter):
            k += 1
            g = np.stack([step_gradient(th[s], th0[s], sd, it, act, beta, dtype, bodies[s], schedule)[0] for s in range(th.shape[0])])
            m = b1 * m + (1 - b1) * g
            v = b2 * v + (1 - b2) * g * g
            bc1, bc2 = 1 - b1 ** k, 1 - b2 ** k
            th = th - (dtype(lr) / bc1) * m / (np.sqrt(v) / np.sqrt(bc2) + eps)        # torch.optim.Adam (no amsgrad)
            if trace:
                hist.append(th[0].copy() if single else th.copy())
    out = th[0] if single else th
    return (out, hist) if trace else out
