"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of linear-blend skinning as the SMPL body model of
`smplx` computes it, and of the two body-model terms of the reference's motion-denoise objective, with ANALYTIC gradients
with respect to the body pose.  Checker of posendf_amd/csrc/pndf_lbs.hip (C ABI pndf_lbs_*).

Parity status: **UNPINNED**.  The reference calls a third-party package: experiments/body_model.py:7-9,27-29 builds
`smplx.SMPL(bm_path, num_betas, batch_size)` and :35-40 calls it with `betas`, `global_orient=None`, `body_pose`;
experiments/motion_denoise.py:86-94 turns its outputs into the temporal and data terms.  `smplx` is neither vendored under
/root/reference nor listed in its requirements.txt (version unknown), and the licensed SMPL model files are not reachable
offline -- so there is no reference output to pin this module on.  It restates the PUBLISHED algorithm of smplx/lbs.py
(`lbs`, `batch_rodrigues`, `batch_rigid_transform`, `blend_shapes`, `vertices2joints`) and smplx/body_models.py
(`SMPL.forward`, `VertexJointSelector`) from their documented behaviour; tests/test_lbs_oracle.py checks the analytic
gradients against torch autograd through a torch restatement of the same formulas, and the shapes / invariants (rest pose,
rigid root rotation, partition of unity).  Model parameters in the tests are synthetic (random, SMPL-shaped).

  verts, joints = lbs(theta)            smplx lbs(): shape blend, joint regression, Rodrigues, pose blend shapes, rigid
                                        transform chain, skinning;  joints = 24 posed joints + vertices[extra_joint_vertex]
  temp = mean_{t < T-1, v} || V[t, v] - V[t+1, v] ||            (motion_denoise.py:88-89),  weight 10 (1 + it)   (:31)
  data = mean_{t, j}       || Jtr[t, j] - Jtr0[t, j] ||          (:93-94, it > 0 only :92),  weight 100 / (1 + it) (:32)
"""
from __future__ import annotations

import numpy as np

from posendf_amd.synth import SMPL_EXTRA_JOINT_VERTICES, SMPL_PARENTS, SMPL_V, make_body_model  # noqa: E402,F401


def synthetic_model(*args, **kwargs):
    """random SMPL-shaped model parameters (posendf_amd.synth.make_body_model: shared with the benchmarks, like make_weights)"""
    return make_body_model(*args, **kwargs)


# ---------------------------------------------------------------- smplx/lbs.py, restated
def batch_rodrigues(rot_vecs, eps=1e-8):
    """smplx batch_rodrigues: angle = ||r + 1e-8||, axis = r / angle, R = I + sin K + (1 - cos) K K.  Returns R [n,3,3]."""
    dt = rot_vecs.dtype
    angle = np.sqrt(((rot_vecs + dt.type(eps)) ** 2).sum(-1, keepdims=True))
    n = rot_vecs / angle
    s, c = np.sin(angle)[..., None], np.cos(angle)[..., None]
    K = np.zeros(rot_vecs.shape[:-1] + (3, 3), dt)
    K[..., 0, 1], K[..., 0, 2] = -n[..., 2], n[..., 1]
    K[..., 1, 0], K[..., 1, 2] = n[..., 2], -n[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -n[..., 1], n[..., 0]
    return np.eye(3, dtype=dt) + s * K + (1 - c) * (K @ K)


def _rodrigues_vjp(r, gR, eps=1e-8):
    """reverse pass of batch_rodrigues: d <gR, R(r)> / d r, r [..,3], gR [..,3,3]."""
    dt = r.dtype
    a = r + dt.type(eps)
    th = np.sqrt((a * a).sum(-1, keepdims=True))
    n = r / th
    s, c = np.sin(th)[..., None], np.cos(th)[..., None]
    K = np.zeros(r.shape[:-1] + (3, 3), dt)
    K[..., 0, 1], K[..., 0, 2] = -n[..., 2], n[..., 1]
    K[..., 1, 0], K[..., 1, 2] = n[..., 2], -n[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -n[..., 1], n[..., 0]
    KK = K @ K
    g_th = (gR * (c * K + s * KK)).sum((-1, -2))[..., None]
    KT = np.swapaxes(K, -1, -2)
    gK = s * gR + (1 - c) * (gR @ KT + KT @ gR)
    gn = np.stack([gK[..., 2, 1] - gK[..., 1, 2], gK[..., 0, 2] - gK[..., 2, 0], gK[..., 1, 0] - gK[..., 0, 1]], -1)
    g_th = g_th - (gn * r).sum(-1, keepdims=True) / (th * th)          # n = r / th
    return gn / th + g_th * a / th


def rest_shape(model, dtype=np.float64):
    """v_shaped = v_template + blend_shapes(betas, shapedirs); J = J_regressor v_shaped  (smplx lbs(), first two steps).
    The betas are fixed during the optimisation (motion_denoise.py:27,67: zeros, requires_grad False)."""
    vt = np.asarray(model["v_template"], dtype)
    sd = np.asarray(model["shapedirs"], dtype)
    b = np.asarray(model["betas"], dtype)
    v_shaped = vt + sd @ b
    J = np.asarray(model["J_regressor"], dtype) @ v_shaped
    return v_shaped, J


def lbs(theta, model, dtype=np.float64, global_orient=None, keep=False):
    """SMPL.forward(betas, body_pose=theta, global_orient) of smplx (create_transl default: a zero translation).
    theta [N,69] axis-angle of the 23 body joints; the root orientation defaults to SMPL's zero-initialised
    `global_orient` parameter (the reference passes root_orient=None, body_model.py:35-40).
    Returns vertices [N,V,3] and joints [N, 24 + n_extra, 3]."""
    theta = np.asarray(theta, dtype).reshape(-1, 69)
    N = theta.shape[0]
    parents = np.asarray(model["parents"])
    nJ = len(parents)
    v_shaped, J = rest_shape(model, dtype)
    go = np.zeros((N, 3), dtype) if global_orient is None else np.asarray(global_orient, dtype).reshape(N, 3)
    full = np.concatenate([go, theta], 1).reshape(N, nJ, 3)
    R = batch_rodrigues(full.reshape(-1, 3)).reshape(N, nJ, 3, 3)
    pf = (R[:, 1:] - np.eye(3, dtype=dtype)).reshape(N, -1)                      # pose_feature [N, 207]
    v_posed = v_shaped[None] + (pf @ np.asarray(model["posedirs"], dtype)).reshape(N, -1, 3)
    rel = J.copy()
    rel[1:] -= J[parents[1:]]
    G_R = np.empty((N, nJ, 3, 3), dtype)
    G_t = np.empty((N, nJ, 3), dtype)
    G_R[:, 0], G_t[:, 0] = R[:, 0], rel[0]
    for i in range(1, nJ):                                                      # batch_rigid_transform: chain of 4x4s
        p = parents[i]
        G_R[:, i] = G_R[:, p] @ R[:, i]
        G_t[:, i] = (G_R[:, p] @ rel[i]) + G_t[:, p]
    A_t = G_t - (G_R @ J[None, :, :, None])[..., 0]                             # rel_transforms: remove the rest pose
    W = np.asarray(model["lbs_weights"], dtype)
    T_R = np.einsum("vj,njab->nvab", W, G_R)
    T_t = np.einsum("vj,nja->nva", W, A_t)
    verts = np.einsum("nvab,nvb->nva", T_R, v_posed) + T_t
    joints = np.concatenate([G_t, verts[:, np.asarray(model["extra_joint_vertex"])]], 1)
    if keep:
        return verts, joints, dict(R=R, full=full, v_posed=v_posed, G_R=G_R, G_t=G_t, T_R=T_R, rel=rel, J=J, W=W, pf=pf)
    return verts, joints


def lbs_vjp(model, cache, g_verts, g_joints, dtype=np.float64):
    """reverse pass of lbs(): d (<g_verts, verts> + <g_joints, joints>) / d theta -> [N,69]."""
    parents = np.asarray(model["parents"])
    nJ = len(parents)
    R, G_R, T_R, v_posed, rel, J, W = (cache[k] for k in ("R", "G_R", "T_R", "v_posed", "rel", "J", "W"))
    N = R.shape[0]
    gV = np.array(g_verts, dtype)
    ex = np.asarray(model["extra_joint_vertex"])
    np.add.at(gV, (slice(None), ex), np.asarray(g_joints, dtype)[:, nJ:])       # joints[:, 24:] = verts[:, extra]
    g_vposed = np.einsum("nvab,nva->nvb", T_R, gV)
    gA_R = np.einsum("vj,nva,nvb->njab", W, gV, v_posed)
    gA_t = np.einsum("vj,nva->nja", W, gV)
    g_pf = g_vposed.reshape(N, -1) @ np.asarray(model["posedirs"], dtype).T      # [N,207]
    gR = np.zeros((N, nJ, 3, 3), dtype)
    gR[:, 1:] += g_pf.reshape(N, nJ - 1, 3, 3)
    gG_R = gA_R - gA_t[..., :, None] * J[None, :, None, :]                      # A_t = G_t - G_R J
    gG_t = gA_t + np.asarray(g_joints, dtype)[:, :nJ]                           # joints[:, :24] = G_t
    for i in range(nJ - 1, 0, -1):
        p = parents[i]
        gR[:, i] += np.swapaxes(G_R[:, p], -1, -2) @ gG_R[:, i]
        gG_R[:, p] += gG_R[:, i] @ np.swapaxes(R[:, i], -1, -2) + gG_t[:, i, :, None] * rel[i][None, None, :]
        gG_t[:, p] += gG_t[:, i]
    g_full = _rodrigues_vjp(cache["full"].reshape(-1, 3), gR.reshape(-1, 3, 3)).reshape(N, nJ, 3)
    return g_full[:, 1:].reshape(N, 69)


# ---------------------------------------------------------------- experiments/motion_denoise.py:86-94, restated
def body_terms(theta, joints0, model, it, dtype=np.float64, coefs=None):
    """ONE sequence theta [T,69]: the unweighted temporal and data terms and the gradient of their WEIGHTED sum
    10 (1 + it) temp + [it > 0] 100 / (1 + it) data  with respect to theta (motion_denoise.py:29-45,86-94).
    `coefs` = (temp weight, data weight) replaces that schedule (experiments/partial_observation.py:31-32).  Like the
    reference there is no epsilon under the square roots: two identical consecutive vertices (or a joint that has not
    moved, which is why the reference skips the data term at it = 0, :92) give a NaN gradient."""
    theta = np.asarray(theta, dtype).reshape(-1, 69)
    T = theta.shape[0]
    verts, joints, cache = lbs(theta, model, dtype, keep=True)
    gV = np.zeros_like(verts)
    gJ = np.zeros_like(joints)
    terms = {}
    if T > 1:
        diff = verts[:-1] - verts[1:]
        nrm = np.sqrt((diff * diff).sum(-1, keepdims=True))
        terms["temp"] = nrm.mean(dtype=dtype)
        with np.errstate(divide="ignore", invalid="ignore"):
            u = diff / nrm * (dtype(10.0 * (1 + it) if coefs is None else coefs[0]) / dtype(nrm.size))
        gV[:-1] += u
        gV[1:] -= u
    if it > 0:
        diff = joints - np.asarray(joints0, dtype)
        nrm = np.sqrt((diff * diff).sum(-1, keepdims=True))
        terms["data"] = nrm.mean(dtype=dtype)
        with np.errstate(divide="ignore", invalid="ignore"):
            gJ += diff / nrm * (dtype(100.0 / (1 + it) if coefs is None else coefs[1]) / dtype(nrm.size))
    return lbs_vjp(model, cache, gV, gJ, dtype), terms
