"""Body-model oracle (oracle/lbs_np.py, parity UNPINNED: smplx is third-party and absent) and the host side of the HIP
linear-blend-skinning path: the oracle's analytic reverse pass against torch autograd through a torch restatement of the
published smplx formulas, its invariants, and the packed model ("blob") of posendf_amd/csrc/pndf_lbs.hip through a
lane-level numpy model of the kernel's four MFMA contractions (no GPU)."""
import ctypes

import numpy as np
import pytest
import torch

from lane_model import G, LANES, P, mfma_16x16x4
from oracle import lbs_np
from oracle.lbs_torch import torch_lbs


def _theta(T, seed=0, scale=0.3):
    rng = np.random.default_rng(seed)
    th = np.cumsum(rng.normal(size=(T, 69)) * 0.05, axis=0) + rng.normal(size=(1, 69)) * scale
    th[min(2, T - 1), 6:9] = 0.0                       # a zero rotation (Rodrigues' epsilon branch)
    return th


def test_oracle_forward_matches_torch_restatement_and_invariants():
    m = lbs_np.synthetic_model(V=300, seed=1)
    th = _theta(6)
    V, J = lbs_np.lbs(th, m)
    Vt, Jt = torch_lbs(torch.tensor(th), m)
    assert np.abs(V - Vt.numpy()).max() < 1e-12 and np.abs(J - Jt.numpy()).max() < 1e-12
    assert J.shape == (6, 24 + len(m["extra_joint_vertex"]), 3)
    v_shaped, Jr = lbs_np.rest_shape(m)
    V0, J0 = lbs_np.lbs(np.zeros((1, 69)), m)             # rest pose (up to Rodrigues' 1e-8)
    assert np.abs(V0[0] - v_shaped).max() < 1e-6 and np.abs(J0[0, :24] - Jr).max() < 1e-12
    assert np.allclose(J[:, 24:], V[:, m["extra_joint_vertex"]])
    assert abs(m["lbs_weights"].sum(1) - 1).max() < 1e-6     # partition of unity


@pytest.mark.parametrize("it", [0, 3])
def test_oracle_terms_gradient_matches_autograd(it):
    m = lbs_np.synthetic_model(V=200, seed=2)
    th = _theta(5, seed=3)
    _, J = lbs_np.lbs(th, m)
    J0 = J + np.random.default_rng(4).normal(size=J.shape) * 0.01
    g, terms = lbs_np.body_terms(th, J0, m, it)
    t = torch.tensor(th, requires_grad=True)
    Vt, Jt = torch_lbs(t, m)
    temp = torch.mean(torch.sqrt(torch.sum((Vt[:-1] - Vt[1:]) ** 2, dim=2)))           # motion_denoise.py:88-89
    loss = 10.0 * (1 + it) * temp
    if it > 0:
        data = torch.mean(torch.sqrt(torch.sum((Jt - torch.tensor(J0)) ** 2, dim=2)))     # :93-94
        loss = loss + 100.0 / (1 + it) * data
        assert abs(terms["data"] - data.item()) < 1e-12
    loss.backward()
    assert abs(terms["temp"] - temp.item()) < 1e-12
    assert np.abs(t.grad.numpy() - g).max() < 1e-10 * max(1.0, np.abs(g).max())
    assert np.abs(g[:, 63:]).max() > 0                     # the hand joints move vertices: they do get a gradient


def _pack(m):
    from posendf_amd.engine import load_library
    lib = load_library()
    V = m["v_template"].shape[0]
    blob = np.zeros(lib.pndf_lbs_packed_floats(V), np.float32)
    J, rel = np.zeros(72, np.float32), np.zeros(72, np.float32)
    arrs = [np.ascontiguousarray(m[k], dtype=np.float32) for k in ("v_template", "shapedirs", "betas", "posedirs", "J_regressor")]
    par = np.ascontiguousarray(m["parents"], dtype=np.int32)
    w = np.ascontiguousarray(m["lbs_weights"], dtype=np.float32)
    ex = np.ascontiguousarray(m["extra_joint_vertex"], dtype=np.int32)
    rc = lib.pndf_lbs_pack_host(V, arrs[2].size, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data,
                                arrs[3].ctypes.data, arrs[4].ctypes.data, par.ctypes.data, w.ctypes.data, ex.ctypes.data, len(ex),
                                blob.ctypes.data, J.ctypes.data, rel.ctypes.data)
    assert rc == 0
    return blob.reshape(-1, 10752), J.reshape(24, 3), rel.reshape(24, 3)


def test_packed_model_through_lane_model_of_the_vertex_kernel():
    """One wave = 16 frames; per 16-vertex group the kernel's four contractions on v_mfma_f32_16x16x4_f32, fed from the REAL
    packed blob with the kernel's LDS index arithmetic (pndf_lbs.hip `lbs_vertex_body`): vertices and the reverse
    contractions must equal the oracle's."""
    m = lbs_np.synthetic_model(V=41, seed=5, extra=(3, 17, 40))          # 3 groups, the last one padded
    blob, J, rel = _pack(m)
    v_shaped, Jr = lbs_np.rest_shape(m, np.float64)
    assert np.abs(J - Jr).max() < 1e-6 and np.allclose(rel[0], J[0]) and np.allclose(rel[5], J[5] - J[m["parents"][5]], atol=1e-7)
    th = _theta(16, seed=6)
    verts, _, c = lbs_np.lbs(th, m, keep=True)
    # what pndf_lbs_pose_kernel hands over: pose feature and A = [G_R | G_t - G_R J], per frame
    pf = np.zeros((16, 208), np.float32)
    pf[:, :207] = c["pf"]
    A = np.concatenate([c["G_R"].reshape(16, 24, 9), c["G_t"] - (c["G_R"] @ Jr[None, :, :, None])[..., 0]], -1).astype(np.float32)
    pfB = [pf[P, 4 * s + G] for s in range(52)]                         # lane (g, p): frame p, k = 4 s + g
    AB = [[A[P, np.minimum(4 * s + G, 23), e] for s in range(6)] for e in range(12)]
    rng = np.random.default_rng(7)
    gV_all = rng.normal(size=verts.shape).astype(np.float32)
    gpf = [np.zeros((64, 4), np.float32) for _ in range(13)]
    gA = [[np.zeros((64, 4), np.float32) for _ in range(2)] for _ in range(12)]
    V = m["v_template"].shape[0]
    for grp in range(blob.shape[0]):
        Bf = blob[grp]
        off = []
        for c3 in range(3):
            acc = np.zeros((64, 4), np.float32)
            for s in range(52):
                acc = mfma_16x16x4(Bf[c3 * 3328 + s * 64 + LANES], pfB[s], acc)
            off.append(acc)
        Tm = []
        for e in range(12):
            acc = np.zeros((64, 4), np.float32)
            for s in range(6):
                acc = mfma_16x16x4(Bf[9984 + s * 64 + LANES], AB[e][s], acc)
            Tm.append(acc)
        fl = Bf[10544:10560].view(np.int32)[4 * G[:, None] + np.arange(4)]
        vp = [Bf[10496 + c3 * 16 + 4 * G[:, None] + np.arange(4)] + off[c3] for c3 in range(3)]
        Vk = [Tm[3 * a] * vp[0] + Tm[3 * a + 1] * vp[1] + Tm[3 * a + 2] * vp[2] + Tm[9 + a] for a in range(3)]
        vid = grp * 16 + 4 * G[:, None] + np.arange(4)                    # [lane, r]
        ok = vid < V
        assert np.array_equal(fl == -2, ~ok)
        for e, v in enumerate(m["extra_joint_vertex"]):
            assert (fl[vid == v] == e).all()
        want = verts[P[:, None], np.minimum(vid, V - 1)]                    # [lane, r, 3]
        for a in range(3):
            assert np.abs(np.where(ok, Vk[a] - want[..., a], 0)).max() < 2e-5
        gV = [np.where(ok, gV_all[P[:, None], np.minimum(vid, V - 1), a], 0).astype(np.float32) for a in range(3)]
        gvp = [Tm[b] * gV[0] + Tm[3 + b] * gV[1] + Tm[6 + b] * gV[2] for b in range(3)]
        base = P * 16 + 4 * G
        for kt in range(13):
            for c3 in range(3):
                for r in range(4):
                    gpf[kt] = mfma_16x16x4(Bf[c3 * 3328 + kt * 256 + base + r], gvp[c3][:, r], gpf[kt])
        for e in range(12):
            X = gV[e // 3] * vp[e % 3] if e < 9 else gV[e - 9]
            for r in range(4):
                gA[e][0] = mfma_16x16x4(Bf[9984 + base + r], X[:, r], gA[e][0])
                gA[e][1] = mfma_16x16x4(Bf[9984 + 256 + base + r], X[:, r], gA[e][1])
    # the oracle's contractions
    g_vposed = np.einsum("nvab,nva->nvb", c["T_R"], gV_all.astype(np.float64))
    want_pf = g_vposed.reshape(16, -1) @ np.asarray(m["posedirs"], np.float64).T                     # [16, 207]
    W = np.asarray(m["lbs_weights"], np.float64)
    want_AR = np.einsum("vj,nva,nvb->njab", W, gV_all.astype(np.float64), c["v_posed"])
    want_At = np.einsum("vj,nva->nja", W, gV_all.astype(np.float64))
    for kt in range(13):
        for r in range(4):
            k = 16 * kt + 4 * G + r
            got = gpf[kt][:, r]
            ref = np.where(k < 207, want_pf[P, np.minimum(k, 206)], 0)
            assert np.abs(got - ref).max() < 1e-3 * max(1.0, np.abs(want_pf).max()), (kt, r)
    for e in range(12):
        for jt in range(2):
            for r in range(4):
                j = 16 * jt + 4 * G + r
                ref = want_AR[P, np.minimum(j, 23), e // 3, e % 3] if e < 9 else want_At[P, np.minimum(j, 23), e - 9]
                ref = np.where(j < 24, ref, 0)
                assert np.abs(gA[e][jt][:, r] - ref).max() < 1e-3 * max(1.0, np.abs(want_AR).max()), (e, jt, r)


def test_pack_refuses_bad_models():
    from posendf_amd.engine import load_library
    lib = load_library()
    m = lbs_np.synthetic_model(V=20, seed=8, extra=(1, 2))
    par = m["parents"].copy()
    par[3] = 7                                           # a parent after its child
    m2 = dict(m, parents=par)
    blob = np.zeros(lib.pndf_lbs_packed_floats(20), np.float32)
    arrs = [np.ascontiguousarray(m2[k], dtype=np.float32) for k in ("v_template", "shapedirs", "betas", "posedirs", "J_regressor")]
    w = np.ascontiguousarray(m2["lbs_weights"], dtype=np.float32)
    ex = np.ascontiguousarray([1, 25], dtype=np.int32)
    args = lambda p, e: (20, 10, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data,
                         arrs[4].ctypes.data, p.ctypes.data, w.ctypes.data, e.ctypes.data, len(e), blob.ctypes.data, None, None)
    assert lib.pndf_lbs_pack_host(*args(np.ascontiguousarray(par, dtype=np.int32), np.ascontiguousarray([1], dtype=np.int32))) == -4
    assert lib.pndf_lbs_pack_host(*args(np.ascontiguousarray(m["parents"], dtype=np.int32), ex)) == -1      # vertex 25 >= V
    # a vertex named twice: refused (the vertex kernels serve one picked joint per vertex)
    good = np.ascontiguousarray(m["parents"], dtype=np.int32)
    assert lib.pndf_lbs_pack_host(*args(good, np.ascontiguousarray([1, 2], dtype=np.int32))) == 0
    assert lib.pndf_lbs_pack_host(*args(good, np.ascontiguousarray([2, 2], dtype=np.int32))) == -1


def test_synthetic_models_never_pick_a_vertex_twice():
    """the default SMPL table folded into a small cloud (`% V`) used to produce duplicates silently"""
    from posendf_amd import synth
    for V in (21, 30, 137, 500, 3000, synth.SMPL_V):
        ex = synth.make_body_model(V=V, seed=1)["extra_joint_vertex"]
        assert len(ex) == 21 and len(set(ex.tolist())) == 21 and ex.min() >= 0 and ex.max() < V
    assert np.array_equal(synth.make_body_model(seed=1)["extra_joint_vertex"], np.asarray(synth.SMPL_EXTRA_JOINT_VERTICES))
    with pytest.raises(ValueError):
        synth.make_body_model(V=20, seed=1)
