"""Lane-level numpy model of the split-precision body-model kernels' data layout (posendf_amd/csrc/pndf_lbs.hip
`lbs_vertex_split_body`, csrc/pndf_lbs_split.h), fed from the REAL packed model (`pndf_lbs_pack_host` ->
`pndf_lbs_pack_split_host`, host only): the swizzled LDS tiles, the transposed reads (`ds_read_b64_tr_b16`, lane mapping as
probed on the hardware: profiles/r02/tr_b16_probe.txt), the row reads, the operand order the pose kernel writes, the
v_mfma_f32_16x16x32_f16 lane layout, hi / lo halves with three products per block, and the half-k-block trick of the reverse
contractions (hi hi + hi lo in ONE MFMA).  Everything the four contractions of a (16 vertex x 16 frame) tile produce must
equal plain numpy contractions of the ORIGINAL model arrays."""
import ctypes

import numpy as np

from oracle import lbs_np

KP, PLANE, SB_PL, SB_WH, SB_WL, SB_VS, SB_FL, SBB = 224, 7168, 21504, 43008, 44032, 45056, 45248, 46080
LANE = np.arange(64)
G, P = LANE // 16, LANE % 16


def unit(r, q):                      # csrc/pndf_lbs_split.h pndf_lbs_sb_unit
    return 32 * (r // 8) + 4 * (r % 8) + (q ^ (2 * (r // 8)))


LANE_TR = 8 * unit(4 * G + P // 4, P % 4)
LANE_RV = 8 * unit(P, G)


def halfs(buf, byte_off, n=4):
    """n fp16 values per lane at per-lane byte offsets -> float64 [64, n]"""
    idx = (np.asarray(byte_off)[:, None] + 2 * np.arange(n)[None, :])
    raw = buf[idx].astype(np.uint16) | (buf[idx + 1].astype(np.uint16) << 8)
    return raw.view(np.float16).astype(np.float64)


def tr_read(buf, base):
    """ds_read_b64_tr_b16 with every lane addressing `base + LANE_TR`: lane i (of a 16-lane row) element j <- element i % 4
    of the 8 bytes addressed by lane 4 j + i / 4 of the same row"""
    own = halfs(buf, base + LANE_TR)                       # what each lane's 8 bytes hold
    out = np.empty((64, 4))
    for lane in range(64):
        row, i = lane // 16, lane % 16
        for j in range(4):
            out[lane, j] = own[16 * row + 4 * j + i // 4, i % 4]
    return out


def mfma(A, B):
    """v_mfma_f32_16x16x32_f16: A [64, 8]: lane l -> row l % 16, k = 8 (l / 16) + i; B [64, 8]: lane l -> column l % 16, same k;
    D [64, 4]: lane l -> rows 4 (l / 16) + r, column l % 16"""
    Am, Bm = np.zeros((16, 32)), np.zeros((32, 16))
    for lane in range(64):
        Am[lane % 16, 8 * (lane // 16):8 * (lane // 16) + 8] = A[lane]
        Bm[8 * (lane // 16):8 * (lane // 16) + 8, lane % 16] = B[lane]
    Dm = Am @ Bm
    return np.stack([Dm[4 * (LANE // 16) + r, LANE % 16] for r in range(4)], 1)


def split(x):
    hi = np.asarray(x, np.float64).astype(np.float16).astype(np.float64)
    lo = (np.asarray(x, np.float64) - hi).astype(np.float16).astype(np.float64)
    return hi, lo


def _pack_split(m):
    from posendf_amd.engine import load_library
    from test_lbs_oracle import _pack
    lib = load_library()
    blob, J, rel = _pack(m)
    V = m["v_template"].shape[0]
    sb = np.zeros(lib.pndf_lbs_packed_split_bytes(V), np.uint8)
    sc = np.zeros(2, np.float32)
    assert sb.size == blob.shape[0] * SBB
    assert lib.pndf_lbs_pack_split_host(V, blob.ctypes.data, sb.ctypes.data, sc.ctypes.data) == 0
    assert lib.pndf_lbs_pack_split_host(0, blob.ctypes.data, sb.ctypes.data, None) == -1
    return blob, sb.reshape(-1, SBB), float(sc[0]), float(sc[1])


def test_split_tiles_are_a_bijection_free_of_bank_conflicts():
    units = sorted(unit(r, q) for r in range(16) for q in range(4))
    assert units == list(range(64))
    # ds_read_b64 / ds_read_b64_tr_b16: two 32-lane groups over 64 banks = 32 slots of 8 bytes
    for half in (slice(0, 32), slice(32, 64)):
        assert len(set((LANE_RV[half] // 8) % 32)) == 32 and len(set((LANE_TR[half] // 8) % 32)) == 32


def test_packed_split_model_through_lane_model_of_the_vertex_kernel():
    m = lbs_np.synthetic_model(V=41, seed=5, extra=(3, 17, 40))          # 3 groups, the last one padded
    blob, sb, p_scale, w_scale = _pack_split(m)
    assert np.log2(p_scale) % 1 == 0 and np.log2(w_scale) % 1 == 0       # powers of two
    V = m["v_template"].shape[0]
    posedirs = np.asarray(m["posedirs"], np.float64).reshape(207, V, 3)
    W = np.asarray(m["lbs_weights"], np.float64)
    assert 2.0 ** 12 <= np.abs(posedirs).max() * p_scale < 2.0 ** 13 and 2.0 ** 12 <= W.max() * w_scale < 2.0 ** 13
    rng = np.random.default_rng(3)
    pf = np.zeros((16, KP))
    pf[:, :207] = rng.uniform(-2, 2, size=(16, 207))                     # |R - I| <= 2
    A = rng.normal(size=(16, 24, 12))                                    # joint transforms per frame
    PF_SCALE, A_SCALE = 4096.0, 1024.0
    # B operands as pndf_lbs_pose_split_kernel writes them: element i of a k-block in lane group g = index 16 (i / 4) + 4 g + i % 4
    i8 = np.arange(8)
    kperm = 16 * (i8 // 4)[None, :] + 4 * G[:, None] + (i8 % 4)[None, :]            # [64, 8]
    pfh, pfl = zip(*[split(pf[P[:, None], 32 * kb + kperm] * PF_SCALE) for kb in range(7)])
    Apad = np.zeros((16, 32, 12))
    Apad[:, :24] = A
    Ah, Al = zip(*[split(Apad[P[:, None], kperm, e] * A_SCALE) for e in range(12)])
    gV = rng.normal(size=(16, V + 16, 3)) * 1e-3                         # d L / d verts per frame (padding vertices unused)
    G_SCALE, X_SCALE = 2.0 ** 20, 2.0 ** 19
    gpf = [np.zeros((64, 4)) for _ in range(13)]
    gA = [[np.zeros((64, 4)) for _ in range(2)] for _ in range(12)]
    want_gpf, want_gA = np.zeros((16, 207)), np.zeros((16, 24, 12))
    for grp in range(sb.shape[0]):
        buf = sb[grp]
        vsl = np.arange(16 * grp, min(16 * grp + 16, V))
        # ---- forward: pose-blend offsets, three products per (k-block, component)
        off = []
        for c3 in range(3):
            acc = np.zeros((64, 4))
            for kb in range(7):
                q = c3 * PLANE + kb * 1024
                Ph = np.concatenate([tr_read(buf, q), tr_read(buf, q + 512)], 1)
                Pl = np.concatenate([tr_read(buf, SB_PL + q), tr_read(buf, SB_PL + q + 512)], 1)
                acc += mfma(Ph, pfh[kb]) + mfma(Ph, pfl[kb]) + mfma(Pl, pfh[kb])
            off.append(acc / (p_scale * PF_SCALE))
        want = np.einsum("tk,kvc->tvc", pf[:, :207], posedirs[:, vsl])                 # [frame, v, comp]
        for c3 in range(3):
            for r in range(4):
                ok = 4 * G + r < len(vsl)
                got = off[c3][:, r]
                assert np.abs(got[ok] - want[P[ok], (4 * G + r)[ok], c3]).max() < 1e-5 * np.abs(want).max()
                assert np.all(got[~ok] == 0)                                           # padded vertices: zero rows
        # ---- forward: skinning transforms T[e] = sum_j W[v, j] A[j][e]
        Wh = np.concatenate([tr_read(buf, SB_WH), tr_read(buf, SB_WH + 512)], 1)
        Wl = np.concatenate([tr_read(buf, SB_WL), tr_read(buf, SB_WL + 512)], 1)
        for e in range(12):
            Tm = (mfma(Wh, Ah[e]) + mfma(Wh, Al[e]) + mfma(Wl, Ah[e])) / (w_scale * A_SCALE)
            wantT = np.einsum("vj,tj->tv", W[vsl], A[:, :, e])
            for r in range(4):
                ok = 4 * G + r < len(vsl)
                assert np.abs(Tm[ok, r] - wantT[P[ok], (4 * G + r)[ok]]).max() < 1e-5 * np.abs(wantT).max()
        # ---- reverse operands from the D layout: lane (g, p) holds vertices 4 g + r of frame p
        vp = [rng.normal(size=(64, 4)) for _ in range(3)]                              # stand-in for v_posed of the tile
        gvp = [np.stack([gV[P, 16 * grp + 4 * G + r, c3] for r in range(4)], 1) * G_SCALE for c3 in range(3)]
        (g0h, g0l), (g1h, g1l), (g2h, g2l) = split(gvp[0]), split(gvp[1]), split(gvp[2])
        B0h, B0l, B1 = np.concatenate([g0h, g1h], 1), np.concatenate([g0l, g1l], 1), np.concatenate([g2h, g2l], 1)
        zero4 = np.zeros((64, 4))
        for kt in range(13):
            rd = lambda base: halfs(buf, base + kt * 512 + LANE_RV)
            c0h, c1h, c2h = rd(0), rd(PLANE), rd(2 * PLANE)
            c0l, c1l, c2l = rd(SB_PL), rd(SB_PL + PLANE), rd(SB_PL + 2 * PLANE)
            A0h = np.concatenate([c0h, c1h], 1)
            gpf[kt] += (mfma(A0h, B0h) + mfma(A0h, B0l) + mfma(np.concatenate([c0l, c1l], 1), B0h)
                        + mfma(np.concatenate([c2h, c2h], 1), B1) + mfma(np.concatenate([c2l, zero4], 1), B1))
        want_gpf += np.einsum("kvc,tvc->tk", posedirs[:, vsl], gV[:, vsl])
        gVs = [np.stack([gV[P, 16 * grp + 4 * G + r, a3] for r in range(4)], 1) * X_SCALE for a3 in range(3)]
        for e in range(12):
            X = gVs[e // 3] * vp[e % 3] if e < 9 else gVs[e - 9]
            xh, xl = split(X)
            Xe = np.concatenate([xh, xl], 1)
            for jt in range(2):
                wh, wl = halfs(buf, SB_WH + jt * 512 + LANE_RV), halfs(buf, SB_WL + jt * 512 + LANE_RV)
                gA[e][jt] += mfma(np.concatenate([wh, wh], 1), Xe) + mfma(np.concatenate([wl, zero4], 1), Xe)
            # the same in plain numpy: sum_v W[v, j] X[v] per frame
            Xfull = np.zeros((16, 16))
            for lane in range(64):
                Xfull[P[lane], 4 * G[lane]:4 * G[lane] + 4] = X[lane] / X_SCALE
            want_gA[:, :, e] += np.einsum("vj,tv->tj", W[vsl], Xfull[:, :len(vsl)])
        # shaped template and flags travel unchanged
        assert np.array_equal(buf[SB_VS:SB_VS + 192].view(np.float32), blob[grp][10496:10544])
        assert np.array_equal(buf[SB_FL:SB_FL + 64].view(np.int32), blob[grp][10544:10560].view(np.int32))
    for kt in range(13):
        for r in range(4):
            k = 16 * kt + 4 * G + r
            ok = k < 207
            got = gpf[kt][:, r] / (p_scale * G_SCALE)
            assert np.abs(got[ok] - want_gpf[P[ok], k[ok]]).max() < 1e-5 * np.abs(want_gpf).max()
            assert np.all(got[~ok] == 0)
    for e in range(12):
        for jt in range(2):
            for r in range(4):
                j = 16 * jt + 4 * G + r
                ok = j < 24
                got = gA[e][jt][:, r] / (w_scale * X_SCALE)
                assert np.abs(got[ok] - want_gA[P[ok], j[ok], e]).max() < 1e-5 * np.abs(want_gA).max()
                assert np.all(got[~ok] == 0)


def test_split_packer_scales_for_extreme_magnitudes_and_null_handle_calls():
    """The operand scales are powers of two that bring the largest |entry| into [2^12, 2^13) whatever the model's magnitudes
    (pose-corrective dirs of 1e-9 or 1e+3, weights of 1e-6), an all-zero model does not turn into infinities, and the
    precision calls refuse a null handle (no device needed)."""
    from posendf_amd.engine import load_library
    from test_lbs_oracle import _pack
    lib = load_library()
    for pscale, wscale in ((1e-9, 1.0), (1e3, 1e-6), (0.0, 0.0)):
        m = dict(lbs_np.synthetic_model(V=20, seed=3, extra=(1,)))
        m["posedirs"] = np.asarray(m["posedirs"], np.float32) * np.float32(pscale)
        m["lbs_weights"] = np.asarray(m["lbs_weights"], np.float32) * np.float32(wscale)
        blob, _, _ = _pack(m)
        sb = np.zeros(lib.pndf_lbs_packed_split_bytes(20), np.uint8)
        sc = np.zeros(2, np.float32)
        assert lib.pndf_lbs_pack_split_host(20, blob.ctypes.data, sb.ctypes.data, sc.ctypes.data) == 0
        assert np.isfinite(sc).all() and (sc > 0).all() and np.all(np.log2(sc.astype(np.float64)) % 1 == 0)
        halfs16 = sb.reshape(-1, SBB)[:, :SB_VS].reshape(-1).view(np.float16)
        assert np.isfinite(halfs16.astype(np.float32)).all() and np.abs(halfs16.astype(np.float32)).max() < 2.0 ** 13
        if pscale > 0:
            pmax = np.abs(m["posedirs"]).max() * float(sc[0])
            assert 2.0 ** 12 <= pmax < 2.0 ** 13
    assert lib.pndf_lbs_set_precision(None, 1) == -1 and lib.pndf_lbs_precision(None) == -1
    assert lib.pndf_lbs_packed_split_bytes(0) == 0 and lib.pndf_lbs_packed_split_bytes(17) == 2 * SBB
