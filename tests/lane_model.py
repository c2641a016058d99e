"""Lane-level numpy model of ONE wave of the fused kernel (posendf_amd/csrc/pndf_kernel.hip).

It consumes the REAL packed weight stream produced by the library's host packer (pndf_pack_host) in the
order the kernel consumes it and applies the documented semantics of v_mfma_f32_16x16x4_f32
(cdna_hip_programming.md section 3: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
D[row = 4 (lane >> 4) + reg][col = lane & 15]).  It therefore checks, without a GPU, the three things a
first GPU run would otherwise have to debug at once: the tile permutation, the phase/chunk order of the
stream, and the claim that the D layout of one layer is the B layout of the next.

Test infrastructure only.
"""
import numpy as np

LANES = np.arange(64)
G = LANES >> 4
P = LANES & 15

# mirrors pndf_layout.h PHASES: KA, CT, NC, NB
PHASES = [(8, 2, 8, 32), (32, 2, 32, 32), (32, 4, 4, 4), (4, 4, 4, 32), (32, 2, 32, 32), (32, 2, 8, 8)]
BIAS_OFF = [0, 256, 768, 1792, 2304, 2560, 2688]
W6_OFF = 2624


def mfma_16x16x4(a, b, c):
    """a, b: [64] one VGPR each; c: [64,4].  Returns D = A.B + C in the C/D register layout."""
    A = np.zeros((16, 4), np.float32)
    B = np.zeros((4, 16), np.float32)
    A[P, G] = a
    B[G, P] = b
    D = (A.astype(np.float64) @ B.astype(np.float64)).astype(np.float32)   # [row, col]
    out = c.copy()
    for r in range(4):
        out[:, r] += D[4 * G + r, P]
    return out


ENC_PAD = 48          # encoder tiles per direction, padded to 3 slots (pndf_layout.h)
ENCB_OFF = 2692
PARENT = (-1, -1, -1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19)


class Stream:
    def __init__(self, stream, pos=0):
        self.t = stream.reshape(-1, 64, 4)
        self.pos = pos

    def tile(self):
        t = self.t[self.pos]
        self.pos += 1
        return t


def run_phase(ph, xin, acc, st, bias_a, masks, slope, bwd):
    KA, CT, NC, NB = PHASES[ph]
    for c in range(NC):
        ch = [np.zeros((64, 4), np.float32) for _ in range(CT)]
        if not bwd:
            for ci in range(CT):
                for r in range(4):
                    ch[ci][:, r] = bias_a[16 * (c * CT + ci) + 4 * G + r]
        for kt in range(KA):
            a = [st.tile() for _ in range(CT)]
            for s in range(4):
                for ci in range(CT):
                    ch[ci] = mfma_16x16x4(a[ci][:, s], xin[kt][:, s], ch[ci])
        if not bwd:
            m = []
            for ci in range(CT):
                pos = ch[ci] > 0
                m.append(pos)
                ch[ci] = np.where(pos, ch[ci], ch[ci] * np.float32(slope))
            masks[(ph, c)] = m
        else:
            m = masks[({3: 2, 4: 1, 5: 0}[ph], c)]
            for ci in range(CT):
                ch[ci] = np.where(m[ci], ch[ci], ch[ci] * np.float32(slope))
        for nbp in range(NB // 2):
            a = [[st.tile() for _ in range(2)] for _ in range(CT)]
            for ci in range(CT):
                for s in range(4):
                    for h in range(2):
                        acc[2 * nbp + h] = mfma_16x16x4(a[ci][h][:, s], ch[ci][:, s], acc[2 * nbp + h])


def load_bias(bias, off, nt):
    out = []
    for t in range(nt):
        x = np.zeros((64, 4), np.float32)
        for r in range(4):
            x[:, r] = bias[off + 16 * t + 4 * G + r]
        out.append(x)
    return out


def act_tiles(x, slope):
    m = [t > 0 for t in x]
    return [np.where(mm, t, t * np.float32(slope)) for t, mm in zip(x, m)], m


def dact_tiles(g, m, slope):
    return [np.where(mm, t, t * np.float32(slope)) for t, mm in zip(g, m)]


def decode(tiles):
    """C/D-layout tiles -> [16 poses, 16*len(tiles)] matrix."""
    out = np.zeros((16, 16 * len(tiles)), np.float32)
    for t, x in enumerate(tiles):
        for r in range(4):
            out[P, 16 * t + 4 * G + r] = x[:, r]
    return out


def _tile_bias(bias, off):
    x = np.zeros((64, 4), np.float32)
    for r in range(4):
        x[:, r] = bias[off + 4 * G + r]
    return x


def _mm(tile, x, c):
    for s in range(4):
        c = mfma_16x16x4(tile[:, s], x[:, s], c)
    return c


def encoder_fwd_wave(q16, stream, bias, slope):
    """q16 [16,21,4] -> (features [16,126], per-joint sign masks) using the encoder tiles at the head of the
    stream (posendf_amd/csrc/pndf_kernel.hip: enc_fwd_joint)."""
    q16 = q16.astype(np.float32)
    denom = np.maximum(np.sqrt((q16 * q16).sum(1)), np.float32(1e-12))   # [16,4]
    st = Stream(stream, 0)
    F, masks = [None] * 21, [None] * 21
    feats = np.zeros((16, 126), np.float32)
    for j in range(21):
        t1, t2 = st.tile(), st.tile()
        nq = (q16[:, j, :] / denom)[P]                      # every lane: its pose's normalised quaternion
        par = F[PARENT[j]] if PARENT[j] >= 0 else np.zeros((64, 4), np.float32)
        X = np.where((G == 0)[:, None], nq, par)
        H = _mm(t1, X, _tile_bias(bias, ENCB_OFF + 32 * j))
        mh = H > 0
        H = np.where(mh, H, H * np.float32(slope))
        Fj = _mm(t2, H, _tile_bias(bias, ENCB_OFF + 32 * j + 16))
        mf = Fj > 0
        Fj = np.where(mf, Fj, Fj * np.float32(slope))
        F[j], masks[j] = Fj, (mh, mf)
        for lane in range(64):
            g, p = lane >> 4, lane & 15
            if g == 1:
                feats[p, 6 * j:6 * j + 4] = Fj[lane]
            elif g == 2:
                feats[p, 6 * j + 4:6 * j + 6] = Fj[lane, :2]
    assert st.pos == 42
    return feats, masks


def encoder_bwd_wave(gx0, masks, stream, slope):
    """gx0 [16,128] (d d / d feature) -> d d / d n [16,84] using the encoder tiles at the tail of the stream."""
    ntiles = stream.size // 256
    st = Stream(stream, ntiles - ENC_PAD)
    GF = []
    for j in range(21):
        x = np.zeros((64, 4), np.float32)
        for lane in range(64):
            g, p = lane >> 4, lane & 15
            if g == 1:
                x[lane] = gx0[p, 6 * j:6 * j + 4]
            elif g == 2:
                x[lane, :2] = gx0[p, 6 * j + 4:6 * j + 6]
        GF.append(x)
    gn = np.zeros((16, 84), np.float32)
    for j in range(20, -1, -1):
        t1, t2 = st.tile(), st.tile()
        mh, mf = masks[j]
        gz2 = np.where(mf, GF[j], GF[j] * np.float32(slope))
        GH = _mm(t1, gz2, np.zeros((64, 4), np.float32))
        gz1 = np.where(mh, GH, GH * np.float32(slope))
        GI = _mm(t2, gz1, np.zeros((64, 4), np.float32))
        for lane in range(16):                             # lane group 0: rows 0..3 = d d / d n_j
            gn[lane, 4 * j:4 * j + 4] = GI[lane]
        if PARENT[j] >= 0:
            GF[PARENT[j]] = GF[PARENT[j]] + GI
    assert st.pos == ntiles - ENC_PAD + 42
    return gn


def trunk_wave(feat16, stream, bias, slope):
    """feat16: [16,126] encoder features of the wave's 16 poses.  Returns (d[16], gx0[16,128], stages)."""
    f = np.zeros((16, 128), np.float32)
    f[:, :126] = feat16
    x0 = []
    for kt in range(8):
        x = np.zeros((64, 4), np.float32)
        for s in range(4):
            x[:, s] = f[P, 16 * kt + 4 * G + s]
        x0.append(x)
    st, masks, stages = Stream(stream, ENC_PAD), {}, {}
    x2 = load_bias(bias, BIAS_OFF[1], 32)
    run_phase(0, x0, x2, st, bias[BIAS_OFF[0]:], masks, slope, False)
    x2, m2 = act_tiles(x2, slope)
    stages["x2"] = decode(x2)
    x4 = load_bias(bias, BIAS_OFF[3], 32)
    run_phase(1, x2, x4, st, bias[BIAS_OFF[2]:], masks, slope, False)
    x4, m4 = act_tiles(x4, slope)
    stages["x4"] = decode(x4)
    x6 = load_bias(bias, BIAS_OFF[5], 4)
    run_phase(2, x4, x6, st, bias[BIAS_OFF[4]:], masks, slope, False)
    x6, m6 = act_tiles(x6, slope)
    stages["x6"] = decode(x6)
    w6 = load_bias(bias, W6_OFF, 4)
    part = np.zeros(64, np.float32)
    for t in range(4):
        for r in range(4):
            part += w6[t][:, r] * x6[t][:, r]
    tot = np.zeros(64, np.float32)
    for l in range(64):
        tot[l] = part[[(l & 15) + 16 * g for g in range(4)]].sum()
    z7 = tot + bias[BIAS_OFF[6]]
    d = np.maximum(z7, 0)
    gz7 = (z7 > 0).astype(np.float32)
    g6 = dact_tiles([w6[t] * gz7[:, None] for t in range(4)], m6, slope)
    g4 = [np.zeros((64, 4), np.float32) for _ in range(32)]
    run_phase(3, g6, g4, st, None, masks, slope, True)
    g4 = dact_tiles(g4, m4, slope)
    stages["g4"] = decode(g4)
    g2 = [np.zeros((64, 4), np.float32) for _ in range(32)]
    run_phase(4, g4, g2, st, None, masks, slope, True)
    g2 = dact_tiles(g2, m2, slope)
    stages["g2"] = decode(g2)
    g0 = [np.zeros((64, 4), np.float32) for _ in range(8)]
    run_phase(5, g2, g0, st, None, masks, slope, True)
    assert st.pos == st.t.shape[0] - ENC_PAD, (st.pos, st.t.shape)
    return d[:16], decode(g0), stages


# ====================================================================== split-precision trunk (f16 x 3)
def mfma_f16_16x16x32(a, b, c):
    """a, b: [64, 8] fp16 (A[i = lane & 15][k = 8 (lane >> 4) + jj], B[k][j = lane & 15]); c: [64,4] fp32.
    Products of fp16 values are exact in fp32; the sum is modelled in fp64 and rounded once per MFMA."""
    A = np.zeros((16, 32), np.float64)
    B = np.zeros((32, 16), np.float64)
    for jj in range(8):
        A[P, 8 * G + jj] = a[:, jj].astype(np.float64)
        B[8 * G + jj, P] = b[:, jj].astype(np.float64)
    D = A @ B
    out = c.astype(np.float64)
    for r in range(4):
        out[:, r] += D[4 * G + r, P]
    return out.astype(np.float32)


def split16(x):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def pack_blocks(tiles):
    """fp32 C/D-layout tiles -> list of (hi, lo) [64,8] fp16 B operands, one per pair of tiles."""
    out = []
    for b in range(len(tiles) // 2):
        x = np.concatenate([tiles[2 * b], tiles[2 * b + 1]], axis=1)      # [64, 8]
        out.append(split16(x))
    return out


class PairStream:
    def __init__(self, stream_f32, tile_pos):
        self.t = stream_f32.reshape(-1, 256)
        self.pos = tile_pos

    def pair(self):
        hi = self.t[self.pos].view(np.float16).reshape(64, 8)
        lo = self.t[self.pos + 1].view(np.float16).reshape(64, 8)
        self.pos += 2
        return hi, lo


def _mm3(w, x, c):
    (wh, wl), (xh, xl) = w, x
    c = mfma_f16_16x16x32(wh, xh, c)
    c = mfma_f16_16x16x32(wh, xl, c)
    c = mfma_f16_16x16x32(wl, xh, c)
    return c


def run_phase_split(ph, xin, acc, st, bias_a, masks, slope, bwd, to_true, oscale, bscale):
    """xin: list of (hi, lo) blocks; acc: list of fp32 tiles.  Stream order A(0) | A(c+1) B(c) ... (pndf_layout.h).
    to_true / oscale / bscale: the per-lane factors of SAct (pndf_kernel_split.hip "operand scaling"), [64] each."""
    KA, CT, NC, NB = PHASES[ph]
    cf = (to_true * oscale).astype(np.float32)[:, None]

    def part_a(c):
        ch = [np.zeros((64, 4), np.float32) for _ in range(CT)]
        if not bwd:
            for ci in range(CT):
                ch[ci] = _tile_bias(bias_a, 16 * (c * CT + ci)) * bscale.astype(np.float32)[:, None]
        for kb in range(KA // 2):
            for ci in range(CT):
                ch[ci] = _mm3(st.pair(), xin[kb], ch[ci])
        ch = [t * cf for t in ch]                               # accumulator -> scaled operand (LeakyReLU commutes)
        if not bwd:
            m = [t > 0 for t in ch]
            masks[(ph, c)] = m
        else:
            m = masks[({3: 2, 4: 1, 5: 0}[ph], c)]
        ch = [np.where(mm, t, t * np.float32(slope)) for t, mm in zip(ch, m)]
        return pack_blocks(ch)

    chb = part_a(0)
    for c in range(NC):
        nxt = part_a(c + 1) if c + 1 < NC else None
        for nb in range(NB):
            for b in range(CT // 2):
                acc[nb] = _mm3(st.pair(), chb[b], acc[nb])
        chb = nxt


# Per-pose power-of-two operand scaling of the split kernel (pndf_kernel_split.hip): weights as s_l W (1 / s_l in the
# bias block at SCALE_OFF + l), every operand tensor as sigma(p) x with sigma chosen from a measured or a-priori bound
SCALE_OFF = ENCB_OFF + 21 * 32
NORM_OFF = SCALE_OFF + 8


def pose_scale(bound):
    e = (np.asarray(bound, np.float32).view(np.uint32) >> 23) & 0xff
    e = np.clip(e, 100, 180).astype(np.uint32)
    return ((np.uint32(267) - e) << np.uint32(23)).view(np.float32)


def pose_max(tiles):
    """max |value| per lane, then over the four lane groups of the pose -> [64]"""
    m = np.max([np.abs(t).max(axis=1) for t in tiles], axis=0)
    return np.array([m[[(l & 15) + 16 * g for g in range(4)]].max() for l in range(64)], np.float32)


def trunk_wave_split(feat16, stream, bias, slope):
    inv = [np.float32(bias[SCALE_OFF + l]) for l in range(6)]
    nrm = [np.float32(bias[NORM_OFF + i]) for i in range(9)]
    assert all(v > 0 and np.log2(v) == np.round(np.log2(v)) for v in inv), inv       # powers of two
    f = np.zeros((16, 128), np.float32)
    f[:, :126] = feat16
    x0 = []
    for kt in range(8):
        x = np.zeros((64, 4), np.float32)
        for s in range(4):
            x[:, s] = f[P, 16 * kt + 4 * G + s]
        x0.append(x)
    st, masks, stages = PairStream(stream, ENC_PAD), {}, {}
    one = np.ones(64, np.float32)

    def fwd_pair(ph, la, lb, xin, bnd, sg_in, nt_out, nrm_w, nrm_b):
        sg_ch = pose_scale(nrm_w * bnd + nrm_b)
        acc = [t * (sg_ch / inv[lb])[:, None] for t in load_bias(bias, BIAS_OFF[lb], nt_out)]
        to_true = inv[la] / sg_in
        run_phase_split(ph, pack_blocks([t * sg_in[:, None] for t in xin]), acc, st, bias[BIAS_OFF[la]:], masks, slope, False,
                        to_true, sg_ch, one / to_true)
        tt = inv[lb] / sg_ch
        z = [t * tt[:, None] for t in acc]
        return z, pose_max(acc) * tt

    bnd = pose_max(x0)
    z2, bnd2 = fwd_pair(0, 0, 1, x0, bnd, pose_scale(bnd), 32, nrm[0], nrm[3])
    x2, m2 = act_tiles(z2, slope)
    stages["x2"] = decode(x2)
    z4, bnd4 = fwd_pair(1, 2, 3, x2, bnd2, pose_scale(bnd2), 32, nrm[1], nrm[4])
    x4, m4 = act_tiles(z4, slope)
    stages["x4"] = decode(x4)
    z6, _ = fwd_pair(2, 4, 5, x4, bnd4, pose_scale(bnd4), 4, nrm[2], nrm[5])
    x6, m6 = act_tiles(z6, slope)
    stages["x6"] = decode(x6)
    w6 = load_bias(bias, W6_OFF, 4)
    part = np.zeros(64, np.float32)
    for t in range(4):
        for r in range(4):
            part += w6[t][:, r] * x6[t][:, r]
    tot = np.zeros(64, np.float32)
    for l in range(64):
        tot[l] = part[[(l & 15) + 16 * g for g in range(4)]].sum()
    z7 = tot + bias[BIAS_OFF[6]]
    d = np.maximum(z7, 0)
    gz7 = (z7 > 0).astype(np.float32)

    def bwd_pair(ph, la, lb, gin, bnd, sg_in, nt_out, nrm_w):
        sg_ch = pose_scale(nrm_w * bnd)
        acc = [np.zeros((64, 4), np.float32) for _ in range(nt_out)]
        run_phase_split(ph, pack_blocks([t * sg_in[:, None] for t in gin]), acc, st, None, masks, slope, True,
                        inv[la] / sg_in, sg_ch, one)
        tt = inv[lb] / sg_ch
        return [t * tt[:, None] for t in acc], pose_max(acc) * tt

    # unit seed; the output derivative multiplies the result
    g6 = dact_tiles([w6[t].copy() for t in range(4)], m6, slope)
    bg = pose_max(w6)
    g4, bg4 = bwd_pair(3, 5, 4, g6, bg, pose_scale(bg), 32, nrm[6])
    g4 = dact_tiles(g4, m4, slope)
    g2, bg2 = bwd_pair(4, 3, 2, g4, bg4, pose_scale(bg4), 32, nrm[7])
    g2 = dact_tiles(g2, m2, slope)
    g0, _ = bwd_pair(5, 1, 0, g2, bg2, pose_scale(bg2), 8, nrm[8])
    assert st.pos == st.t.shape[0] - ENC_PAD, (st.pos, st.t.shape)
    g0 = [t * gz7[:, None] for t in g0]
    return d[:16], decode(g0), stages
