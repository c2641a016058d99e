"""Lane-level numpy model of the NEXT trunk layout ("pair-32"): a PAIR of waves shares 32 poses and
runs the fused layer pairs on v_mfma_f32_32x32x16_f16, each wave reading only HALF of the weight tiles.  Nothing here is
product code yet: the model pins the data layout before a kernel is written, exactly as tests/lane_model.py did for the
16-pose kernels (which then ran correctly on their first launch).

What it establishes (CPU only, float64 arithmetic -- the layout is the subject, not the rounding):
  * the C/D layout of the 32x32 MFMAs (cdna_hip_programming.md: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
    chains into the B operand of the next layer without leaving the lane, given the k permutation RHO in the weight tiles:
    registers 0..7 of a 32-row tile are the k-block of its rows 0..15, registers 8..15 that of rows 16..31;
  * part A with the CONTRACTION split between the two waves (each holds half of the input k-blocks), part B with the
    OUTPUT ROWS split (each holds half of the accumulators): a wave reads half of the weight tiles of a phase;
  * per chunk the waves exchange 2 KB of partial sums and 2 KB of activated operand halves (through LDS in a kernel), and
    NOTHING between phases: the rows a wave accumulates in part B are the k-blocks it contracts over in the next part A.
"""
import numpy as np

LANES = 64


def rho(h, j):
    """row (within a 16-row block) that register j (0..7) of lane half h holds in the C/D layout"""
    return (j & 3) + 8 * (j >> 2) + 4 * h


def mfma_32x32x16(a, b, c):
    """a, b: [64 lanes][8] operand elements, c: [64][16] accumulator registers; documented lane semantics"""
    A = np.zeros((32, 16))
    B = np.zeros((16, 32))
    for lane in range(LANES):
        m, h = lane & 31, lane >> 5
        A[m, 8 * h:8 * h + 8] = a[lane]
        B[8 * h:8 * h + 8, m] = b[lane]
    D = A @ B
    out = c.copy()
    for lane in range(LANES):
        n, h = lane & 31, lane >> 5
        for r in range(16):
            out[lane, r] += D[(r & 3) + 8 * (r >> 2) + 4 * h, n]
    return out


def tile32(M, nt, kb):
    """weight tile (32 rows x 16 k) as the A operand, with the k permutation that matches the producing layer's C/D
    layout: element j of lane (h, m) = M[32 nt + m][16 kb + rho(h, j)]"""
    t = np.zeros((LANES, 8))
    for lane in range(LANES):
        m, h = lane & 31, lane >> 5
        for j in range(8):
            t[lane, j] = M[32 * nt + m, 16 * kb + rho(h, j)]
    return t


def operand_from_rows(X, kb):
    """B operand of k-block kb from a [K][32 poses] matrix, in the SAME permuted order (what a lane's own C/D registers give)"""
    b = np.zeros((LANES, 8))
    for lane in range(LANES):
        n, h = lane & 31, lane >> 5
        for j in range(8):
            b[lane, j] = X[16 * kb + rho(h, j), n]
    return b


def kblocks_of_tile(d):
    """one 32-row C/D tile -> its two k-blocks, lane-local: registers 0..7 and 8..15"""
    return d[:, 0:8].copy(), d[:, 8:16].copy()


def lrelu(z):
    return np.where(z > 0, z, 0.01 * z)


def fused_pair_p32(x_blocks, WA, bA, WB, accs, stats):
    """One fused layer pair on a wave pair.  x_blocks[w]: the input k-blocks wave w holds (half of them, B operands);
    accs[w]: the accumulator tiles (32 rows each) of the output rows wave w owns, preloaded with layer B's bias.
    WA: [R_A][K_A], WB: [N_B][R_A].  Returns nothing: accs are updated in place."""
    K_A, R_A, N_B = WA.shape[1], WA.shape[0], WB.shape[0]
    nkb, half = K_A // 16, K_A // 32
    nt_half = N_B // 64
    for c in range(R_A // 32):                       # chunk = one 32-row tile of the intermediate layer
        # ---- part A: each wave contracts over ITS half of the k-blocks
        part = []
        for w in range(2):
            p = np.zeros((LANES, 16))
            for i in range(half):
                kb = w * half + i
                p = mfma_32x32x16(tile32(WA, c, kb), x_blocks[w][i], p)
                stats["tiles_read"][w] += 1
            part.append(p)
        # ---- exchange 1: wave 0 finalises rows 0..15 (registers 0..7), wave 1 rows 16..31 (registers 8..15)
        z = [part[0][:, 0:8] + part[1][:, 0:8], part[0][:, 8:16] + part[1][:, 8:16]]
        stats["exchanged_floats"] += 2 * LANES * 8
        y = []
        for w in range(2):
            bias = np.zeros((LANES, 8))
            for lane in range(LANES):
                for j in range(8):
                    bias[lane, j] = bA[32 * c + 16 * w + rho(lane >> 5, j)]
            y.append(lrelu(z[w] + bias))             # = the B operand of k-block 2c + w of part B, lane-local
        # ---- exchange 2: both waves need both k-blocks of the chunk
        stats["exchanged_floats"] += 2 * LANES * 8
        # ---- part B: each wave updates ITS half of the output rows
        for w in range(2):
            for t in range(nt_half):
                nt = w * nt_half + t
                for q in range(2):
                    accs[w][t] = mfma_32x32x16(tile32(WB, nt, 2 * c + q), y[q], accs[w][t])
                    stats["tiles_read"][w] += 1
    assert nkb == 2 * half


def test_pair32_layout_chains_two_fused_pairs():
    rng = np.random.default_rng(0)
    K0, R0, N0, R1, N1 = 128, 256, 128, 256, 64          # (lin0,lin1) then (lin2,lin3), scaled down; 32 poses
    X = rng.normal(size=(K0, 32))
    W0, b0 = rng.normal(size=(R0, K0)) / np.sqrt(K0), rng.normal(size=R0) * 0.1
    W1, b1 = rng.normal(size=(N0, R0)) / np.sqrt(R0), rng.normal(size=N0) * 0.1
    W2, b2 = rng.normal(size=(R1, N0)) / np.sqrt(N0), rng.normal(size=R1) * 0.1
    W3, b3 = rng.normal(size=(N1, R1)) / np.sqrt(R1), rng.normal(size=N1) * 0.1
    # reference
    x2 = lrelu(W1 @ lrelu(W0 @ X + b0[:, None]) + b1[:, None])
    z4 = W3 @ lrelu(W2 @ x2 + b2[:, None]) + b3[:, None]

    def bias_tiles(b, w, nt_half):
        tiles = []
        for t in range(nt_half):
            d = np.zeros((LANES, 16))
            for lane in range(LANES):
                for r in range(16):
                    d[lane, r] = b[32 * (w * nt_half + t) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)]
            tiles.append(d)
        return tiles

    stats = {"tiles_read": [0, 0], "exchanged_floats": 0}
    # wave w holds k-blocks [w * K0/32, (w + 1) * K0/32) of the input
    xb = [[operand_from_rows(X, w * (K0 // 32) + i) for i in range(K0 // 32)] for w in range(2)]
    acc = [bias_tiles(b1, w, N0 // 64) for w in range(2)]
    fused_pair_p32(xb, W0, b0, W1, acc, stats)
    # activation of the accumulator layer, lane-local; its tiles ARE the next phase's k-blocks of the same wave
    xb2 = []
    for w in range(2):
        blocks = []
        for d in acc[w]:
            lo, hi = kblocks_of_tile(lrelu(d))
            blocks += [lo, hi]
        xb2.append(blocks)
    assert len(xb2[0]) == N0 // 32                       # half of the next contraction's k-blocks, with no exchange
    acc2 = [bias_tiles(b3, w, N1 // 64) for w in range(2)]
    fused_pair_p32(xb2, W2, b2, W3, acc2, stats)
    # read the result back through the documented C/D layout
    got = np.zeros_like(z4)
    for w in range(2):
        for t, d in enumerate(acc2[w]):
            for lane in range(LANES):
                for r in range(16):
                    got[32 * (w * (N1 // 64) + t) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31] = d[lane, r]
    assert np.abs(got - z4).max() < 1e-10
    # each wave read exactly half of the weight tiles of both phases
    total_tiles = (R0 // 32) * (K0 // 16) + (N0 // 32) * (R0 // 16) + (R1 // 32) * (N0 // 16) + (N1 // 32) * (R1 // 16)
    assert stats["tiles_read"] == [total_tiles // 2, total_tiles // 2]
    # exchange volume per wave and chunk: two stores + two loads of 64 lanes x 8 values, whatever the layer widths
    chunks = R0 // 32 + R1 // 32
    assert stats["exchanged_floats"] == chunks * 2 * LANES * 8 * 2


def test_pair32_exchange_is_small_at_amass_widths():
    """(512 -> 1024 -> 512), the phase that dominates: per chunk a wave no longer reads (32 + 32) / 2 weight tiles of 512
    elements (hi + lo: 4 B each) and instead stores + loads 2 x 2 x 512 values of 4 B through LDS: 12.5 % of the saving"""
    K_A, N_B = 512, 512
    tiles_saved = (K_A // 16 + 2 * (N_B // 32)) // 2
    exchanged = 2 * 2 * LANES * 8
    assert exchanged / (tiles_saved * 512) == 0.125
