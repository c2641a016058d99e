"""CPU check of the MFMA tile permutation / stream order / register-chaining claim (tests/lane_model.py)
against the numpy oracle, using the library's real host packer.  No GPU needed."""
import numpy as np
import pytest

from conftest import d_err, golden_weights, rel_err
from oracle import posendf_np as onp


@pytest.mark.parametrize("act,slope", [("lrelu", 0.01), ("relu", 0.0)])
def test_lane_model_matches_oracle(act, slope):
    from posendf_amd import engine, synth
    import lane_model as lm
    sd = golden_weights("mixed")
    stream, bias = engine.pack_host(sd)
    q = synth.make_poses(16, seed=7, signed=True)
    dbg = {}
    d, dq = onp.forward_grad(q, sd, act, debug=dbg)
    # encoder on the MFMA pipe: features from the raw poses, and d d / d n back from d d / d feature
    feat_m, emasks = lm.encoder_fwd_wave(q, stream, bias, slope)
    assert rel_err(feat_m, dbg["feat"]) < 1e-5
    gx0_pad = np.zeros((16, 128), np.float32)
    gx0_pad[:, :126] = dbg["gx"][0]
    gn_m = lm.encoder_bwd_wave(gx0_pad, emasks, stream, slope)
    assert rel_err(gn_m, dbg["gn"].reshape(16, 84)) < 2e-5
    d_m, gx0_m, stages = lm.trunk_wave(feat_m, stream, bias, slope)
    # forward activations of the accumulator layers, distance, and d d / d feature
    assert rel_err(stages["x2"], onp._act(dbg["zs"][1], act, 100.0)) < 1e-5
    assert rel_err(stages["x4"], onp._act(dbg["zs"][3], act, 100.0)) < 1e-5
    assert rel_err(stages["x6"], onp._act(dbg["zs"][5], act, 100.0)) < 1e-5
    assert d_err(d_m, d[:, 0]) < 5e-5
    assert rel_err(gx0_m[:, :126], dbg["gx"][0]) < 2e-5
    assert np.all(gx0_m[:, 126:] == 0)


def test_packed_blocks_layout():
    from posendf_amd import engine, synth
    sd = synth.make_weights(3)
    stream, bias = engine.pack_host(sd)
    assert stream.size == 10720 * 256 and bias.size == 2692 + 21 * 32 + 8 + 12      # + weight scales + chunk-layer norms
    # bias block
    assert np.array_equal(bias[0:256], sd["dfnet.lin0.bias"])
    assert np.array_equal(bias[2560:2624], sd["dfnet.lin5.bias"])
    assert np.array_equal(bias[2624:2688], sd["dfnet.lin6.weight"][0])
    assert bias[2688] == sd["dfnet.lin6.bias"][0]
    # encoder biases: joint 3: b1 at +0..9, b2 on rows 4..9 of the second 16
    eb = bias[2692 + 32 * 3:2692 + 32 * 4]
    assert np.array_equal(eb[0:10], sd["enc.net.3.net.0.bias"]) and np.all(eb[10:16] == 0)
    assert np.array_equal(eb[20:26], sd["enc.net.3.net.2.bias"]) and np.all(eb[16:20] == 0) and np.all(eb[26:] == 0)
    # encoder forward tiles open the stream: tile 2j = W1 of joint j, tile 2j+1 = W2 on rows 4..9
    t = stream[: 48 * 256].reshape(48, 64, 4)
    w1, w2 = sd["enc.net.3.net.0.weight"], sd["enc.net.3.net.2.weight"]
    for lane in (0, 9, 21, 40):
        r, g = lane & 15, lane >> 4
        want = np.zeros(4, np.float32)
        for s_ in range(4):
            k = 4 * g + s_
            want[s_] = w1[r, k] if (r < 10 and k < 10) else 0.0
        assert np.array_equal(t[6][lane], want)
        want = np.zeros(4, np.float32)
        for s_ in range(4):
            k = 4 * g + s_
            want[s_] = w2[r - 4, k] if (4 <= r < 10 and k < 10) else 0.0
        assert np.array_equal(t[7][lane], want)
    assert np.all(t[42:] == 0)
    # first trunk tile = tile(W0 padded, nt=0, kt=0): lane l -> W0[l & 15][4 (l >> 4) + s]
    w0 = sd["dfnet.lin0.weight"]
    t0 = stream[48 * 256:49 * 256].reshape(64, 4)
    for lane in (0, 5, 17, 63):
        assert np.array_equal(t0[lane], w0[lane & 15, 4 * (lane >> 4):4 * (lane >> 4) + 4])
    # every weight appears in the stream exactly twice (forward tile + transposed tile): checksum of squares
    tot = sum(float((sd[f"dfnet.lin{l}.weight"].astype(np.float64) ** 2).sum()) for l in range(6))
    tot += sum(float((sd[f"enc.net.{j}.net.{k}.weight"].astype(np.float64) ** 2).sum()) for j in range(21) for k in (0, 2))
    assert abs(float((stream.astype(np.float64) ** 2).sum()) - 2 * tot) < 1e-6 * tot


@pytest.mark.parametrize("weights", ["mixed", (3, 0.5, 0.3), (2, 3.0, 0.05), "huge-lin3", "tiny-lin1", "narrow"], ids=str)
def test_split_precision_lane_model(weights):
    """f16 x 3 trunk (pndf_kernel_split.hip) modelled at lane level with the real split packer: checks the block
    permutation, the software-pipelined stream order, the per-pose operand scaling and -- the point of the scheme -- that
    three fp16 MFMAs per product block keep fp32-class accuracy (vs the fp64 oracle, next to the fp32 oracle's own
    error) whatever the magnitudes: a small-gain net (gradients ~1e-6: the fixed 2^10 gradient scale of round 1 lost the
    lo halves there), a large-gain one, and layers scaled by 1e+4 / 1e-4 (nothing may overflow or flush)."""
    from posendf_amd import engine, synth
    import lane_model as lm
    if isinstance(weights, tuple):
        sd = synth.make_weights(*weights)
    elif weights == "narrow":       # hidden widths below configs/amass.yaml: packed zero padded into the same tile layout
        sd = synth.make_weights(5, 2.0, 0.1, dims=(126, 192, 384, 700, 300, 200, 48, 1))
    else:
        sd = dict(golden_weights("mixed"))
        if weights == "huge-lin3":
            sd["dfnet.lin3.weight"] = sd["dfnet.lin3.weight"] * np.float32(1e4)
            sd["dfnet.lin4.weight"] = sd["dfnet.lin4.weight"] * np.float32(1e-4)
        if weights == "tiny-lin1":
            sd["dfnet.lin1.weight"] = sd["dfnet.lin1.weight"] * np.float32(1e-4)
            sd["dfnet.lin1.bias"] = sd["dfnet.lin1.bias"] * np.float32(1e-4)
            sd["dfnet.lin2.weight"] = sd["dfnet.lin2.weight"] * np.float32(1e4)
    stream, bias = engine.pack_host(sd, split=True)
    q = synth.make_poses(16, seed=7, signed=True)
    dbg, dbg64 = {}, {}
    d32, _ = onp.forward_grad(q, sd, "lrelu", debug=dbg)
    d64, _ = onp.forward_grad(q, sd, "lrelu", dtype=np.float64, debug=dbg64)
    d_m, gx0_m, stages = lm.trunk_wave_split(dbg["feat"], stream, bias, 0.01)
    assert np.isfinite(d_m).all() and np.isfinite(gx0_m).all()
    w4 = dbg64["zs"][3].shape[1]                      # 512, or less for the narrow network (the rest is zero padding)
    assert rel_err(stages["x4"][:, :w4], onp._act(dbg64["zs"][3], "lrelu", 100.0)) < 2e-5 and np.all(stages["x4"][:, w4:] == 0)
    e_split = d_err(d_m, d64[:, 0])
    e_fp32 = d_err(d32[:, 0], d64[:, 0])
    g_split = rel_err(gx0_m[:, :126], dbg64["gx"][0])
    g_fp32 = rel_err(dbg["gx"][0], dbg64["gx"][0])
    print(weights, "split vs fp64: d", e_split, "gx0", g_split, "| fp32 oracle vs fp64: d", e_fp32, "gx0", g_fp32)
    assert e_split < max(5e-5, 3 * e_fp32), (e_split, e_fp32)
    assert g_split < max(5e-6, 3 * g_fp32), (g_split, g_fp32)
