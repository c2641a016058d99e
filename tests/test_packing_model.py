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
    stream, enc, bias = engine.pack_host(sd)
    q = synth.make_poses(16, seed=7, signed=True)
    dbg = {}
    d, dq = onp.forward_grad(q, sd, act, debug=dbg)
    d_m, gx0_m, stages = lm.trunk_wave(dbg["feat"], stream, bias, slope)
    # forward activations of the accumulator layers, distance, and d d / d feature
    assert rel_err(stages["x2"], onp._act(dbg["zs"][1], act, 100.0)) < 1e-5
    assert rel_err(stages["x4"], onp._act(dbg["zs"][3], act, 100.0)) < 1e-5
    assert rel_err(stages["x6"], onp._act(dbg["zs"][5], act, 100.0)) < 1e-5
    assert d_err(d_m, d[:, 0]) < 2e-5
    assert rel_err(gx0_m[:, :126], dbg["gx"][0]) < 2e-5
    assert np.all(gx0_m[:, 126:] == 0)


def test_packed_blocks_layout():
    from posendf_amd import engine, synth
    sd = synth.make_weights(3)
    stream, enc, bias = engine.pack_host(sd)
    assert stream.size == 10624 * 256
    # bias block
    assert np.array_equal(bias[0:256], sd["dfnet.lin0.bias"])
    assert np.array_equal(bias[2560:2624], sd["dfnet.lin5.bias"])
    assert np.array_equal(bias[2624:2688], sd["dfnet.lin6.weight"][0])
    assert bias[2688] == sd["dfnet.lin6.bias"][0]
    # encoder block: joint 0 (root, 120 floats) then joint 3 (first child) at 360
    assert np.array_equal(enc[0:40], sd["enc.net.0.net.0.weight"].ravel())
    assert np.array_equal(enc[40:50], sd["enc.net.0.net.0.bias"])
    assert np.array_equal(enc[52:112], sd["enc.net.0.net.2.weight"].ravel())
    assert np.array_equal(enc[112:118], sd["enc.net.0.net.2.bias"])
    assert np.array_equal(enc[360:460], sd["enc.net.3.net.0.weight"].ravel())
    assert np.array_equal(enc[360 + 112:360 + 172], sd["enc.net.3.net.2.weight"].ravel())
    # first tile of the stream = tile(W0 padded, nt=0, kt=0): lane l -> W0[l & 15][4 (l >> 4) + s]
    w0 = sd["dfnet.lin0.weight"]
    t0 = stream[:256].reshape(64, 4)
    for lane in (0, 5, 17, 63):
        assert np.array_equal(t0[lane], w0[lane & 15, 4 * (lane >> 4):4 * (lane >> 4) + 4])
    # every weight appears in the stream exactly twice (forward tile + transposed tile): checksum of squares
    tot = sum(float((sd[f"dfnet.lin{l}.weight"].astype(np.float64) ** 2).sum()) for l in range(6))
    assert abs(float((stream.astype(np.float64) ** 2).sum()) - 2 * tot) < 1e-6 * tot
