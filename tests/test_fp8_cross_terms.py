"""VERDICT r2 item 6 -- "say what the >= 10x target costs, then try the one arithmetic that could reach it": the cross terms
Wh xl + Wl xh of the split-precision trunk on scaled fp8 (double-rate MFMA), emulated on the CPU (tests/fp8_cross_model.py).
Kill criterion set BEFORE the run: p95 of the error of d and of d d / d q <= 3e-5 on every sweep weight set and activation.
It fails by one to two orders of magnitude (an e4m3 operand carries 4 significant bits: the cross terms, 2^-11 of the
product, come out 2^-4 accurate, i.e. the product 2^-15 = 3e-5 PER LAYER, twelve layers deep) -- so precision `f16f8` was
not built.  This test pins the numbers behind that paragraph of DESIGN.md; the full 6 x 3 table is in profiles/r03/."""
import numpy as np
import pytest

import fp8_cross_model as fm
from conftest import d_rows, rel_err_rows
from oracle import posendf_np as onp
from posendf_amd import synth


def test_e4m3_rounding_model():
    x = np.array([0.0, 1.0, 1.0625, 1.07, 448.0, 500.0, 2.0 ** -9, 2.0 ** -10 * 0.9, -3.3])
    assert np.array_equal(fm.q_e4m3(x), np.array([0.0, 1.0, 1.0, 1.125, 448.0, 448.0, 2.0 ** -9, 0.0, -3.25]))


@pytest.mark.parametrize("weights", [(0, 2.0, 0.1), (4, 2.5, 0.05)], ids=["live", "s4g25"])
@pytest.mark.parametrize("act", ["lrelu", "softplus"])
def test_fp8_cross_terms_miss_the_kill_criterion(weights, act):
    sd = synth.make_weights(*weights)
    q = synth.make_poses(128, seed=77)
    d64, g64 = onp.forward_grad(q, sd, act, dtype=np.float64)
    p95 = {}
    for mode in ("f16x3", "f16f8"):
        d, g = fm.forward_grad(q, sd, act, mode, block=32)
        p95[mode] = (np.percentile(d_rows(d, d64), 95), np.percentile(rel_err_rows(g, g64), 95))
    print(weights, act, p95)
    assert max(p95["f16x3"]) < 3e-5                     # the product's arithmetic passes the same criterion with room
    assert max(p95["f16f8"]) > 3e-5                     # the fp8 cross terms do not
    assert p95["f16f8"][1] > 30 * p95["f16x3"][1]       # ... by more than an order of magnitude on the gradient


@pytest.mark.parametrize("act", ["lrelu", "softplus"])
def test_int8_cross_terms_miss_the_kill_criterion(act):
    """the cross terms on int8 (v_mfma_i32_16x16x64_i8, double rate) -- `f16i8` with a scale per operand
    row / pose and two int32 accumulations, `f16i8c` the ONE-instruction form [Wh | Wl] . [xl ; xh] whose lo scales are tied to
    the hi scales.  Kill criterion set before the run: p95 of d or d d / d q above 3e-5 on any weight set.  int8 is fixed point
    against the largest element of a row / pose (crest factors of 5-10 after an activation): it buys 2.5-3x over e4m3, not the
    16x its bit count suggests, and the gain-2.5 held-out set fails by an order of magnitude; the tied form is no better than
    e4m3.  No micro-benchmark, no kernel (profiles/r05/int8_cross_terms.txt has the 6 x 2 table)."""
    sd = synth.make_weights(4, 2.5, 0.05)
    q = synth.make_poses(128, seed=77)
    d64, g64 = onp.forward_grad(q, sd, act, dtype=np.float64)
    p95 = {}
    for mode in ("f16x3", "f16f8", "f16i8", "f16i8c"):
        d, g = fm.forward_grad(q, sd, act, mode, block=32)
        p95[mode] = (np.percentile(d_rows(d, d64), 95), np.percentile(rel_err_rows(g, g64), 95))
    print(act, p95)
    assert max(p95["f16x3"]) < 3e-5
    assert max(p95["f16i8"]) > 3e-5 and max(p95["f16i8c"]) > 3e-5          # both arms miss the criterion on this set
    assert p95["f16i8"][0] < 0.7 * p95["f16f8"][0]                         # ... although int8 does beat e4m3
    assert p95["f16i8c"][0] > p95["f16i8"][0]                              # and the tie of the one-instruction form costs bits
