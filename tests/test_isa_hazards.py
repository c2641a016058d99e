"""Static ISA checks of the shipped library (tools/isa_hazards.py): registers written inside inline asm and read
by an MFMA as an operand keep their two wait states, and nothing but the weight ring writes M0 in a kernel that issues
LDS-DMA.  Needs the built library, no GPU."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import isa_hazards  # noqa: E402

LIB = os.path.join(REPO, "posendf_amd", "lib", "libposendf_amd.so")
LIB_DEBUG = os.path.join(REPO, "posendf_amd", "lib", "libposendf_amd_debug.so")      # (the instrumented builds: the same rules)


def test_checker_flags_a_short_distance_and_a_foreign_m0_write():
    mf = ("v_mfma_f32_16x16x32_f16", "a[0:3], v[10:13], v[20:23], a[0:3]")
    near = [("v_fma_mixhi_f16", "v21, v1, -1.0, v2 op_sel:[1,0,0]"), ("v_mov_b32_e32", "v5, v6"), mf]
    bad, closest, sites = isa_hazards.check_wait_states(near)
    assert sites == 1 and closest == 1 and len(bad) == 1
    padded = [near[0], ("s_nop", "1"), mf]
    assert isa_hazards.check_wait_states(padded)[0] == [] and isa_hazards.check_wait_states(padded)[1] == 2
    other = [near[0], mf[:1] + ("a[0:3], v[10:13], v[30:33], a[0:3]",)]           # another register: no hazard
    assert isa_hazards.check_wait_states(other)[0] == []
    ring = [("s_mov_b32", "m0, s5"), ("s_nop", "0"), ("global_load_lds_dwordx4", "v1, s[2:3]")]
    assert isa_hazards.check_m0(ring) == ([], 1)
    assert len(isa_hazards.check_m0(ring + [("s_mov_b32", "m0, s7"), ("v_mov_b32_e32", "v1, v2"), ("v_mov_b32_e32", "v1, v2"), ("v_mov_b32_e32", "v1, v2")])[0]) == 1
    assert len(isa_hazards.check_m0(ring + [("s_set_gpr_idx_on", "s3, gpr_idx(SRC0)")])[0]) == 1


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built")
def test_shipped_kernels_keep_their_wait_states_and_m0():
    ks = isa_hazards.kernels(LIB)
    assert any(k.startswith("pndf_fused_split_relu_kernel") for k in ks) and any(k.startswith("pndf_generic_") for k in ks)
    assert not any("timing" in k or "_dbg" in k or "probe" in k for k in ks), sorted(ks)      # those live in the debug library
    if os.path.exists(LIB_DEBUG):
        dk = isa_hazards.kernels(LIB_DEBUG)
        assert any(k.endswith("_timing") for k in dk)
        ks = {**ks, **dk}
    findings = []
    for name, ins in ks.items():
        bad_w, _, _ = isa_hazards.check_wait_states(ins)
        bad_m, _ = isa_hazards.check_m0(ins)
        findings += [(name, b) for b in bad_w] + [(name, b) for b in bad_m]
    assert not findings, findings
