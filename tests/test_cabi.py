"""CPU-side checks of the C ABI: the library loads, exports every symbol include/posendf_amd.h declares,
validates arguments, and refuses to run without a gfx950 device (no CPU fallback).  No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from posendf_amd import engine
    return engine.load_library()


def _declared(header):
    return set(re.findall(r"\b(pndf_[a-z0-9_]+)\s*\(", open(os.path.join(REPO, "include", header)).read()))


def test_exports_match_header(lib):
    declared = _declared("posendf_amd.h")
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    from posendf_amd import engine
    assert set(engine.EXPORTS) == declared


def test_debug_entry_points_live_in_their_own_header(lib):
    """the drop-in boundary carries no bring-up / profiling entry point -- those are declared in
    include/posendf_amd_debug.h, which a maintainer binding the library never includes."""
    from posendf_amd import engine
    public = open(os.path.join(REPO, "include", "posendf_amd.h")).read()
    assert "debug" not in public.lower()
    dbg = _declared("posendf_amd_debug.h")
    assert dbg == set(engine.DEBUG_EXPORTS) and all(n.startswith("pndf_debug_") for n in dbg)
    assert not (dbg & set(engine.EXPORTS))
    for name in dbg:
        assert hasattr(lib, name), f"{name} declared in the debug header but not exported"      # (answered by the debug library, lazily)
    # ... and the PRODUCT library carries none of it: no pndf_debug_* symbol, no instrumented / stage-dump kernel, no probe
    import subprocess
    import __graft_entry__ as ge
    syms = subprocess.run(["nm", "-D", "--defined-only", ge.LIB], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in syms.splitlines() if ln.strip()]
    assert any(n == "pndf_project" for n in names) and any(n == "pndf_fused_split_relu_kernel" for n in names)
    leaked = [n for n in names if "debug" in n or "timing" in n or n.endswith("_dbg") or "probe" in n]
    assert not leaked, leaked
    dsyms = subprocess.run(["nm", "-D", "--defined-only", engine.debug_library_path(ge.LIB)], capture_output=True, text=True, check=True).stdout
    assert all(any(ln.split()[-1] == n for ln in dsyms.splitlines() if ln.strip()) for n in dbg)


def test_product_library_carries_no_experiment(lib, tmp_path):
    """The lab is quarantined from the product (csrc/pndf_experiment.h): every translation unit of the library exports one
    word with a bit per tuning / ablation macro that differed from its product default at build time; all of them are 0 in
    the library the package loads, pndf_version() says so, and an experiment macro without the umbrella does not compile."""
    import subprocess
    from posendf_amd import engine
    for name in engine.EXPERIMENT_WORDS:
        assert ctypes.c_uint.in_dll(lib, name).value == 0, name
    for name in engine.DEBUG_EXPERIMENT_WORDS:      # the debug library built next to it: the same rule
        assert ctypes.c_uint.in_dll(lib._debug(), name).value == 0, name
    assert engine.experiment_word(lib) == 0 and lib.pndf_experiment_word() == 0 and lib.pndf_debug_experiment_word() == 0
    assert b"experiments=0x00000000" in lib.pndf_version()
    hdr = os.path.join(REPO, "posendf_amd", "csrc", "pndf_experiment.h")
    cc = ["gcc", "-x", "c", "-fsyntax-only", hdr]
    assert subprocess.run(cc, capture_output=True).returncode == 0
    for macro in ("PNDF_ABLATE=2", "PNDF_SP_DIAG=4", "PNDF_RING_ALIAS_F", "PNDF_NT_MODE=1", "PNDF_RING_SLOTS=4", "PNDF_DMA_EARLY=0",
                  "PNDF_LBS_DIAG=1", "PNDF_EXP_LO_BITS=4", "PNDF_RING_PIECES=2", "PNDF_RING_STAMPS=1"):
        bad = subprocess.run(cc + ["-D" + macro], capture_output=True, text=True)
        assert bad.returncode != 0 and "PNDF_EXPERIMENT" in bad.stderr, macro
        assert subprocess.run(cc + ["-D" + macro, "-DPNDF_EXPERIMENT=1"], capture_output=True).returncode == 0, macro
    # the word itself: a probe translation unit built with an arm set reports the arm's bit and the umbrella's
    src = tmp_path / "word.c"
    src.write_text('#include "pndf_experiment.h"\n#include <stdio.h>\nint main(void) { printf("%08x\\n", PNDF_EXPERIMENT_WORD); return 0; }\n')
    exe = tmp_path / "word"
    inc = ["-I", os.path.dirname(hdr)]
    for flags, want in (([], "00000000"), (["-DPNDF_EXPERIMENT=1", "-DPNDF_ABLATE=1024"], "80000001"),
                        (["-DPNDF_EXPERIMENT=1", "-DPNDF_SP_DIAG=32", "-DPNDF_DMA_EARLY=0"], "80000202"),
                        (["-DPNDF_TU_RING_PIECES=2", "-DPNDF_TU_RING_STAMPS=1"], "00000000")):      # a wrapper unit's structural values
        subprocess.run(["gcc", *inc, *flags, str(src), "-o", str(exe)], check=True)
        assert subprocess.run([str(exe)], capture_output=True, text=True).stdout.strip() == want, flags
    # and the build entry refuses flags for the product path
    import __graft_entry__ as ge
    with pytest.raises(ValueError, match="product library takes no compile flags"):
        ge.build_library(ge.LIB, flags=["-DPNDF_ABLATE=2"])


def test_default_config(lib):
    from posendf_amd.engine import PndfConfig
    cfg = PndfConfig()
    lib.pndf_default_config(ctypes.byref(cfg), 1, 100.0)
    assert cfg.num_joints == 21 and cfg.n_dims == 8
    assert list(cfg.dims[:8]) == [126, 256, 512, 1024, 512, 256, 64, 1]
    assert list(cfg.parent[:21]) == [-1, -1, -1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19]


def test_create_rejects_unsupported(lib):
    from posendf_amd.engine import PndfConfig
    h = ctypes.c_void_p()
    cfg = PndfConfig()
    lib.pndf_default_config(ctypes.byref(cfg), 7, 100.0)            # unknown activation code
    assert lib.pndf_create(ctypes.byref(h), ctypes.byref(cfg), 0) == -4
    assert b"activation" in lib.pndf_last_error(None)
    lib.pndf_default_config(ctypes.byref(cfg), 2, -1.0)             # softplus needs beta > 0
    assert lib.pndf_create(ctypes.byref(h), ctypes.byref(cfg), 0) == -1
    lib.pndf_default_config(ctypes.byref(cfg), 2, 100.0)
    cfg.precision = 2                                               # the plain-f16 comparison kernel is relu-family only
    assert lib.pndf_create(ctypes.byref(h), ctypes.byref(cfg), 0) == -4
    lib.pndf_default_config(ctypes.byref(cfg), 1, 100.0)
    cfg.precision = 9                                               # unknown precision code
    assert lib.pndf_create(ctypes.byref(h), ctypes.byref(cfg), 0) == -4
    lib.pndf_default_config(ctypes.byref(cfg), 1, 100.0)
    assert cfg.precision == 0
    cfg.dims[2] = 2048                                              # wider than 1024: refused
    assert lib.pndf_create(ctypes.byref(h), ctypes.byref(cfg), 0) == -4
    cfg.dims[2] = 512
    cfg.n_dims = 10                                                 # deeper than n_dims 9: refused
    assert lib.pndf_create(ctypes.byref(h), ctypes.byref(cfg), 0) == -4
    cfg.n_dims = 8
    # (wider than amass.yaml up to 1024, and other depths, run on the runtime-planned kernels: tests/test_depth.py)
    cfg.dims[2] = 640
    assert lib.pndf_create(ctypes.byref(h), ctypes.byref(cfg), 0) in (0, -6)
    if h.value:
        lib.pndf_destroy(h)
        h = ctypes.c_void_p()
    cfg.dims[2] = 512
    cfg.dims[2] = 384                                               # narrower: accepted (runs zero padded) -- the
    assert lib.pndf_create(ctypes.byref(h), ctypes.byref(cfg), 0) in (0, -6)    # next hurdle is the device
    if h.value:
        lib.pndf_destroy(h)
        h = ctypes.c_void_p()
    cfg.dims[2] = 512
    cfg.parent[3] = 0
    assert lib.pndf_create(ctypes.byref(h), ctypes.byref(cfg), 0) == -4


def test_no_device_is_a_loud_error(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from posendf_amd.engine import Engine, PndfError
    with pytest.raises(PndfError, match="no HIP device|no CPU fallback"):
        Engine("lrelu")


def test_pack_host_rejects_bad_tables(lib):
    from posendf_amd import engine, synth
    sd = synth.make_weights(1)
    stream, bias = engine.pack_host(sd, lib)
    assert np.isfinite(stream).all()
    bad = dict(sd)
    bad["dfnet.lin3.weight"] = bad["dfnet.lin3.weight"][:, :-1]
    with pytest.raises(engine.PndfError):
        engine.pack_host(bad, lib)


def test_split_packer_scales_each_layer_and_refuses_the_unscalable(lib):
    """The fp16 hi/lo split carries s_l W with a per-layer power of two s_l (largest |weight| -> [2^12, 2^13)), so any
    weight magnitude packs; a layer without a finite non-zero weight cannot be scaled and is refused, not degraded."""
    from posendf_amd import engine, synth
    from lane_model import SCALE_OFF
    sd = synth.make_weights(1)
    _, bias = engine.pack_host(sd, lib, split=True)
    for scale_by in (1e-6, 1e7):                                 # far outside any fixed scaling window
        odd = dict(sd)
        odd["dfnet.lin2.weight"] = sd["dfnet.lin2.weight"] * np.float32(scale_by)
        stream, b2 = engine.pack_host(odd, lib, split=True)
        trunk = stream.reshape(-1, 256)[48:48 + 10624].view(np.float16).astype(np.float32)     # between the encoder sections
        assert np.isfinite(trunk).all() and np.abs(trunk).max() <= 2.0 ** 13            # hi = rne(scaled weight) may round up to 2^13
        inv = b2[SCALE_OFF:SCALE_OFF + 6]
        assert (np.log2(inv) == np.round(np.log2(inv))).all()
        mx = np.abs(odd["dfnet.lin2.weight"]).max() / inv[2]
        assert 2.0 ** 12 <= mx < 2.0 ** 13
        assert (inv[[0, 1, 3, 4, 5]] == bias[SCALE_OFF:SCALE_OFF + 6][[0, 1, 3, 4, 5]]).all()
    zero = dict(sd)
    zero["dfnet.lin4.weight"] = np.zeros_like(sd["dfnet.lin4.weight"])
    engine.pack_host(zero, lib)                                  # fine for the exact fp32 stream
    with pytest.raises(engine.PndfError):
        engine.pack_host(zero, lib, split=True)
    nan = dict(sd)
    w = sd["dfnet.lin0.weight"].copy()
    w[3, 5] = np.nan
    nan["dfnet.lin0.weight"] = w
    with pytest.raises(engine.PndfError):
        engine.pack_host(nan, lib, split=True)


def test_facade_surface_cpu():
    """Reference-compatible surface without touching the engine: state-dict keys, train=True objective
    (pinned against the reference in tests/golden); train=False of a `train.device: cpu` model runs on the library's host
    twins (tests/test_cpu_twin.py holds them to the golden vectors), any other device type is refused."""
    import torch
    from conftest import golden_weights, load_golden
    from posendf_amd import PoseNDF, amass_config, synth
    from posendf_amd.engine import PndfError
    net = PoseNDF(amass_config("lrelu", "cpu"))
    assert list(net.state_dict().keys()) == list(synth.state_dict_shapes().keys())
    assert sum(p.numel() for p in net.parameters()) == 1365565
    net.load_state_dict({k: torch.from_numpy(v) for k, v in golden_weights("live").items()})
    assert net.eval() is net
    g = load_golden("lrelu", "live")
    loss, ld = net(torch.from_numpy(g["train_q"]).clone(), torch.from_numpy(g["train_dist"]),
                   torch.from_numpy(g["train_man"]), train=True, eikonal=1.0)
    assert abs(loss.item() - g["train_loss"]) < 1e-6
    assert abs(ld["man_loss"].item() - g["train_man_loss"]) < 1e-6
    assert abs(ld["eikonal"].item() - g["train_eikonal"]) < 1e-5
    loss.backward()                                                 # weight gradients flow (train_posendf.py:98)
    assert net.dfnet.lin0.weight.grad is not None
    d = net(torch.from_numpy(g["q"]), train=False)["dist_pred"]
    assert d.shape == (len(g["q"]), 1) and d.device.type == "cpu"
    assert net._engine_for(torch.device("cpu")).kernel_name() == "pndf_cpu (host twin)"
    with pytest.raises(PndfError):
        net._engine_for(torch.device("meta"))


def test_reference_checkpoint_interchange(tmp_path):
    """the reference's on-disk checkpoint (model/train_posendf.py:147-156: a dict with
    'epoch' / 'model_state_dict' / 'optimizer_state_dict', legacy non-zip serialisation) loads unchanged, the way
    experiments/sample_poses.py:90-91 does it, and round-trips."""
    import torch
    from conftest import golden_weights
    from posendf_amd import PoseNDF, amass_config
    src = PoseNDF(amass_config("lrelu", "cpu"))
    src.load_state_dict({k: torch.from_numpy(v) for k, v in golden_weights("mixed").items()})
    opt = torch.optim.Adam(src.parameters(), lr=1e-5, weight_decay=1e-4)          # train_posendf.py:30
    path = tmp_path / "checkpoint_epoch_best.tar"
    torch.save({"epoch": 7, "model_state_dict": src.state_dict(), "optimizer_state_dict": opt.state_dict()}, path,
               _use_new_zipfile_serialization=False)
    ckpt = torch.load(path, map_location="cpu")["model_state_dict"]                # sample_poses.py:90
    dst = PoseNDF(amass_config("lrelu", "cpu"))
    dst.load_state_dict(ckpt)                                                      # sample_poses.py:91
    dst.eval()
    for (ka, a), (kb, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert ka == kb and torch.equal(a, b)
    # the packed engine weights are a pure function of the state dict
    from posendf_amd import engine
    s1, b1 = engine.pack_host({k: v.numpy() for k, v in src.state_dict().items()})
    s2, b2 = engine.pack_host({k: v.numpy() for k, v in dst.state_dict().items()})
    assert np.array_equal(s1, s2) and np.array_equal(b1, b2)


def test_header_is_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/posendf_amd.h must compile as C99 (no C++-isms, no torch / HIP types in
    the signatures) and a C translation unit must be able to name every entry point."""
    import shutil
    import subprocess
    from conftest import REPO
    from posendf_amd import engine
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "use_header.c"
    refs = "\n".join(f"    (void)&{name};" for name in engine.EXPORTS + engine.DEBUG_EXPORTS)
    src.write_text('#include "posendf_amd.h"\n#include "posendf_amd_debug.h"\nint main(void) {\n' + refs + "\n    return sizeof(pndf_config) > 0 ? 0 : 1;\n}\n")
    r = subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(REPO, "include"),
                        str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_weight_fingerprint_sees_replaced_parameters_and_modules():
    """The facade re-packs the engine's weights when the fingerprint changes.  Its cached walk of the module tree is validated
    by identity on every call: in-place updates, a Parameter replaced by attribute assignment, a swapped submodule and
    load_state_dict(assign=True) all change it (the cache used to go stale on the middle two)."""
    import torch
    import torch.nn as nn
    from posendf_amd import PoseNDF, amass_config
    net = PoseNDF(amass_config("lrelu", "cpu"))
    f = [net._fingerprint()]
    assert net._fingerprint() == f[0]
    with torch.no_grad():
        net.dfnet.lin6.bias.add_(1.0)
    f.append(net._fingerprint())
    net.dfnet.lin0.weight = nn.Parameter(net.dfnet.lin0.weight.detach().clone())
    f.append(net._fingerprint())
    old = net.dfnet.lin3
    net.dfnet.lin3 = nn.Linear(old.in_features, old.out_features)
    f.append(net._fingerprint())
    net.enc.net[4].net[0].bias = nn.Parameter(torch.zeros(10))
    f.append(net._fingerprint())
    net.load_state_dict({k: v.clone() for k, v in net.state_dict().items()}, assign=True)
    f.append(net._fingerprint())
    assert len(set(f)) == len(f)
    assert len(f[-1]) == 98
