#!/usr/bin/env python3
"""Host-side AddressSanitizer pass over the C ABI's host-only code (`-fsanitize=address` build option).

Builds the library with `-fsanitize=address -fno-gpu-sanitize` (host code instrumented, device code untouched) into
gpurun_ab/lib_asan.so and drives every entry point that needs no device -- the weight packers (`pndf_pack_host`,
`pndf_pack_host_split`: full architecture, encoder-less, narrower hidden layers), the body-model packers
(`pndf_lbs_pack_host`, `pndf_lbs_pack_split_host`: SMPL size and a ragged small model), their refusal paths, and the host twins `pndf_*_cpu` -- with
EXACTLY sized numpy buffers, in a child process that preloads the ASan runtime.  Any out-of-bounds host access of the
packers ends the child with an AddressSanitizer report.  Needs no GPU.

usage: python tools/asan_host_check.py [--no-build]      (prints "asan host check: clean" and exits 0, or the report)
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "gpurun_ab", "lib_asan.so")
FLAGS = ["-fsanitize=address", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g"]

CHILD = r"""
import ctypes, sys
import numpy as np
sys.path.insert(0, %(repo)r)
from posendf_amd import synth            # numpy only
from ctypes import POINTER, c_void_p, c_int64, c_int32, c_int
lib = ctypes.CDLL(%(lib)r)
lib.pndf_packed_sizes.argtypes = [POINTER(c_int64)] * 2
lib.pndf_packed_sizes.restype = None
for name in ("pndf_pack_host", "pndf_pack_host_split"):
    getattr(lib, name).argtypes = [POINTER(c_void_p), POINTER(c_int64), c_int, c_void_p, c_void_p]
    getattr(lib, name).restype = c_int
lib.pndf_lbs_packed_floats.argtypes = [c_int32]; lib.pndf_lbs_packed_floats.restype = c_int64
lib.pndf_lbs_packed_split_bytes.argtypes = [c_int32]; lib.pndf_lbs_packed_split_bytes.restype = c_int64
lib.pndf_lbs_pack_host.argtypes = [c_int32, c_int32] + [c_void_p] * 8 + [c_int32, c_void_p, c_void_p, c_void_p]
lib.pndf_lbs_pack_host.restype = c_int
lib.pndf_lbs_pack_split_host.argtypes = [c_int32, c_void_p, c_void_p, c_void_p]
lib.pndf_lbs_pack_split_host.restype = c_int

n0, n1 = c_int64(), c_int64()
lib.pndf_packed_sizes(ctypes.byref(n0), ctypes.byref(n1))
calls = 0
for dims in (synth.DFNET_DIMS, synth.DFNET_DIMS_NOENC, (126, 192, 384, 700, 300, 200, 48, 1), (126, 1, 16, 17, 15, 2, 1, 1)):
    sd = synth.make_weights(3, 1.5, 0.1, dims=dims)
    keys = list(synth.state_dict_shapes(dims).keys())
    arrs = [np.ascontiguousarray(sd[k]) for k in keys]
    ptrs = (c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    numel = (c_int64 * len(arrs))(*[a.size for a in arrs])
    for fn in (lib.pndf_pack_host, lib.pndf_pack_host_split):
        stream, bias = np.empty(n0.value, np.float32), np.empty(n1.value, np.float32)
        rc = fn(ptrs, numel, len(arrs), stream.ctypes.data, bias.ctypes.data)
        assert rc == 0, (dims, rc)
        calls += 1
    # refusal paths must not touch anything: one tensor short, and a wider layer
    assert lib.pndf_pack_host(ptrs, numel, len(arrs) - 1, stream.ctypes.data, bias.ctypes.data) != 0
for V, extra, nb in ((6890, synth.SMPL_EXTRA_JOINT_VERTICES, 10), (137, (5, 60, 136), 3), (17, (), 0)):
    m = synth.make_body_model(V=V, n_betas=max(nb, 1), seed=2, extra=extra)
    f = lambda k: np.ascontiguousarray(m[k], dtype=np.float32)
    vt, sd_, pd, jr, w = f("v_template"), np.ascontiguousarray(f("shapedirs")[:, :, :nb]), f("posedirs"), f("J_regressor"), f("lbs_weights")
    betas = np.linspace(-1, 1, nb).astype(np.float32) if nb else np.zeros(1, np.float32)
    par = np.array(m["parents"], np.int32); par[0] = -1
    ex = np.ascontiguousarray(np.asarray(m["extra_joint_vertex"], np.int32)) if len(extra) else np.zeros(1, np.int32)
    blob = np.empty(lib.pndf_lbs_packed_floats(V), np.float32)
    J, rel = np.empty(72, np.float32), np.empty(72, np.float32)
    rc = lib.pndf_lbs_pack_host(V, nb, vt.ctypes.data, sd_.ctypes.data if nb else None, betas.ctypes.data if nb else None, pd.ctypes.data,
                                jr.ctypes.data, par.ctypes.data, w.ctypes.data, ex.ctypes.data if len(extra) else None, len(extra),
                                blob.ctypes.data, J.ctypes.data, rel.ctypes.data)
    assert rc == 0, (V, rc)
    sblob = np.empty(lib.pndf_lbs_packed_split_bytes(V), np.uint8)
    sc = np.empty(2, np.float32)
    assert lib.pndf_lbs_pack_split_host(V, blob.ctypes.data, sblob.ctypes.data, sc.ctypes.data) == 0
    calls += 2
    if len(extra) > 1:       # a vertex named twice is refused (ADVICE r3), before anything is written
        dup = ex.copy(); dup[1] = dup[0]
        assert lib.pndf_lbs_pack_host(V, nb, vt.ctypes.data, sd_.ctypes.data if nb else None, betas.ctypes.data if nb else None,
                                      pd.ctypes.data, jr.ctypes.data, par.ctypes.data, w.ctypes.data, dup.ctypes.data, len(extra),
                                      blob.ctypes.data, None, None) != 0
# the host twins (pndf_*_cpu): every configuration, ragged batches, exactly sized pose / distance / gradient buffers
class Cfg(ctypes.Structure):
    _fields_ = [("act", c_int32), ("beta", ctypes.c_float), ("num_joints", c_int32), ("n_dims", c_int32), ("dims", c_int32 * 16),
                ("parent", c_int32 * 32), ("precision", c_int32)]
lib.pndf_default_config.argtypes = [POINTER(Cfg), c_int32, ctypes.c_float]; lib.pndf_default_config.restype = None
lib.pndf_cpu_create.argtypes = [POINTER(c_void_p), POINTER(Cfg)]
lib.pndf_cpu_load_weights.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_int64), c_int]
lib.pndf_forward_cpu.argtypes = [c_void_p, c_void_p, c_void_p, c_int64]
lib.pndf_forward_grad_cpu.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64]
lib.pndf_project_cpu.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int]
lib.pndf_cpu_destroy.argtypes = [c_void_p]
twin = 0
for act, dims in ((1, synth.DFNET_DIMS), (2, synth.DFNET_DIMS_NOENC), (0, (126, 192, 384, 700, 300, 200, 48, 1))):
    cfg = Cfg(); lib.pndf_default_config(ctypes.byref(cfg), act, 100.0)
    for i, w in enumerate(dims): cfg.dims[i] = w
    h = c_void_p(); assert lib.pndf_cpu_create(ctypes.byref(h), ctypes.byref(cfg)) == 0
    sd = synth.make_weights(1, 2.0, 0.1, dims=dims)
    arrs = [np.ascontiguousarray(sd[k]) for k in synth.state_dict_shapes(dims)]
    ptrs = (c_void_p * len(arrs))(*[a.ctypes.data for a in arrs]); numel = (c_int64 * len(arrs))(*[a.size for a in arrs])
    assert lib.pndf_cpu_load_weights(h, ptrs, numel, len(arrs)) == 0
    for B in (1, 31, 32, 33, 70):
        q = np.ascontiguousarray(synth.make_poses(B, seed=B)); d = np.empty(B, np.float32); dq = np.empty((B, 84), np.float32)
        go = np.linspace(-1, 2, B).astype(np.float32); qo = np.empty_like(q)
        assert lib.pndf_forward_cpu(h, q.ctypes.data, d.ctypes.data, B) == 0
        assert lib.pndf_forward_grad_cpu(h, q.ctypes.data, go.ctypes.data, d.ctypes.data, dq.ctypes.data, B) == 0
        assert lib.pndf_project_cpu(h, q.ctypes.data, qo.ctypes.data, d.ctypes.data, B, 2) == 0
        assert np.isfinite(qo).all() and np.isfinite(dq).all()
        twin += 3
    lib.pndf_cpu_destroy(h)
print("asan host check: clean (%%d packer calls, %%d host-twin calls)" %% (calls, twin))
"""


def main():
    sys.path.insert(0, REPO)
    import __graft_entry__ as g
    if "--no-build" not in sys.argv:
        g.build_library(LIB, flags=FLAGS, tag="_asan")
    import glob
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rt:
        raise SystemExit("no libclang_rt.asan-x86_64.so under /opt/rocm/lib/llvm")
    env = dict(os.environ, LD_PRELOAD=rt[0], ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
    p = subprocess.run([sys.executable, "-c", CHILD % dict(repo=REPO, lib=LIB)], env=env, capture_output=True, text=True, timeout=1800)
    sys.stdout.write(p.stdout)
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-6000:])
    raise SystemExit(p.returncode)


if __name__ == "__main__":
    main()
