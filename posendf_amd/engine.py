"""ctypes binding of libposendf_amd.so (include/posendf_amd.h).

This module is deliberately thin: tensors are passed as raw device pointers (`tensor.data_ptr()`),
the current HIP stream as an integer handle.  There is NO CPU fallback: if the library is missing the
import of `Engine` fails loudly, and if no gfx950 device is visible `Engine(...)` raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import numpy as np

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libposendf_amd.so")

ACT_CODES = {"relu": 0, "lrelu": 1, "softplus": 2}
PRECISION_CODES = {"fp32": 0, "f16x3": 1, "f16": 2, "bf16": 3}


class PndfConfig(ctypes.Structure):
    _fields_ = [("act", c_int32), ("beta", c_float), ("num_joints", c_int32), ("n_dims", c_int32),
                ("dims", c_int32 * 16), ("parent", c_int32 * 32), ("precision", c_int32), ("enc_act", c_int32), ("enc_beta", c_float)]


class DenoiseWeights(ctypes.Structure):
    """pndf_denoise_weights: the loss weights of one outer iteration, evaluated"""
    _fields_ = [("prior_coef", c_float), ("prior_power", c_int32), ("temp_coef", c_float), ("data_coef", c_float)]


class PndfError(RuntimeError):
    pass


def debug_library_path(product_path: str) -> str:
    """libposendf_amd.so -> libposendf_amd_debug.so (variant builds: lib_<name>.so -> lib_<name>_debug.so, next to it)"""
    base, ext = os.path.splitext(product_path)
    return base + "_debug" + ext


class _PndfLibrary(ctypes.CDLL):
    """The PRODUCT library.  It exports no `pndf_debug_*` symbol (include/posendf_amd_debug.h lives in libposendf_amd_debug.so);
    asking this object for one loads the debug library that was built next to it -- on first use only, so a process that never
    profiles maps the product library alone --, binds it to this library's `pndf_internal_*` hooks and answers from there."""

    def __getattr__(self, name):
        if name.startswith("pndf_debug_"):
            fn = getattr(self._debug(), name)
            setattr(self, name, fn)
            return fn
        return super().__getattr__(name)

    def _debug(self):
        dbg = self.__dict__.get("_pndf_debug_lib")
        if dbg is None:
            path = debug_library_path(self._name)
            if not os.path.exists(path):
                raise PndfError(f"{path} not found (the debug library is built with the product one: __graft_entry__.build())")
            dbg = ctypes.CDLL(path)
            H = c_void_p
            dbg.pndf_debug_bind.argtypes = [c_void_p, c_void_p, c_void_p]
            dbg.pndf_debug_bind.restype = c_int
            dbg.pndf_debug_experiment_word.restype = ctypes.c_uint
            dbg.pndf_debug_forward_grad.argtypes = [H, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]
            dbg.pndf_debug_forward_grad.restype = c_int
            dbg.pndf_debug_floats.restype = c_int64
            dbg.pndf_debug_project_timing.argtypes = [H, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]
            dbg.pndf_debug_project_timing.restype = c_int
            dbg.pndf_debug_timing_regions.restype = c_int
            dbg.pndf_debug_timing_layout.argtypes = [c_int]
            dbg.pndf_debug_timing_layout.restype = c_int
            dbg.pndf_debug_mem_probe.argtypes = [c_int, c_void_p, c_int]
            dbg.pndf_debug_mem_probe.restype = c_int
            dbg.pndf_debug_ring_stream.argtypes = [c_int, c_int, c_void_p]
            dbg.pndf_debug_ring_stream.restype = c_int
            hooks = [ctypes.cast(ctypes.CDLL.__getattr__(self, n), c_void_p) for n in
                     ("pndf_internal_launch", "pndf_internal_describe", "pndf_internal_fail")]
            if dbg.pndf_debug_bind(*hooks) != 0:
                raise PndfError("pndf_debug_bind failed")
            self.__dict__["_pndf_debug_lib"] = dbg
        return dbg


def load_library(path: str | None = None) -> ctypes.CDLL:
    path = path or os.environ.get("PNDF_LIBRARY") or _LIB_PATH      # PNDF_LIBRARY: A/B runs of two builds on one box
    # PyTorch is the plumbing for device memory and streams, so the library must share PyTorch's HIP runtime:
    # import torch FIRST so that its bundled libamdhip64 is the one already mapped when the loader resolves
    # this library's dependency (the other order puts two HIP runtimes in the process and pndf_create then
    # sees no device).  Without torch installed the system runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise PndfError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). The engine has no fallback path.")
    lib = _PndfLibrary(path)
    H = c_void_p
    lib.pndf_default_config.argtypes = [POINTER(PndfConfig), c_int32, c_float]
    lib.pndf_default_config.restype = None
    lib.pndf_create.argtypes = [POINTER(H), POINTER(PndfConfig), c_int]
    lib.pndf_destroy.argtypes = [H]
    lib.pndf_load_weights.argtypes = [H, POINTER(c_void_p), POINTER(c_int64), c_int]
    lib.pndf_forward.argtypes = [H, c_void_p, c_void_p, c_int64, c_void_p]
    lib.pndf_forward_grad.argtypes = [H, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]
    lib.pndf_project.argtypes = [H, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]
    lib.pndf_packed_sizes.argtypes = [POINTER(c_int64)] * 2
    lib.pndf_packed_sizes.restype = None
    lib.pndf_pack_host.argtypes = [POINTER(c_void_p), POINTER(c_int64), c_int, c_void_p, c_void_p]
    lib.pndf_pack_host_split.argtypes = [POINTER(c_void_p), POINTER(c_int64), c_int, c_void_p, c_void_p]
    lib.pndf_aa2quat.argtypes = [c_void_p, c_void_p, c_int64, c_void_p]
    lib.pndf_aa2quat.restype = c_int
    lib.pndf_denoise_update.argtypes = [c_void_p] * 8 + [c_int32] * 4 + [c_float, c_void_p]
    lib.pndf_denoise_update.restype = c_int
    lib.pndf_denoise_update_w.argtypes = [c_void_p] * 9 + [c_int32, c_int32, POINTER(DenoiseWeights), c_int32, c_float, c_void_p]
    lib.pndf_denoise_update_w.restype = c_int
    lib.pndf_lbs_terms_grad_w.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_float, c_void_p, c_void_p, c_void_p]
    lib.pndf_lbs_terms_grad_w.restype = c_int
    lib.pndf_denoise_update_body.argtypes = [c_void_p] * 9 + [c_int32] * 4 + [c_float, c_void_p]
    lib.pndf_denoise_update_body.restype = c_int
    LH = c_void_p
    lib.pndf_lbs_create.argtypes = [POINTER(LH), c_int32, c_int32] + [c_void_p] * 8 + [c_int32, c_int]
    lib.pndf_lbs_create.restype = c_int
    lib.pndf_lbs_destroy.argtypes = [LH]
    lib.pndf_lbs_num_joints.argtypes = [LH]
    lib.pndf_lbs_num_joints.restype = c_int32
    lib.pndf_lbs_num_vertices.argtypes = [LH]
    lib.pndf_lbs_num_vertices.restype = c_int32
    lib.pndf_lbs_workspace_floats.argtypes = [LH, c_int32, c_int32]
    lib.pndf_lbs_workspace_floats.restype = c_int64
    lib.pndf_lbs_forward.argtypes = [LH, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.pndf_lbs_forward.restype = c_int
    lib.pndf_lbs_terms_grad.argtypes = [LH, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]
    lib.pndf_lbs_terms_grad.restype = c_int
    lib.pndf_lbs_backward.argtypes = [LH, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]
    lib.pndf_lbs_backward.restype = c_int
    lib.pndf_lbs_packed_floats.argtypes = [c_int32]
    lib.pndf_lbs_packed_floats.restype = c_int64
    lib.pndf_lbs_pack_host.argtypes = [c_int32, c_int32] + [c_void_p] * 8 + [c_int32, c_void_p, c_void_p, c_void_p]
    lib.pndf_lbs_pack_host.restype = c_int
    lib.pndf_lbs_packed_split_bytes.argtypes = [c_int32]
    lib.pndf_lbs_packed_split_bytes.restype = c_int64
    lib.pndf_lbs_pack_split_host.argtypes = [c_int32, c_void_p, c_void_p, c_void_p]
    lib.pndf_lbs_pack_split_host.restype = c_int
    lib.pndf_lbs_set_precision.argtypes = [LH, c_int32]
    lib.pndf_lbs_set_precision.restype = c_int
    lib.pndf_lbs_precision.argtypes = [LH]
    lib.pndf_lbs_precision.restype = c_int32
    lib.pndf_lbs_last_error.argtypes = [LH]
    lib.pndf_lbs_last_error.restype = c_char_p
    lib.pndf_quat_topk.argtypes = [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p,
                                   c_void_p, c_void_p]
    lib.pndf_quat_topk.restype = c_int
    CH = c_void_p
    lib.pndf_cpu_create.argtypes = [POINTER(CH), POINTER(PndfConfig)]
    lib.pndf_cpu_destroy.argtypes = [CH]
    lib.pndf_cpu_load_weights.argtypes = [CH, POINTER(c_void_p), POINTER(c_int64), c_int]
    lib.pndf_forward_cpu.argtypes = [CH, c_void_p, c_void_p, c_int64]
    lib.pndf_forward_grad_cpu.argtypes = [CH, c_void_p, c_void_p, c_void_p, c_void_p, c_int64]
    lib.pndf_project_cpu.argtypes = [CH, c_void_p, c_void_p, c_void_p, c_int64, c_int]
    lib.pndf_cpu_last_error.argtypes = [CH]
    lib.pndf_cpu_last_error.restype = c_char_p
    for name in ("pndf_cpu_create", "pndf_cpu_destroy", "pndf_cpu_load_weights", "pndf_forward_cpu", "pndf_forward_grad_cpu",
                 "pndf_project_cpu"):
        getattr(lib, name).restype = c_int
    lib.pndf_last_error.argtypes = [H]
    lib.pndf_last_error.restype = c_char_p
    lib.pndf_version.restype = c_char_p
    lib.pndf_experiment_word.restype = ctypes.c_uint
    lib.pndf_kernel_name.argtypes = [H]
    lib.pndf_kernel_name.restype = c_char_p
    for name in ("pndf_create", "pndf_destroy", "pndf_load_weights", "pndf_forward", "pndf_forward_grad",
                 "pndf_project", "pndf_pack_host", "pndf_pack_host_split"):
        getattr(lib, name).restype = c_int
    return lib


# bring-up / profiling / measurement aids: include/posendf_amd_debug.h, NOT the drop-in boundary
DEBUG_EXPORTS = ("pndf_debug_bind", "pndf_debug_experiment_word", "pndf_debug_forward_grad", "pndf_debug_floats", "pndf_debug_project_timing",
                 "pndf_debug_timing_regions", "pndf_debug_timing_layout", "pndf_debug_mem_probe", "pndf_debug_ring_stream")
# per-translation-unit experiment words (csrc/pndf_experiment.h): data symbols, all zero in a product build
EXPERIMENT_WORDS = ("pndf_experiment_word_capi", "pndf_experiment_word_fp32", "pndf_experiment_word_split", "pndf_experiment_word_split_x2",
                    "pndf_experiment_word_lbs", "pndf_experiment_word_generic")
DEBUG_EXPERIMENT_WORDS = ("pndf_experiment_word_debug", "pndf_experiment_word_fp32_timing", "pndf_experiment_word_split_timing",
                          "pndf_experiment_word_fp32_dbg", "pndf_experiment_word_probe")


def experiment_word(lib=None) -> int:
    """OR of the library's per-translation-unit experiment words, read from the data symbols themselves (not through
    pndf_experiment_word(), which a lab build could have touched): 0 for a product build."""
    lib = lib or load_library()
    w = 0
    for name in EXPERIMENT_WORDS:
        w |= int(ctypes.c_uint.in_dll(lib, name).value)
    dbg = lib._debug() if isinstance(lib, _PndfLibrary) else None      # (the debug library next to it: the same rule)
    for name in (DEBUG_EXPERIMENT_WORDS if dbg is not None else ()):
        w |= int(ctypes.c_uint.in_dll(dbg, name).value)
    return w


# include/posendf_amd.h, the public header
EXPORTS = ("pndf_default_config", "pndf_create", "pndf_destroy", "pndf_load_weights", "pndf_forward",
           "pndf_forward_grad", "pndf_project", "pndf_experiment_word",
           "pndf_packed_sizes", "pndf_pack_host", "pndf_pack_host_split", "pndf_aa2quat", "pndf_denoise_update", "pndf_denoise_update_body", "pndf_denoise_update_w", "pndf_lbs_terms_grad_w", "pndf_quat_topk",
           "pndf_lbs_create", "pndf_lbs_destroy", "pndf_lbs_set_precision", "pndf_lbs_precision", "pndf_lbs_num_joints", "pndf_lbs_num_vertices", "pndf_lbs_workspace_floats",
           "pndf_lbs_forward", "pndf_lbs_terms_grad", "pndf_lbs_backward", "pndf_lbs_packed_floats", "pndf_lbs_pack_host", "pndf_lbs_packed_split_bytes", "pndf_lbs_pack_split_host",
           "pndf_lbs_last_error", "pndf_last_error", "pndf_version", "pndf_kernel_name",
           "pndf_cpu_create", "pndf_cpu_destroy", "pndf_cpu_load_weights", "pndf_forward_cpu", "pndf_forward_grad_cpu", "pndf_project_cpu",
           "pndf_cpu_last_error")


def state_dict_order(encoder: bool = True, n_lin: int = 7):
    """Keys in the order pndf_load_weights expects (== reference state_dict order): the 84 encoder tensors (with the
    structure encoder; without it -- model.StrEnc.use = False, in_dim 84 -- none), then weight and bias of every
    dfnet.lin{l}: 98 / 14 tensors for configs/amass.yaml's seven linear layers."""
    from .synth import DFNET_DIMS, DFNET_DIMS_NOENC, state_dict_shapes
    first = (DFNET_DIMS if encoder else DFNET_DIMS_NOENC)[0]
    return list(state_dict_shapes((first,) + (1,) * n_lin).keys())


def _tensor_table(sd_np):
    n_lin = sum(1 for k in sd_np if k.startswith("dfnet.lin") and k.endswith(".weight"))
    keys = state_dict_order(encoder=any(k.startswith("enc.") for k in sd_np), n_lin=n_lin)
    arrs = [np.ascontiguousarray(np.asarray(sd_np[k], dtype=np.float32)) for k in keys]
    ptrs = (c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    numel = (c_int64 * len(arrs))(*[a.size for a in arrs])
    return arrs, ptrs, numel


def pack_host(sd_np, lib=None, split=False):
    """Host-only packing (no device): returns (stream[STEP_TILES*256], bias) float32 arrays; split=True gives
    the split-precision stream (trunk tiles = fp16 hi/lo pairs, view them with .view(np.float16))."""
    lib = lib or load_library()
    n = [c_int64(), c_int64()]
    lib.pndf_packed_sizes(*[ctypes.byref(x) for x in n])
    stream = np.empty(n[0].value, np.float32)
    bias = np.empty(n[1].value, np.float32)
    arrs, ptrs, numel = _tensor_table(sd_np)
    fn = lib.pndf_pack_host_split if split else lib.pndf_pack_host
    rc = fn(ptrs, numel, len(arrs), stream.ctypes.data, bias.ctypes.data)
    if rc != 0:
        raise PndfError(f"pndf_pack_host failed ({rc})")
    return stream, bias


def _set_encoder_act(cfg, act, beta, enc_act, enc_beta):
    """model.StrEnc.act / beta when they differ from model.DFNet's (reference net_modules.py:128 reads its own keys; every config
    of the reference sets them equal): a mixed pair runs on the runtime-planned kernels."""
    if enc_act is not None and enc_act not in ACT_CODES:
        raise PndfError(f"unknown encoder activation {enc_act!r}")
    if enc_act is not None and (enc_act != act or (enc_act == "softplus" and enc_beta is not None and float(enc_beta) != float(beta))):
        cfg.enc_act = ACT_CODES[enc_act]
        cfg.enc_beta = float(enc_beta if enc_beta is not None else beta)


class Engine:
    """One engine per device.  All compute methods take raw device pointers and a stream handle."""

    def __init__(self, act: str = "lrelu", beta: float = 100.0, device: int = 0, lib=None, precision: str = "fp32",
                 encoder: bool = True, hidden=None, enc_act: str | None = None, enc_beta: float | None = None):
        self.lib = lib or load_library()
        if act not in ACT_CODES:
            raise PndfError(f"unknown activation {act!r}")
        if precision not in PRECISION_CODES:
            raise PndfError(f"unknown precision {precision!r} (fp32, f16x3, f16, bf16)")
        cfg = PndfConfig()
        self.lib.pndf_default_config(ctypes.byref(cfg), ACT_CODES[act], float(beta))
        cfg.precision = PRECISION_CODES[precision]
        _set_encoder_act(cfg, act, beta, enc_act, enc_beta)
        if not encoder:
            cfg.dims[0] = 84          # model.StrEnc.use = False: DFNet on the 21 x 4 normalised quaternions
        if hidden is not None:
            # model.DFNet.dims (reference net_modules.py:14-28: a free list).  Six hidden widths within configs/amass.yaml's run on
            # the fused kernels (narrower ones zero padded); any other list of 1 .. 7 widths up to 1024 on the runtime-planned
            # kernels (csrc/pndf_generic.hip: exact fp32, or split-precision fp16 MFMAs for f16x3 / f16); pndf_create refuses the rest
            hidden = [int(w) for w in hidden]
            if not 1 <= len(hidden) <= 7:
                raise PndfError(f"DFNet with {len(hidden)} hidden layers: 1 .. 7 are implemented")
            cfg.n_dims = len(hidden) + 2
            for i in range(1, len(cfg.dims)):
                cfg.dims[i] = 0
            for i, w in enumerate(hidden):
                cfg.dims[i + 1] = w
            cfg.dims[len(hidden) + 1] = 1
        self.precision = precision
        self.handle = c_void_p()
        rc = self.lib.pndf_create(ctypes.byref(self.handle), ctypes.byref(cfg), int(device))
        if rc != 0:
            msg = self.lib.pndf_last_error(None).decode()
            self.handle = None
            raise PndfError(f"pndf_create failed ({rc}): {msg}")
        self.device = device
        self.act = act

    def _check(self, rc, what):
        if rc != 0:
            raise PndfError(f"{what} failed ({rc}): {self.lib.pndf_last_error(self.handle).decode()}")

    def load_weights(self, sd_np):
        arrs, ptrs, numel = _tensor_table(sd_np)
        self._check(self.lib.pndf_load_weights(self.handle, ptrs, numel, len(arrs)), "pndf_load_weights")

    def kernel_name(self) -> str:
        """the device kernel this engine's compute calls launch (after load_weights)"""
        return self.lib.pndf_kernel_name(self.handle).decode()

    def forward(self, q_ptr, d_ptr, B, stream=0):
        self._check(self.lib.pndf_forward(self.handle, q_ptr, d_ptr, B, stream), "pndf_forward")

    def forward_grad(self, q_ptr, gout_ptr, d_ptr, dq_ptr, B, stream=0):
        self._check(self.lib.pndf_forward_grad(self.handle, q_ptr, gout_ptr, d_ptr, dq_ptr, B, stream),
                    "pndf_forward_grad")

    def project(self, q_in_ptr, q_out_ptr, d_ptr, B, steps, stream=0):
        self._check(self.lib.pndf_project(self.handle, q_in_ptr, q_out_ptr, d_ptr, B, int(steps), stream),
                    "pndf_project")

    def debug_forward_grad(self, q_ptr, d_ptr, dq_ptr, B, dump_ptr, stream=0):
        self._check(self.lib.pndf_debug_forward_grad(self.handle, q_ptr, d_ptr, dq_ptr, B, dump_ptr, stream),
                    "pndf_debug_forward_grad")

    def debug_floats(self):
        return int(self.lib.pndf_debug_floats())

    REGION_NAMES = ("enc fwd + x0", "P1 lin0,lin1", "act x2", "P2 lin2,lin3", "act x4", "P3 lin4,lin5", "act x6+lin6+g6",
                    "P4 lin5T,lin4T +dact", "P5 lin3T,lin2T +dact", "P6 lin1T,lin0T", "enc bwd", "norm bwd+update+sync")

    def project_timing(self, q, steps=3, out=None):
        """pndf_debug_project_timing on a CUDA tensor of poses: the instrumented kernel's per-region shader cycles of one
        step (mean over waves), the effective shader clock, and the weight ring's sampled events -- how long a wave sits in
        the ring's counted wait (the slot's DMA had not landed) and in its barrier, per slot."""
        import torch
        B = int(q.shape[0])
        R = int(self.lib.pndf_debug_timing_regions())
        nreg, ngrp, nring, period, slots = (int(self.lib.pndf_debug_timing_layout(i)) for i in range(5))
        cyc = torch.zeros((-(-B // 64)) * 4 * R, dtype=torch.int64, device=q.device)
        out = torch.empty_like(q) if out is None else out
        st = torch.cuda.current_stream(q.device).cuda_stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self._check(self.lib.pndf_debug_project_timing(self.handle, q.data_ptr(), out.data_ptr(), B, int(steps), cyc.data_ptr(), st),
                    "pndf_debug_project_timing")
        e1.record()
        torch.cuda.synchronize(q.device)
        ms = e0.elapsed_time(e1)
        rows = cyc.cpu().numpy().reshape(-1, R).astype(np.float64)
        reg = rows[:, :nreg].copy()
        reg[:, 3] += rows[:, nreg:nreg + ngrp].sum(1)       # per-group stamps (if built in) take their time out of region 3
        per_step = reg.mean(0) / steps
        ring = rows[:, nreg + ngrp:nreg + ngrp + nring]
        n = max(ring[:, 2].sum(), 1.0)
        cus = torch.cuda.get_device_properties(q.device).multi_processor_count
        rounds = -(-(-(-B // 64)) // cus)
        total = float(per_step.sum())
        return {"steps": int(steps), "launch_ms": ms, "workgroup_rounds": rounds,
                "cycles_per_wave_step": total,
                "effective_sclk_ghz": reg.sum(1).mean() * rounds / (ms * 1e-3) / 1e9,
                "regions": {name: float(c) for name, c in zip(self.REGION_NAMES, per_step)},
                "ring": {"look_ahead_slots": slots - 1, "sampled_every": period, "sampled_slots_per_wave": float(ring[:, 2].mean()),
                         "wait_cycles_per_slot": float(ring[:, 0].sum() / n), "barrier_cycles_per_slot": float(ring[:, 1].sum() / n),
                         "stamp_floor_cycles": float(ring[:, 3].mean()),
                         "wait_cycles_per_slot_p99_wave": float(np.percentile(ring[:, 0] / np.maximum(ring[:, 2], 1), 99))},
                "spread_over_waves": [float(reg.sum(1).min() / steps), float(reg.sum(1).max() / steps)]}

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pndf_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CpuEngine:
    """The host twins `pndf_*_cpu` behind the interface of `Engine` (raw HOST pointers; `stream` is accepted and ignored):
    what `PoseNDF` runs on when its config says `train.device: cpu`, as the reference's class does (model/posendf.py:35,64).
    Plain C++ on the host cores -- not the oracle, and never a fallback of the device engine."""

    def __init__(self, act: str = "lrelu", beta: float = 100.0, lib=None, encoder: bool = True, hidden=None,
                 enc_act: str | None = None, enc_beta: float | None = None):
        self.lib = lib or load_library()
        if act not in ACT_CODES:
            raise PndfError(f"unknown activation {act!r}")
        cfg = PndfConfig()
        self.lib.pndf_default_config(ctypes.byref(cfg), ACT_CODES[act], float(beta))
        _set_encoder_act(cfg, act, beta, enc_act, enc_beta)
        if not encoder:
            cfg.dims[0] = 84
        if hidden is not None:
            hidden = [int(w) for w in hidden]
            if not 1 <= len(hidden) <= 7:
                raise PndfError(f"DFNet with {len(hidden)} hidden layers: 1 .. 7 are implemented")
            cfg.n_dims = len(hidden) + 2
            for i in range(1, len(cfg.dims)):
                cfg.dims[i] = 0
            for i, w in enumerate(hidden):
                cfg.dims[i + 1] = w
            cfg.dims[len(hidden) + 1] = 1
        self.precision = "fp32"
        self.handle = c_void_p()
        rc = self.lib.pndf_cpu_create(ctypes.byref(self.handle), ctypes.byref(cfg))
        if rc != 0:
            msg = self.lib.pndf_cpu_last_error(None).decode()
            self.handle = None
            raise PndfError(f"pndf_cpu_create failed ({rc}): {msg}")
        self.device = "cpu"
        self.act = act

    def _check(self, rc, what):
        if rc != 0:
            raise PndfError(f"{what} failed ({rc}): {self.lib.pndf_cpu_last_error(self.handle).decode()}")

    def load_weights(self, sd_np):
        arrs, ptrs, numel = _tensor_table(sd_np)
        self._check(self.lib.pndf_cpu_load_weights(self.handle, ptrs, numel, len(arrs)), "pndf_cpu_load_weights")

    def kernel_name(self) -> str:
        return "pndf_cpu (host twin)"

    def forward(self, q_ptr, d_ptr, B, stream=0):
        self._check(self.lib.pndf_forward_cpu(self.handle, q_ptr, d_ptr, B), "pndf_forward_cpu")

    def forward_grad(self, q_ptr, gout_ptr, d_ptr, dq_ptr, B, stream=0):
        self._check(self.lib.pndf_forward_grad_cpu(self.handle, q_ptr, gout_ptr, d_ptr, dq_ptr, B), "pndf_forward_grad_cpu")

    def project(self, q_in_ptr, q_out_ptr, d_ptr, B, steps, stream=0):
        self._check(self.lib.pndf_project_cpu(self.handle, q_in_ptr, q_out_ptr, d_ptr, B, int(steps)), "pndf_project_cpu")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pndf_cpu_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
