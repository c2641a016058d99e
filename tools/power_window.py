#!/usr/bin/env python3
"""Energy per launch and throttler residency of the fused kernels (bench.py: power_window -- two `amd-smi metric` readings
while the kernel runs back to back): which limiter holds the clock, per kernel.  usage: python tools/power_window.py
[precision:act ...]   default: f16x3:lrelu f16x3:softplus fp32:lrelu f16:lrelu
       python tools/power_window.py --libs name=path.so ... [precision:act]   the same arm through several library builds
       (tools/build_variants.py), one process each (PNDF_LIBRARY): the energy of what an ablation build leaves out"""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from posendf_amd import PoseNDF, amass_config, synth  # noqa: E402


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--libs":
        import subprocess
        libs = [a for a in sys.argv[2:] if "=" in a]
        arms = [a for a in sys.argv[2:] if "=" not in a] or ["f16x3:lrelu"]
        for spec in libs:
            name, _, path = spec.partition("=")
            p = subprocess.run([sys.executable, os.path.abspath(__file__), *arms], env=dict(os.environ, PNDF_LIBRARY=os.path.abspath(path)),
                               capture_output=True, text=True, timeout=600)
            for line in p.stdout.splitlines():
                if line.startswith("{"):
                    print(json.dumps({"build": name, **json.loads(line)}), flush=True)
            if p.returncode != 0:
                print(json.dumps({"build": name, "error": p.stderr[-400:]}), flush=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--ring-only":
        # the weight ring alone (csrc/pndf_probe.hip: probe_ring), no arithmetic: package power while every CU streams 11 MB passes
        import ctypes
        from posendf_amd import engine
        lib = engine.load_library()
        torch.zeros(1, device="cuda")
        sec = ctypes.c_double()
        passes = 1000
        assert lib.pndf_debug_ring_stream(0, passes, ctypes.byref(sec)) == 0
        idle = bench._amdsmi_metric()
        r = bench.power_window(lambda: lib.pndf_debug_ring_stream(0, passes, ctypes.byref(sec)), lambda: None, sec.value * passes * 1e3, bdf=bench.device_bdf(0))
        r.pop("what", None)
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        bytes_per_call = passes * 670 * 16384 * cus
        print(json.dumps({"load": "ring only (probe_ring)", "sec_per_pass": sec.value, "tb_per_s_into_lds": 670 * 16384 * cus / sec.value / 1e12,
                          "idle_socket_power_w_before": idle and idle["socket_power_w"], "bytes_per_call": bytes_per_call,
                          "pj_per_byte_incl_idle_power": r.get("energy_j_per_launch", 0) / bytes_per_call * 1e12, **r}), flush=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--lbs":
        # the body-model pass (pndf_lbs_terms_grad, 512 x 300 frames) through the same window: is IT held by the power limit?
        from posendf_amd import BodyModel
        S, T = 512, 300
        m = synth.make_body_model(seed=11)
        for prec in ("f16x3", "fp32"):
            bm = BodyModel(m, device="cuda:0", extra_joint_vertex=m["extra_joint_vertex"], precision=prec)
            g = torch.Generator().manual_seed(0)
            theta = (torch.cumsum(0.02 * torch.randn(S, T, 69, generator=g), dim=1) + 0.3 * torch.randn(S, 1, 69, generator=g)).cuda()
            j0 = bm.joints_of(theta + 0.02)
            out = torch.empty_like(theta)
            for _ in range(3):
                bm.terms_grad(theta, j0, 2, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                bm.terms_grad(theta, j0, 2, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            r = bench.power_window(lambda: bm.terms_grad(theta, j0, 2, out=out), torch.cuda.synchronize, ms, bdf=bench.device_bdf(0))
            r.pop("what", None)
            print(json.dumps({"load": f"pndf_lbs_terms_grad {S} x {T} frames, {prec}", "kernel_ms": ms, **r}), flush=True)
        return
    arms = sys.argv[1:] or ["f16x3:lrelu", "f16x3:softplus", "fp32:lrelu", "f16:lrelu"]
    q = torch.from_numpy(synth.make_poses(65536, seed=1234)).cuda()
    sd = {k: torch.from_numpy(v) for k, v in synth.make_weights(0, 2.0, 0.1).items()}
    for arm in arms:
        prec, act = arm.split(":")
        half_ckpt = prec.endswith("h")                  # "f16x3h": the same network as a half-precision checkpoint (two-term kernels)
        prec = prec.rstrip("h")
        cfg = amass_config(act, "cuda:0")
        cfg["engine"] = {"precision": prec}
        net = PoseNDF(cfg)
        net.load_state_dict({k: v.half().float() for k, v in sd.items()} if half_ckpt else sd)
        net.eval()
        for _ in range(3):
            net.project(q, steps=100)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            net.project(q, steps=100)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        r = bench.power_window(lambda: net.project(q, steps=100), torch.cuda.synchronize, ms, bdf=bench.device_bdf(0))
        r.pop("what", None)
        print(json.dumps({"kernel": net._engine_for(q.device).kernel_name(), "kernel_ms": ms, **r}), flush=True)


if __name__ == "__main__":
    main()
