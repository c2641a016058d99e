#!/usr/bin/env python3
"""The runtime-planned kernels (csrc/pndf_generic.hip) on one GPU: project(B = 65,536, 10 steps) for the depth / width cases of
tests/golden/make_golden_depth.py, and for configs/amass.yaml itself against the fused exact-fp32 kernel (PNDF_FORCE_GENERIC=1 in a
child process), exact fp32 form and split-precision form (precision f16x3: arms 10 ..).  Prints one JSON line per arm: kernel, ms,
algorithmic TFLOP/s (4 x sum in x out FLOP per pose-step), fraction of the fp32 (fp16 for the split arms) MFMA peak.
usage: python tools/bench_generic.py [arm ...] > profiles/r06/generic_arch.jsonl"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
PEAK = 157.3
PEAK_F16 = 2500.0

CHILD = r"""
import json, sys, torch
sys.path.insert(0, %(repo)r)
from posendf_amd import PoseNDF, amass_config, synth
hidden, act, enc, B, steps, prec = %(hidden)r, %(act)r, %(enc)r, %(B)d, %(steps)d, %(prec)r
cfg = amass_config(act, "cuda:0"); cfg["engine"] = {"precision": prec}
cfg["model"]["DFNet"]["dims"] = hidden; cfg["model"]["StrEnc"]["use"] = enc
if not enc: cfg["model"]["DFNet"]["in_dim"] = 84
net = PoseNDF(cfg)
dims = (126 if enc else 84, *hidden, 1)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(0, 2.0, 0.1, dims=dims).items()}); net.eval()
q = torch.from_numpy(synth.make_poses(B, seed=1234)).cuda()
net.project(q, steps=steps); torch.cuda.synchronize()
ms = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out, d = net.project(q, steps=steps); e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
flop = 4 * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
m = sorted(ms)[1]
print(json.dumps({"kernel": net._engine_for(q.device).kernel_name(), "precision": prec, "hidden": hidden, "act": act, "encoder": enc, "batch": B, "steps": steps,
                  "ms": m, "pose_steps_per_s": B * steps / (m * 1e-3), "flop_per_pose_step": flop,
                  "tflops": B * steps * flop / (m * 1e-3) / 1e12, "checksum": float(out.double().sum().item())}))
"""

AMASS = [256, 512, 1024, 512, 256, 64]
ARMS = [("amass.yaml, fused fp32 kernel", AMASS, "lrelu", True, {}),
        ("amass.yaml, runtime-planned", AMASS, "lrelu", True, {"PNDF_FORCE_GENERIC": "1"}),
        ("amass.yaml softplus, fused fp32 kernel", AMASS, "softplus", True, {}),
        ("amass.yaml softplus, runtime-planned", AMASS, "softplus", True, {"PNDF_FORCE_GENERIC": "1"}),
        ("four hidden layers", [192, 320, 160, 48], "lrelu", True, {}),
        ("seven hidden layers", [128, 256, 512, 1024, 512, 256, 64], "lrelu", True, {}),
        ("seven hidden layers, softplus", [128, 256, 512, 1024, 512, 256, 64], "softplus", True, {}),
        ("amass depth, wider", [512, 1024, 1024, 640, 256, 128], "relu", True, {}),
        ("one hidden layer", [300], "lrelu", True, {}),
        ("no encoder, three hidden layers", [200, 100, 50], "lrelu", False, {}),
        # precision f16x3: the split-precision trunk (fraction of the fp16 MFMA peak, algorithmic FLOP: three MFMAs per product block)
        ("f16x3: amass.yaml, fused split kernel", AMASS, "lrelu", True, {"prec": "f16x3"}),
        ("f16x3: amass.yaml, runtime-planned", AMASS, "lrelu", True, {"PNDF_FORCE_GENERIC": "1", "prec": "f16x3"}),
        ("f16x3: amass.yaml softplus, runtime-planned", AMASS, "softplus", True, {"PNDF_FORCE_GENERIC": "1", "prec": "f16x3"}),
        ("f16x3: seven hidden layers", [128, 256, 512, 1024, 512, 256, 64], "lrelu", True, {"prec": "f16x3"}),
        ("f16x3: amass depth, wider", [512, 1024, 1024, 640, 256, 128], "relu", True, {"prec": "f16x3"}),
        ("f16x3: four hidden layers", [192, 320, 160, 48], "lrelu", True, {"prec": "f16x3"})]

only = [int(a) for a in sys.argv[1:] if a.isdigit()]      # arm indices (default: all)
for i, (name, hidden, act, enc, env) in enumerate(ARMS):
    if only and i not in only:
        continue
    env = dict(env)
    prec = env.pop("prec", "fp32")
    code = CHILD % dict(repo=REPO, hidden=hidden, act=act, enc=enc, B=65536, steps=10, prec=prec)
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    if p.returncode != 0:
        print(json.dumps({"arm": name, "error": p.stderr[-600:]}), flush=True)
        continue
    r = json.loads(p.stdout.strip().splitlines()[-1])
    if prec == "fp32":
        print(json.dumps({"arm": name, **r, "frac_of_fp32_mfma_peak": r["tflops"] / PEAK}), flush=True)
    else:
        print(json.dumps({"arm": name, **r, "frac_of_fp16_mfma_peak": r["tflops"] / PEAK_F16}), flush=True)
