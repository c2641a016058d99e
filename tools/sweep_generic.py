#!/usr/bin/env python3
"""Randomised architectures through the runtime-planned kernels (csrc/pndf_generic.hip), both forms (precision fp32 / f16x3), against the
numpy oracle with the gates of tests/test_depth.py's extremes test: depth 1 .. 7, widths 1 .. 1024 (biased towards tile / group / pass
edges), the three activations, with and without the encoder.  usage: python tools/sweep_generic.py [n_networks] [seed]
Prints one line per network and a summary; exit code 1 on any failure."""
import os
import sys
import traceback

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import test_depth as td  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
EDGES = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 384, 385, 511, 512, 513, 640, 767, 768, 896, 1000, 1023, 1024]
fails = 0
for i in range(n):
    depth = int(rng.integers(1, 8))
    hidden = [int(rng.choice(EDGES)) if rng.random() < 0.6 else int(rng.integers(1, 1025)) for _ in range(depth)]
    if depth == 6 and all(w <= a for w, a in zip(hidden, (256, 512, 1024, 512, 256, 64))):
        hidden[2] = 1024 + 0 * hidden[2] if hidden[0] > 256 else hidden[2]
        hidden[0] = 257                     # (six hidden widths within amass.yaml's run on the FUSED kernels: not this sweep's subject)
    act = str(rng.choice(["lrelu", "relu", "softplus"]))
    enc = bool(rng.random() < 0.7)
    for precision in ("fp32", "f16x3"):
        try:
            td.live_weights((126 if enc else 84, *hidden, 1), act)
        except AssertionError:
            print(f"[{i}] {hidden} {act} enc={enc}: no live weight set (dead network), skipped", flush=True)
            break
        try:
            td.test_runtime_planned_kernels_width_and_depth_extremes.__wrapped__(hidden, act, enc, precision) \
                if hasattr(td.test_runtime_planned_kernels_width_and_depth_extremes, "__wrapped__") \
                else td.test_runtime_planned_kernels_width_and_depth_extremes(hidden, act, enc, precision)
            print(f"[{i}] {hidden} {act} enc={enc} {precision}: ok", flush=True)
        except Exception as exc:      # noqa: BLE001
            fails += 1
            print(f"[{i}] {hidden} {act} enc={enc} {precision}: FAIL {type(exc).__name__}: {str(exc)[:300]}", flush=True)
            traceback.print_exc(limit=2)
print(f"sweep_generic: {fails} failure(s)")
sys.exit(1 if fails else 0)
