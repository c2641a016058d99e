#!/usr/bin/env python3
"""Per-region shader-cycle breakdown of one projection step (s_memtime-instrumented kernel), vs the ideal
MFMA issue time of each region (32 cycles per v_mfma_f32_16x16x4_f32)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from posendf_amd import PoseNDF, amass_config, synth  # noqa: E402

NAMES = ["enc fwd + x0", "P1 lin0,lin1", "act x2", "P2 lin2,lin3", "act x4", "P3 lin4,lin5", "act x6+lin6+g6",
         "P4 lin5T,lin4T +dact", "P5 lin3T,lin2T +dact", "P6 lin1T,lin0T", "enc bwd", "norm bwd+update+sync"]
TILES = [0, 640, 0, 4096, 0, 576, 0, 576, 4096, 640, 0, 0]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
    act = sys.argv[3] if len(sys.argv) > 3 else "lrelu"
    B = 65536 if act != "softplus" else 16384       # softplus: persistent grid, the stamps are per workgroup (one block each)
    cfg = amass_config(act, "cuda:0")
    cfg["engine"] = {"precision": prec}
    net = PoseNDF(cfg)
    cyc_per_tile = {"fp32": 128, "f16x3": 24, "f16": 8}[prec]   # per stream tile: 4 MFMAs x 32 cycles; 1.5 x 16; 0.5 x 16
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(0, 2.0, 0.1).items()})
    q = torch.from_numpy(synth.make_poses(B, seed=1)).cuda()
    eng = net._engine_for(q.device)
    R = eng.lib.pndf_debug_timing_regions()
    cyc = torch.zeros((B // 64) * 4 * R, dtype=torch.int64, device="cuda")
    out = torch.empty_like(q)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = eng.lib.pndf_debug_project_timing(eng.handle, q.data_ptr(), out.data_ptr(), B, steps, cyc.data_ptr(), 0)
    e1.record()
    assert rc == 0
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    rounds = -(-(B // 64) // 256)
    print(f"instrumented launch {ms:.2f} ms for {steps} steps, {rounds} workgroup rounds per CU -> effective shader clock "
          f"{cyc.cpu().numpy().reshape(-1, R).sum(1).mean() * rounds / (ms * 1e-3) / 1e9:.2f} GHz")
    call = cyc.cpu().numpy().reshape(-1, R).astype(np.float64) / steps
    c = call[:, :12].copy()
    if prec in ("fp32", "f16x3"):
        c[:, 3] += call[:, 12:].sum(1)      # the per-group stamps of the (lin2,lin3) phase take their time out of region 3
    mean = c.mean(0)
    tot = mean.sum()
    print(f"[{prec} {act}] per wave-step: total {tot:,.0f} shader cycles; ideal MFMA {sum(TILES) * cyc_per_tile:,} ({sum(TILES) * cyc_per_tile / tot * 100:.1f} %)")
    for n, m, t in zip(NAMES, mean, TILES):
        ideal = t * cyc_per_tile
        extra = f"  ideal {ideal:9,d}  eff {ideal / m * 100:5.1f} %  over {m - ideal:9,.0f}" if t else f"  {'':40s}"
        print(f"{n:24s} {m:11,.0f} cyc {m / tot * 100:5.1f} %{extra}")
    grp = call[:, 12:].mean(0)
    if prec == "fp32":
        grp = grp / 32                       # per chunk of the (lin2,lin3) phase; ideal = 16 MFMAs = 512 cycles
        print("per-group cycles inside one (lin2,lin3) chunk (ideal 512; includes the s_memtime stamp itself):")
        print("  part A:", " ".join(f"{v:5.0f}" for v in grp[:16]))
        print("  part B:", " ".join(f"{v:5.0f}" for v in grp[16:]))
    if prec == "f16x3" and grp.any():       # only in a build with -DPNDF_GROUP_STAMPS=1 (the stamps distort the loop)
        grp = grp / 31                       # per chunk of the (lin2,lin3) loop (31 iterations; ideal = 12 MFMAs = 192 cycles)
        print("per-group cycles inside one (lin2,lin3) chunk (ideal 192; includes the s_memtime stamp itself):")
        print("  part A(c+1):", " ".join(f"{v:5.0f}" for v in grp[:8]))
        print("  part B(c)  :", " ".join(f"{v:5.0f}" for v in grp[8:16]))
    print("spread over waves (min/max of total):", c.sum(1).min(), c.sum(1).max())
    # the weight ring's two synchronous events, sampled (pndf_device.h: ring_midslot_sync, PNDF_RING_STAMPS)
    nreg, ngrp, nring, period, slots = (int(eng.lib.pndf_debug_timing_layout(i)) for i in range(5))
    ring = cyc.cpu().numpy().reshape(-1, R).astype(np.float64)[:, nreg + ngrp:nreg + ngrp + nring]
    n = max(ring[:, 2].sum(), 1.0)
    per_wave = ring[:, 0] / np.maximum(ring[:, 2], 1)
    print(f"ring (look-ahead {slots - 1} slots; every {period}th slot sampled, {ring[:, 2].mean():.0f} samples per wave): counted vmcnt wait "
          f"{ring[:, 0].sum() / n:6.1f} cycles per slot (two stamps back to back: {ring[:, 3].mean():.0f}), barrier {ring[:, 1].sum() / n:6.1f} "
          f"cycles per slot; per-wave wait p50 / p99 / max {np.percentile(per_wave, 50):.0f} / {np.percentile(per_wave, 99):.0f} / {per_wave.max():.0f}")
    print(f"  -> of the {tot:,.0f} cycles of a step, {670 * (ring[:, 0].sum() / n - ring[:, 3].mean()):,.0f} are spent waiting for a slot's DMA "
          f"and {670 * ring[:, 1].sum() / n:,.0f} in the ring's barrier (670 slots per step)")


if __name__ == "__main__":
    main()
