#!/usr/bin/env python3
"""Bring-up diagnostic (GPU box): run the debug variant of the fused kernel on 64 poses and compare every
dumped stage (encoder features, x2, x4, x6, d, g4, g2, g0, d d/d n, d d/d q) with the numpy oracle, then
time forward / forward_grad / project at a few sizes.  Prints a compact table; exit code 1 on mismatch."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import posendf_np as onp            # noqa: E402  (diagnostic tool = test infrastructure)
from posendf_amd import PoseNDF, amass_config, synth  # noqa: E402

DBG = dict(FEAT=(0, 126), X2=(126, 128), X4=(254, 128), X6=(382, 16), D=(398, 1), G4=(399, 128), G2=(527, 128),
           G0=(655, 32), GN=(687, 84), DQ=(771, 84))


def decode_tiles(raw, ntiles):
    """raw [ntiles*4, 256] (reg-major, thread-minor) -> [64 poses, 16*ntiles]: lane (g,p) reg (t,r) holds
    row 16 t + 4 g + r of pose wave*16 + p."""
    out = np.zeros((64, 16 * ntiles), np.float32)
    tid = np.arange(256)
    wave, lane = tid >> 6, tid & 63
    g, p = lane >> 4, lane & 15
    for t in range(ntiles):
        for r in range(4):
            out[wave * 16 + p, 16 * t + 4 * g + r] = raw[4 * t + r]
    return out


def per_pose(raw, n):
    """raw [n, 256] values replicated over the 4 lane groups -> [64, n] (taken from lane group 0) and the
    max disagreement between lane groups."""
    tid = np.arange(256)
    wave, lane = tid >> 6, tid & 63
    g, p = lane >> 4, lane & 15
    out = np.zeros((64, n), np.float32)
    sel = g == 0
    out[(wave * 16 + p)[sel]] = raw[:, sel].T
    spread = 0.0
    for gg in range(1, 4):
        s2 = g == gg
        o2 = np.zeros_like(out)
        o2[(wave * 16 + p)[s2]] = raw[:, s2].T
        spread = max(spread, float(np.abs(o2 - out).max()))
    return out, spread


def err(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def main():
    act = sys.argv[1] if len(sys.argv) > 1 else "lrelu"
    sd = synth.make_weights(0, 2.5, 0.05)
    cfg = amass_config(act, "cuda:0")
    cfg["engine"] = {"precision": "fp32"}          # the stage-dump kernel exists for the exact fp32 arithmetic
    net = PoseNDF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    q_np = synth.make_poses(64, seed=7, signed=True)
    q = torch.from_numpy(q_np).cuda()
    eng = net._engine_for(q.device)
    nd = eng.debug_floats()
    dump = torch.full((nd,), float("nan"), device="cuda")
    d = torch.empty(64, device="cuda")
    dq = torch.empty(64, 84, device="cuda")
    eng.debug_forward_grad(q.data_ptr(), d.data_ptr(), dq.data_ptr(), 64, dump.data_ptr(), 0)
    torch.cuda.synchronize()
    raw = dump.cpu().numpy().reshape(-1, 256)
    dbg = {}
    d_o, dq_o = onp.forward_grad(q_np, sd, act, debug=dbg)
    A = lambda z: onp._act(z, act, 100.0)
    gx = dbg["gx"]           # gx[l] = dL/dx_l (input of lin l)
    dz = lambda l: onp._dact(dbg["zs"][l], act, 100.0)
    rows = []
    feat, sp = per_pose(raw[0:126], 126)
    rows.append(("features", err(feat, dbg["feat"]), sp))
    rows.append(("x2", err(decode_tiles(raw[126:254], 32), A(dbg["zs"][1])), 0))
    rows.append(("x4", err(decode_tiles(raw[254:382], 32), A(dbg["zs"][3])), 0))
    rows.append(("x6", err(decode_tiles(raw[382:398], 4), A(dbg["zs"][5])), 0))
    dd, sp = per_pose(raw[398:399], 1)
    rows.append(("d(dump)", err(dd, d_o), sp))
    rows.append(("g4*m4", err(decode_tiles(raw[399:527], 32), gx[4] * dz(3)), 0))
    rows.append(("g2*m2", err(decode_tiles(raw[527:655], 32), gx[2] * dz(1)), 0))
    rows.append(("g0", err(decode_tiles(raw[655:687], 8)[:, :126], gx[0]), 0))
    gn, sp = per_pose(raw[687:771], 84)
    rows.append(("dd/dn", err(gn, dbg["gn"].reshape(64, 84)), sp))
    rows.append(("d(out)", err(d.cpu().numpy()[:, None], d_o), 0))
    rows.append(("dd/dq(out)", err(dq.cpu().numpy(), dq_o.reshape(64, 84)), 0))
    bad = False
    for name, e, s in rows:
        flag = "" if (e < 1e-4 and s == 0) else "   <-- MISMATCH"
        bad |= bool(flag)
        print(f"{name:12s} rel-err {e:9.2e}  lane-group spread {s:9.2e}{flag}")
    # production kernel, several sizes
    for B in (1, 63, 64, 65, 200, 4096):
        qn = synth.make_poses(B, seed=B, signed=True)
        qt = torch.from_numpy(qn).cuda().requires_grad_(True)
        dd = net(qt, train=False)["dist_pred"]
        (gg,) = torch.autograd.grad(dd.sum(), qt)
        do, go = onp.forward_grad(qn, sd, act)
        e1 = err(dd.detach().cpu().numpy(), do)
        e2 = float(np.median(np.abs(gg.cpu().numpy() - go).reshape(B, -1).max(1) / np.abs(go).reshape(B, -1).max(1).clip(1e-30)))
        qp, _ = net.project(torch.from_numpy(qn), steps=5)
        q64, _ = onp.project(qn, sd, steps=5, act=act, dtype=np.float64)      # truth trajectory
        q32, _ = onp.project(qn, sd, steps=5, act=act)                         # the reference's arithmetic
        rows_e = lambda a: np.abs(a.reshape(B, -1) - q64.reshape(B, -1)).max(1) / np.abs(q64.reshape(B, -1)).max(1)
        mine, ref = rows_e(qp.cpu().numpy().astype(np.float64)), rows_e(q32.astype(np.float64))
        # kinks (ReLU / LeakyReLU sign flips within rounding) make a few poses diverge for ANY fp32
        # implementation; gate on the median and on the fraction of diverged poses vs the oracle's own
        ok = e1 < 1e-4 and np.median(mine) < 1e-5 and (mine > 1e-4).mean() <= 2 * (ref > 1e-4).mean() + max(0.003, 2.0 / B)
        flag = "" if ok else "   <-- MISMATCH"
        bad |= bool(flag)
        print(f"B={B:5d}  d {e1:8.2e}  grad(median) {e2:8.2e}  project5 vs fp64: median {np.median(mine):8.2e} "
              f"frac>1e-4 {(mine > 1e-4).mean():.4f} (oracle fp32: {(ref > 1e-4).mean():.4f}) max {mine.max():8.2e}{flag}")
    # timing
    for B, steps in ((65536, 1), (65536, 10), (65536, 100)):
        qt = torch.from_numpy(synth.make_poses(B, seed=1)).cuda()
        net.project(qt, steps=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        net.project(qt, steps=steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tf = B * steps * 5450416 / dt / 1e12
        print(f"project B={B} steps={steps}: {dt * 1e3:9.2f} ms  {B / dt:12.0f} poses/s  {tf:7.2f} TFLOP/s "
              f"({tf / 157.3 * 100:5.1f}% of fp32 MFMA peak)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
