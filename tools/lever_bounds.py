#!/usr/bin/env python3
"""Upper bounds of kernel levers, measured on the PRODUCT kernel with valid operand data (same box, one process per arm):
    * encoder: the same network launched encoder-less (model.StrEnc.use = False: the trunk is identical -- lin0's K is padded
      to 128 either way -- and the stream's encoder sections are still walked slot by slot) against the full one: everything
      the 21 BoneMLPs cost, i.e. what ANY encoder rewrite could at most recover;
    * per-launch fixed cost: steps = 1 and 2 against 100.
usage: python tools/lever_bounds.py [act] [precision]"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, sys, torch, numpy as np
sys.path.insert(0, %(repo)r)
from posendf_amd import PoseNDF, amass_config, synth
act, prec, enc = %(act)r, %(prec)r, %(enc)d
cfg = amass_config(act, "cuda:0"); cfg["engine"] = {"precision": prec}
dims = synth.DFNET_DIMS
if not enc:
    cfg["model"]["StrEnc"]["use"] = False; cfg["model"]["DFNet"]["in_dim"] = 84; dims = synth.DFNET_DIMS_NOENC
net = PoseNDF(cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(0, 2.0, 0.1, dims=dims).items()}); net.eval()
q = torch.from_numpy(synth.make_poses(65536, seed=1234)).cuda()
net.project(q, steps=100); torch.cuda.synchronize()
ms = []
for _ in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out, d = net.project(q, steps=100); e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
print(json.dumps({"ms": ms, "dmean": float(d.mean().item()), "finite": bool(torch.isfinite(out).all())}))
"""


def main():
    act = sys.argv[1] if len(sys.argv) > 1 else "lrelu"
    prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
    res = {1: [], 0: []}
    for rnd in range(2):
        for enc in (1, 0):
            p = subprocess.run([sys.executable, "-c", CHILD % dict(repo=REPO, act=act, prec=prec, enc=enc)], capture_output=True,
                               text=True, timeout=600)
            if p.returncode != 0:
                print("arm", enc, "FAILED", p.stderr[-600:])
                continue
            out = json.loads(p.stdout.strip().splitlines()[-1])
            res[enc] += out["ms"]
            print(f"round {rnd} encoder={enc}: {['%.2f' % m for m in out['ms']]} ms, mean d {out['dmean']:.5f}, finite {out['finite']}", flush=True)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items() if v}
    if len(med) == 2:
        print(f"[{act} {prec}] with encoder {med[1]:.2f} ms, encoder-less {med[0]:.2f} ms: the encoder costs {med[1] - med[0]:.2f} ms = "
              f"{(med[1] - med[0]) / med[1] * 100:.1f} % of a launch (upper bound of any encoder rewrite)")


if __name__ == "__main__":
    main()
