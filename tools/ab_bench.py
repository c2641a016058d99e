#!/usr/bin/env python3
"""Same-box A/B of library builds: every library is timed in its own process (PNDF_LIBRARY), round-robin, several
rounds, so that the box's clock drift hits all arms alike.  Usage: python tools/ab_bench.py [--act lrelu] [--precision
f16x3] [--rounds 3] name=path ...   Prints per-arm kernel ms (HIP events) and the effective poses/s."""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, torch
sys.path.insert(0, %(repo)r)
from posendf_amd import PoseNDF, amass_config, synth
act, prec, B, steps, reps = %(act)r, %(prec)r, %(B)d, %(steps)d, %(reps)d
cfg = amass_config(act, "cuda:0"); cfg["engine"] = {"precision": prec}
net = PoseNDF(cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(0, 2.0, 0.1).items()}); net.eval()
q = torch.from_numpy(synth.make_poses(B, seed=1234)).cuda()
net.project(q, steps=steps); torch.cuda.synchronize()
ms = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out, d = net.project(q, steps=steps); e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
print(json.dumps({"ms": ms, "checksum": float(out.double().sum().item()), "dmean": float(d.mean().item())}))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--act", default="lrelu")
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("arms", nargs="+")
    a = ap.parse_args()
    arms = [x.partition("=")[::2] for x in a.arms]
    res = {n: [] for n, _ in arms}
    chk = {}
    for r in range(a.rounds):
        for name, path in arms:
            env = dict(os.environ, PNDF_LIBRARY=os.path.abspath(path))
            code = CHILD % dict(repo=REPO, act=a.act, prec=a.precision, B=a.batch, steps=a.steps, reps=a.reps)
            p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
            if p.returncode != 0:
                print(name, "FAILED", p.stderr[-800:])
                continue
            out = json.loads(p.stdout.strip().splitlines()[-1])
            res[name] += out["ms"]
            chk[name] = (out["checksum"], out["dmean"])
    for name, _ in arms:
        ms = sorted(res[name])
        if not ms:
            continue
        med = ms[len(ms) // 2]
        print(f"{name:12s} act={a.act} prec={a.precision} median {med:8.2f} ms  min {ms[0]:8.2f}  max {ms[-1]:8.2f}  "
              f"-> {a.batch / med * 1e3:10.0f} poses/s  checksum {chk[name][0]:.9g} dmean {chk[name][1]:.6g}", flush=True)


if __name__ == "__main__":
    main()
