"""ORACLE (test infrastructure, NOT product code) -- PyTorch-CPU restatement of the reference's hot path,
used ONLY as the timed `cpu_baseline` of bench.py ("port": the reference's own Python cannot travel to the
GPU box) and cross-checked against the golden vectors in tests/test_oracle_torch.py.

Same op sequence as the reference: nn.Linear / LeakyReLU|ReLU|Softplus / cat / F.normalize(dim=1) /
torch.autograd.grad, so that its wall time is representative of "the reference's CPU PyTorch path"
(model/posendf.py:62-76, model/network/net_modules.py:46-72,140-170, experiments/sample_poses.py:67-74).
Parity status: PINNED (tests/golden, see oracle/posendf_np.py header).
"""
from __future__ import annotations

import torch
import torch.nn as nn

PARENT = (-1, -1, -1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19)   # net_utils.py:46


def _act(kind, beta, out=False):
    if kind == "softplus":
        return nn.Softplus(beta=beta)
    if kind == "relu" or out:
        return nn.ReLU()
    return nn.LeakyReLU()


class RefNet(nn.Module):
    """enc.net[i].net[0|2] / dfnet.lin{l}: the reference's parameter tree, flattened into one module."""

    def __init__(self, act="lrelu", beta=100.0, dims=(126, 256, 512, 1024, 512, 256, 64, 1)):
        super().__init__()
        enc = nn.Module()
        enc.net = nn.ModuleList()
        for p in PARENT:
            m = nn.Module()
            m.net = nn.Sequential(nn.Linear(4 if p < 0 else 10, 10), _act(act, beta), nn.Linear(10, 6), _act(act, beta))
            enc.net.append(m)
        self.enc = enc
        df = nn.Module()
        for l in range(len(dims) - 1):
            setattr(df, f"lin{l}", nn.Linear(dims[l], dims[l + 1]))
        self.dfnet = df
        self.nl = len(dims) - 1
        self.actv, self.out_actv = _act(act, beta), _act(act, beta, out=True)

    def forward(self, pose):
        x = torch.nn.functional.normalize(pose.reshape(-1, 21, 4), dim=1)      # posendf.py:71
        feats = [None] * 21
        for i, p in enumerate(PARENT):                                           # net_modules.py:162-168
            inp = x[:, i, :] if p < 0 else torch.cat((x[:, i, :], feats[p]), dim=-1)
            feats[i] = self.enc.net[i].net(inp)
        h = torch.cat(feats, dim=-1)
        for l in range(self.nl):                                                 # net_modules.py:51-69
            h = getattr(self.dfnet, f"lin{l}")(h)
            h = self.actv(h) if l < self.nl - 1 else self.out_actv(h)
        return h


def project(net, q, steps):
    """experiments/sample_poses.py:67-74 (detached between steps: values identical)."""
    d = None
    for _ in range(steps):
        q = q.detach().requires_grad_(True)
        d = net(q)
        (g,) = torch.autograd.grad(d, q, grad_outputs=torch.ones_like(d))
        q = q - (d * g.reshape(-1, 84)).reshape(-1, 21, 4)
    return q.detach(), d.detach()
