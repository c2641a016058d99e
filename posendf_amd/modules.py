"""Parameter containers for the Pose-NDF network (PyTorch modules).

These exist for three reasons only: (1) reference checkpoints load unchanged -- the attribute tree
`enc.net[i].net[0|2]`, `dfnet.lin{l}` reproduces the 98 state-dict keys of the reference
(model/network/net_modules.py:14-28,78-107,116-128); (2) optimisers can see `parameters()`; (3) the
train=True path (weight gradients + eikonal double backward, model/posendf.py:77-99) stays on stock
PyTorch-ROCm, which SURVEY.md section 8 marks out of scope for the HIP kernels.
The inference path (train=False) never calls these modules' forward: it goes to the HIP engine.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .synth import BONE_DIM, FEAT, PARENT


def _activation(kind: str, beta: float, output: bool = False) -> nn.Module:
    # net_modules.py:30-41 / :86-107: lrelu -> LeakyReLU() hidden + ReLU output; relu -> ReLU/ReLU;
    # softplus -> Softplus(beta) for hidden AND output.
    if kind == "softplus":
        return nn.Softplus(beta=beta)
    if kind == "relu" or output:
        return nn.ReLU()
    if kind == "lrelu":
        return nn.LeakyReLU()
    raise ValueError(f"unknown activation {kind!r}")


class BoneMLP(nn.Module):
    """4|10 -> 10 -> 6 per-joint MLP (net_modules.py:75-111); both layers use the hidden activation."""

    def __init__(self, has_parent: bool, act: str, beta: float):
        super().__init__()
        fan_in = BONE_DIM + (FEAT if has_parent else 0)
        width = BONE_DIM + FEAT
        self.net = nn.Sequential(nn.Linear(fan_in, width), _activation(act, beta),
                                 nn.Linear(width, FEAT), _activation(act, beta))

    def forward(self, x):
        return self.net(x)


class StructureEncoder(nn.Module):
    """21 BoneMLPs chained along the SMPL parent table (net_modules.py:114-170)."""

    def __init__(self, opt):
        super().__init__()
        self.parent_mapping = list(PARENT)
        self.num_joints = len(self.parent_mapping)
        self.out_dim = self.num_joints * FEAT
        self.net = nn.ModuleList(BoneMLP(p != -1, opt["act"], opt["beta"]) for p in self.parent_mapping)

    def get_out_dim(self):
        return self.out_dim

    def forward(self, quat):
        feats = []
        for j, (mlp, p) in enumerate(zip(self.net, self.parent_mapping)):
            own = quat[:, j, :]
            feats.append(mlp(own if p == -1 else torch.cat((own, feats[p]), dim=-1)))
        return torch.cat(feats, dim=-1)


class DFNet(nn.Module):
    """MLP in_dim -> dims... -> 1 with hidden activation and output activation (net_modules.py:9-72)."""

    def __init__(self, opt):
        super().__init__()
        widths = [opt["in_dim"], *opt["dims"], 1]
        self.num_layers = len(widths)
        for l in range(self.num_layers - 1):
            setattr(self, f"lin{l}", nn.Linear(widths[l], widths[l + 1]))
        self.actv = _activation(opt["act"], opt["beta"])
        self.out_actv = _activation(opt["act"], opt["beta"], output=True)

    def forward(self, p):
        x = p.reshape(len(p), -1)
        last = self.num_layers - 2
        for l in range(last + 1):
            x = getattr(self, f"lin{l}")(x)
            x = self.actv(x) if l < last else self.out_actv(x)
        return x
