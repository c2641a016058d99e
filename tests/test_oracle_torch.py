"""Pin the PyTorch-CPU restatement used as bench.py's cpu_baseline against the reference-generated vectors."""
import numpy as np
import torch

from conftest import d_err, rel_err


def test_torch_oracle_matches_golden(golden_case):
    act, regime, g, sd = golden_case
    from oracle.posendf_torch import RefNet, project
    net = RefNet(act)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    q = torch.from_numpy(g["q"]).requires_grad_(True)
    d = net(q)
    (dq,) = torch.autograd.grad(d.sum(), q)
    assert d_err(d.detach().numpy(), g["d_f32"]) < 1e-6
    assert rel_err(dq.numpy(), g["dq_f32"]) < 1e-6
    q1, d1 = project(net, torch.from_numpy(g["q"]), 1)
    assert rel_err(q1.numpy(), g["q1_f32"]) < 1e-6
