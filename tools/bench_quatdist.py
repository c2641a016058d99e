#!/usr/bin/env python3
"""HBM roofline of the quaternion distance + top-k op (reference data/prepare_traindata.py:159: k_faiss = 500 candidates
per query pose, k = 5).  Algorithmic bytes = B * (K + 1) * 336 read + B * k * 12 written."""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from posendf_amd import dist_utils  # noqa: E402


def main():
    B, K, k = 8192, 500, 5
    g = torch.Generator(device="cuda").manual_seed(0)
    noise = torch.nn.functional.normalize(torch.randn(B, 21, 4, device="cuda", generator=g), dim=2)
    valid = torch.nn.functional.normalize(torch.randn(B, K, 21, 4, device="cuda", generator=g), dim=3)
    out = {}
    for name in ("geo", "euc"):
        calc = getattr(dist_utils, name)(B, device="cuda:0")
        calc.dist_calc(noise, valid, K, k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            calc.dist_calc(noise, valid, K, k)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        nbytes = B * (K + 1) * 336 + B * k * 12
        # PyTorch-ROCm restatement of the reference's op on the same GPU
        def ref():
            n = noise.unsqueeze(1)
            d = torch.mean(1 - torch.abs(torch.sum(valid * n, dim=3)), dim=2) if name == "geo" else \
                torch.mean(torch.sqrt(torch.sum((n - valid) ** 2, dim=3)), dim=2)
            return torch.topk(d, k=k, largest=False)
        ref()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            ref()
        e1.record()
        torch.cuda.synchronize()
        out[name] = {"ms": ms, "GB/s": nbytes / ms / 1e6, "frac_of_8TBs": nbytes / ms / 1e6 / 8000,
                     "queries_per_s": B / ms * 1e3, "torch_rocm_ms": e0.elapsed_time(e1) / 5}
    print(json.dumps({"workload": f"B={B} queries x K={K} candidates, k={k}", **out}))


if __name__ == "__main__":
    main()
