"""Quaternion pose distances + k nearest candidates on the HIP engine -- the interface of the reference's
`data/dist_utils.py` (classes `geo` :32-50 and `euc` :9-30, called from `data/prepare_traindata.py:159`):

    calc = geo(batch_size, device)                      # or euc(...); weighted=True uses the joint-rank weights
    val, idx = calc.dist_calc(noise_quats[B,21,4], valid_quat[B,K,21,4], k_faiss, k_dist)

Differences from the reference: `k_dist` is honoured (the reference hard-codes k=5 and ignores it); ties are broken
towards the lower index (torch.topk leaves them unspecified).  One HBM-bound kernel (csrc/pndf_quatdist.hip); no
CPU fallback.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from .engine import PndfError, load_library

JOINT_RANK = (7, 7, 7, 6, 6, 6, 5, 5, 5, 4, 4, 4, 4, 4, 3, 3, 3, 2, 2, 1, 1)      # dist_utils.py:17,40


class _QuatDist:
    metric = -1

    def __init__(self, batch_size, device="cuda", weighted=False):
        self.device = torch.device(device)
        self.batch_size = batch_size
        self.weighted = weighted
        rank = torch.tensor(JOINT_RANK, dtype=torch.float32)
        self.joint_weights = torch.nn.functional.normalize(rank, dim=0)             # dist_utils.py:18,41
        self._w = (ctypes.c_float * 21)(*self.joint_weights.tolist()) if weighted else None
        self._lib = load_library()

    def dist_calc(self, noise_quats, valid_quat, k_faiss, k_dist=5):
        noise = noise_quats.to(self.device, torch.float32).reshape(-1, 21, 4).contiguous()
        B = noise.shape[0]
        valid = valid_quat.to(self.device, torch.float32).reshape(B, -1, 21, 4).contiguous()
        K = valid.shape[1]
        if K != int(k_faiss):
            raise PndfError(f"valid_quat holds {K} candidates per pose, k_faiss says {k_faiss}")
        k = int(k_dist)
        vals = torch.empty(B, k, device=self.device, dtype=torch.float32)
        idx = torch.empty(B, k, device=self.device, dtype=torch.int64)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = self._lib.pndf_quat_topk(noise.data_ptr(), valid.data_ptr(), B, K, self.metric, self._w, k,
                                      vals.data_ptr(), idx.data_ptr(), stream)
        if rc != 0:
            raise PndfError(f"pndf_quat_topk failed ({rc}): B={B} K={K} k={k} (k <= min(K, 16), K <= ~1850)")
        return vals, idx


class geo(_QuatDist):
    """mean_j (1 - |<q_valid_j, q_noise_j>|)   (dist_utils.py:43-47)"""
    metric = 0


class euc(_QuatDist):
    """mean_j ||q_noise_j - q_valid_j||_2      (dist_utils.py:21-27)"""
    metric = 1
