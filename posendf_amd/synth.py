"""Deterministic synthetic weights and poses for the Pose-NDF hot path.

No pretrained checkpoint is reachable offline (reference README.md:61 is a network link), so the
benchmarks, fixtures and tests all use this generator.  It reproduces the *shapes and key names*
of the reference parameter tree (reference model/network/net_modules.py:14-28,78-107,116-128; 98
tensors, 1,365,565 parameters for configs/amass.yaml) and draws values in the "live regime"
SURVEY.md section 7 describes: uniform(-1, 1)/sqrt(fan_in) * gain with dfnet.lin6.bias pinned, so
that the output ReLU does not clip every pose to d == 0.

Pure numpy: usable by the product (bench.py), by the oracle and by the fixture generator.
"""
from __future__ import annotations

import numpy as np

# reference model/network/net_utils.py:46 -- reproduced as data (21-entry SMPL parent table)
PARENT = (-1, -1, -1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19)
NUM_JOINTS = 21
FEAT = 6            # local_feature_size, reference net_modules.py:116
BONE_DIM = 4
HID = BONE_DIM + FEAT   # 10, reference net_modules.py:84
DFNET_DIMS = (126, 256, 512, 1024, 512, 256, 64, 1)   # configs/amass.yaml:26,30
DFNET_DIMS_NOENC = (84,) + DFNET_DIMS[1:]              # model.StrEnc.use = False: DFNet on the 21 x 4 quaternions


def state_dict_shapes(dims=DFNET_DIMS):
    """Ordered {key: shape} of the reference state dict.  dims[0] == 84 is the
    encoder-less variant (reference model/posendf.py:40-42,73-74): 14 DFNet tensors only."""
    shapes = {}
    for i, p in (enumerate(PARENT) if dims[0] != NUM_JOINTS * BONE_DIM else ()):
        fin = BONE_DIM if p == -1 else BONE_DIM + FEAT
        shapes[f"enc.net.{i}.net.0.weight"] = (HID, fin)
        shapes[f"enc.net.{i}.net.0.bias"] = (HID,)
        shapes[f"enc.net.{i}.net.2.weight"] = (FEAT, HID)
        shapes[f"enc.net.{i}.net.2.bias"] = (FEAT,)
    for l in range(len(dims) - 1):
        shapes[f"dfnet.lin{l}.weight"] = (dims[l + 1], dims[l])
        shapes[f"dfnet.lin{l}.bias"] = (dims[l + 1],)
    return shapes


def make_weights(seed: int = 0, gain: float = 2.0, out_bias: float = 0.1, dims=DFNET_DIMS):
    """Deterministic fp32 weights keyed like the reference state dict."""
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape in state_dict_shapes(dims).items():
        layer = key.rsplit(".", 1)[0]
        fan_in = state_dict_shapes(dims)[layer + ".weight"][1]
        bound = gain / np.sqrt(fan_in)
        sd[key] = rng.uniform(-bound, bound, size=shape).astype(np.float32)
    last = f"dfnet.lin{len(dims) - 2}.bias"
    sd[last] = np.full(sd[last].shape, out_bias, dtype=np.float32)
    return sd


def make_poses(batch: int, seed: int = 1234, signed: bool = False, offset: int = 0):
    """Synthetic input poses, reference experiments/sample_poses.py:96-97:
    normalize(torch.rand(B,21,4), dim=2) -- components U[0,1) then per-quaternion unit norm.
    signed=True draws U(-1,1) instead (full-sphere quaternions).  `offset` selects a window of a
    conceptually infinite stream so that shards of one global batch are reproducible per rank."""
    rng = np.random.default_rng([seed, offset])
    q = rng.random((batch, NUM_JOINTS, 4), dtype=np.float32)
    if signed:
        q = q * 2.0 - 1.0
    n = np.sqrt((q.astype(np.float64) ** 2).sum(-1, keepdims=True))
    return (q / np.maximum(n, 1e-12)).astype(np.float32)


def make_candidates(B: int, K: int, seed: int):
    """Inputs of the quaternion distance + top-k op (reference data/dist_utils.py): B signed-unit-quaternion query
    poses and K candidate poses each, with an exact match + a tie (query 0), and an antipodal copy (query 1:
    geodesic distance 0, euclidean 2).  Used by tests/golden/make_golden_dist.py and the tests, so only outputs are
    stored as fixtures."""
    noise = make_poses(B, seed=seed, signed=True)
    valid = make_poses(B * K, seed=seed + 100, signed=True).reshape(B, K, 21, 4)
    if K > 7:
        valid[0, 3] = noise[0]
        valid[0, 7] = noise[0]
        if B > 1:
            valid[1, 5] = -noise[1]
    return noise, valid
