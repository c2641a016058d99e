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


# ---- synthetic SMPL-shaped body model (posendf_amd.BodyModel, csrc/pndf_lbs.hip): the licensed SMPL file is not reachable,
# so tests and benchmarks draw random parameters with its shapes and structure.
# SMPL kinematic tree (smplx: model.parents = kintree_table[0], root = -1) -- data of the model file, reproduced here
# only as the default of the synthetic model generator
SMPL_PARENTS = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21)
SMPL_V = 6890
# smplx VertexJointSelector for SMPL (vertex_ids['smplh']): nose, eyes, ears, feet (big toe, small toe, heel; L then R),
# finger tips (thumb .. pinky; L then R) -- 21 extra joints, 45 in all.  Restated from memory of the published table:
# part of what "parity unpinned" covers; the product takes the list as a parameter.
SMPL_EXTRA_JOINT_VERTICES = (332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
                             2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133)


def make_body_model(V=SMPL_V, n_betas=10, seed=0, parents=SMPL_PARENTS, extra=SMPL_EXTRA_JOINT_VERTICES, max_influences=4):
    """A random body model with SMPL's shapes and structure: a template cloud of ~1.7 m, a joint regressor with convex
    rows, skinning weights with <= 4 non-zeros per vertex that sum to one, shape dirs of ~1 cm and pose dirs of ~1 mm per
    unit of the pose feature (SMPL's pose-corrective magnitudes)."""
    rng = np.random.default_rng(seed)
    J = len(parents)
    v_template = (rng.normal(size=(V, 3)) * np.array([0.25, 0.55, 0.12])).astype(np.float32)
    shapedirs = (rng.normal(size=(V, 3, n_betas)) * 0.01).astype(np.float32)
    posedirs = (rng.normal(size=((J - 1) * 9, V * 3)) * 0.002).astype(np.float32)
    jr = rng.random((J, V)) ** 8
    J_regressor = (jr / jr.sum(1, keepdims=True)).astype(np.float32)
    w = np.zeros((V, J), np.float64)
    for v in range(V):
        idx = rng.choice(J, max_influences, replace=False)
        w[v, idx] = rng.random(max_influences) + 0.05
    lbs_weights = (w / w.sum(1, keepdims=True)).astype(np.float32)
    # a smaller cloud than SMPL's folds the table's vertex ids into range; folded ids that collide move on to the next free
    # vertex (the kernels serve one picked joint per vertex and pndf_lbs_pack_host refuses duplicates)
    if len(extra) > V:
        raise ValueError(f"{len(extra)} vertex-picked joints need at least as many vertices, V = {V}")
    taken, uniq = set(), []
    for e in extra:
        e = int(e) % V
        while e in taken:
            e = (e + 1) % V
        taken.add(e)
        uniq.append(e)
    extra = tuple(uniq)
    return dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, J_regressor=J_regressor,
                parents=np.asarray(parents, np.int32), lbs_weights=lbs_weights,
                extra_joint_vertex=np.asarray(extra, np.int32), betas=np.zeros(n_betas, np.float32))
