"""ORACLE (test infrastructure only -- tests/ and smoke() may import it; the product never does).

CPU restatement of the reference's training-data distance op: quaternion "geodesic" and euclidean
pose distances between one query pose and K candidate poses, followed by the k smallest.
Follows /root/reference/data/dist_utils.py:
    :9-30   class euc  -- mean (or joint-weighted sum) over 21 joints of ||q_noise - q_valid||_2, topk(k=5, largest=False)
    :32-50  class geo  -- mean (or joint-weighted sum) over 21 joints of 1 - |<q_valid, q_noise>|, same top-k
    :17-18 / :40-41    -- joint weights = L2-normalised [7,7,7,6,6,6,5,5,5,4,4,4,4,4,3,3,3,2,2,1,1]
(k is hard-coded to 5 there, the k_dist argument is ignored; here it is a parameter.)
PINNED: tests/golden/quat_dist.npz was produced by importing that module (tests/golden/make_golden_dist.py).
"""
import numpy as np

JOINT_RANK = np.array([7, 7, 7, 6, 6, 6, 5, 5, 5, 4, 4, 4, 4, 4, 3, 3, 3, 2, 2, 1, 1], dtype=np.float32)


def joint_weights():
    w = JOINT_RANK.astype(np.float32)
    return w / max(float(np.sqrt((w.astype(np.float64) ** 2).sum())), 1e-12)     # F.normalize(joint_rank, dim=0)


def pose_distances(noise, valid, metric="geo", weighted=False, dtype=np.float32):
    """noise [B,21,4], valid [B,K,21,4] -> [B,K]."""
    n = np.asarray(noise, dtype=dtype)[:, None]
    v = np.asarray(valid, dtype=dtype)
    if metric == "geo":
        per_joint = 1.0 - np.abs((v * n).sum(axis=3))
    elif metric == "euc":
        d = n - v
        per_joint = np.sqrt((d * d).sum(axis=3))
    else:
        raise ValueError(metric)
    if weighted:
        return (joint_weights().astype(dtype) * per_joint).sum(axis=2)
    return per_joint.mean(axis=2)


def dist_calc(noise, valid, k=5, metric="geo", weighted=False, dtype=np.float32):
    """-> (values [B,k] ascending, indices [B,k]); ties broken towards the lower index."""
    d = pose_distances(noise, valid, metric, weighted, dtype)
    idx = np.argsort(d, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(d, idx, axis=1), idx
