"""Multi-GPU host logic for the projection path: the batch of poses shards by contiguous blocks with no
collective during the projection steps (every op of the path is per pose); the only
collective is one all-gather of the projected poses (+ distances) at the end (RCCL over xGMI when the backend
is `nccl`).  Backend-agnostic so that the same code is exercised with `gloo` on CPU in the tests."""
from __future__ import annotations

import torch


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank `rank`; the first total % world ranks get one extra pose."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_blocks(x: torch.Tensor, total: int, group=None):
    """Gather per-rank blocks (possibly ragged by one row) into the [total, ...] tensor, in rank order."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return x
    world = dist.get_world_size(group)
    rows = max(shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world))
    pad = x
    if x.shape[0] < rows:   # ragged tail: pad to the common block size, trimmed below
        pad = torch.cat([x, x.new_zeros((rows - x.shape[0],) + tuple(x.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    parts = []
    for r, b in enumerate(bufs):
        lo, hi = shard_bounds(total, r, world)
        parts.append(b[: hi - lo])
    return torch.cat(parts)


def project_sharded(project_fn, q_shard: torch.Tensor, steps: int, total: int, group=None):
    """project_fn(q_shard, steps) -> (q_out, d_last) on this rank's block, then the single final gather.
    Returns (q_all [total,21,4], d_all [total,1]) on every rank."""
    q_out, d_last = project_fn(q_shard, steps)
    return all_gather_blocks(q_out, total, group), all_gather_blocks(d_last, total, group)


def denoise_sharded(optimize_fn, theta_shard: torch.Tensor, total_sequences: int, group=None):
    """Motion denoising shards by WHOLE sequences (BASELINE.json configs[4]: 512 sequences over 8 GPUs): the
    per-sequence mean of the pose prior and the temporal coupling never cross a sequence (reference
    experiments/motion_denoise.py:83,88-89 -- one main() per sequence :171-188), so there is no collective until
    the final gather of the denoised poses.  optimize_fn(theta_shard [S_r,T,69]) -> denoised [S_r,T,69]."""
    out = optimize_fn(theta_shard)
    return all_gather_blocks(out, total_sequences, group)
