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


def all_gather_blocks(x: torch.Tensor, total: int, group=None, out: torch.Tensor | None = None):
    """Gather per-rank blocks (possibly ragged by one row) into the [total, ...] tensor, in rank order.
    Equal blocks (total % world == 0, the benchmark's case) are gathered with ONE all_gather_into_tensor straight into
    `out` (allocated by the caller once, or here): no per-rank receive buffers, no concatenation.  Ragged blocks are
    padded to the common size, gathered the same way and trimmed."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return x
    world = dist.get_world_size(group)
    x = x.contiguous()
    if x.is_cuda and dist.get_backend(group) == "gloo":
        # gloo is the CPU / test backend (RCCL refuses two ranks on one device, gloo does not: bench.py
        # PNDF_BENCH_BACKEND=gloo): device blocks are staged through the host explicitly instead of relying on gloo's
        # own device-tensor support
        host = all_gather_blocks(x.cpu(), total, group)
        if out is None:
            return host.to(x.device)
        out.copy_(host)
        return out
    if total % world == 0:
        assert x.shape[0] == total // world, (x.shape, total, world)
        if out is None:
            out = x.new_empty((total,) + tuple(x.shape[1:]))
        COLLECTIVES["all_gather_into_tensor"] += 1
        dist.all_gather_into_tensor(out, x, group=group)
        return out
    rows = -(-total // world)                  # the first total % world ranks hold `rows`, the others rows - 1
    pad = x
    if x.shape[0] < rows:
        pad = torch.cat([x, x.new_zeros((rows - x.shape[0],) + tuple(x.shape[1:]))])
    buf = x.new_empty((world * rows,) + tuple(x.shape[1:]))
    COLLECTIVES["all_gather_into_tensor"] += 1
    dist.all_gather_into_tensor(buf, pad, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        parts.append(buf[r * rows: r * rows + (hi - lo)])
    res = torch.cat(parts)
    if out is not None:
        out.copy_(res)
        return out
    return res


ROW_FLOATS = 85      # one projected pose on the wire: 84 quaternion components + its last distance


def pack_rows(q_out: torch.Tensor, d_last: torch.Tensor) -> torch.Tensor:
    """[rows,21,4] poses + [rows,1] distances -> ONE [rows,85] send buffer, so that the final gather is one collective
    with one message per rank (22.3 MB for 65,536 poses) instead of two (a 22 MB and a 0.26 MB one)."""
    rows = q_out.shape[0]
    return torch.cat([q_out.reshape(rows, ROW_FLOATS - 1), d_last.reshape(rows, 1).to(q_out.dtype)], dim=1)


def unpack_rows(rows85: torch.Tensor):
    """Views (no copy) into a gathered [total,85] buffer: poses [total,21,4] (row stride 85) and distances [total,1]."""
    return rows85[:, :ROW_FLOATS - 1].unflatten(1, (21, 4)), rows85[:, ROW_FLOATS - 1:]


COLLECTIVES = {"all_gather_into_tensor": 0}      # calls issued by this module (the tests assert ONE per projection pass)


def gather_projected(q_out: torch.Tensor, d_last: torch.Tensor, total: int, group=None, out: torch.Tensor | None = None):
    """The single final collective of the projection path: every rank contributes its block of (pose, distance) rows,
    every rank receives all of them in rank order.  `out`: a preallocated [total,85] receive buffer (bench.py allocates
    it once).  Returns (q_all [total,21,4], d_all [total,1]) as views into the receive buffer."""
    return unpack_rows(all_gather_blocks(pack_rows(q_out, d_last), total, group, out=out))


def project_sharded(project_fn, q_shard: torch.Tensor, steps: int, total: int, group=None, out: torch.Tensor | None = None):
    """project_fn(q_shard, steps) -> (q_out, d_last) on this rank's block, then the single final gather (ONE
    all_gather_into_tensor of 85 floats per pose).  Returns (q_all [total,21,4], d_all [total,1]) on every rank."""
    q_out, d_last = project_fn(q_shard, steps)
    return gather_projected(q_out, d_last, total, group, out=out)


def denoise_sharded(optimize_fn, theta_shard: torch.Tensor, total_sequences: int, group=None):
    """Motion denoising shards by WHOLE sequences (BASELINE.json configs[4]: 512 sequences over 8 GPUs): the
    per-sequence mean of the pose prior and the temporal coupling never cross a sequence (reference
    experiments/motion_denoise.py:83,88-89 -- one main() per sequence :171-188), so there is no collective until
    the final gather of the denoised poses.  optimize_fn(theta_shard [S_r,T,69]) -> denoised [S_r,T,69]."""
    out = optimize_fn(theta_shard)
    return all_gather_blocks(out, total_sequences, group)


def run_virtual_shards(fn, x_all: torch.Tensor, shards: int, group=None, out=None):
    """One-device rehearsal of an N-rank job (SURVEY.md section 4 "8 virtual shards"): `fn` runs on every contiguous block
    of `x_all` in rank order -- exactly what rank r of a `shards`-rank job would run on its shard -- and each block's
    outputs are placed where the final all-gather puts that rank's block.  When a process group exists (a single forced
    rank on a one-GPU box) every block goes through the SAME collective the N-rank job uses, `all_gather_into_tensor`
    straight into its window of the preallocated receive buffer, so the collective sees the message sizes of the real
    job.  fn(block) -> tensor or tuple of tensors whose leading dimension is the block's.  Returns the gathered tensor(s)
    with leading dimension len(x_all)."""
    import torch.distributed as dist
    total = x_all.shape[0]
    use_dist = dist.is_available() and dist.is_initialized()
    if use_dist and dist.get_world_size(group) != 1:
        raise ValueError("run_virtual_shards emulates the ranks of a job on ONE process; use project_sharded / "
                         "denoise_sharded on a real multi-rank group")
    outs = None
    for r in range(shards):
        lo, hi = shard_bounds(total, r, shards)
        res = fn(x_all[lo:hi])
        single = not isinstance(res, (tuple, list))
        res = (res,) if single else tuple(res)
        if outs is None:
            outs = out if out is not None else tuple(y.new_empty((total,) + tuple(y.shape[1:])) for y in res)
            outs = (outs,) if isinstance(outs, torch.Tensor) else tuple(outs)
        for dst, y in zip(outs, res):
            if hi == lo:
                continue
            if use_dist:
                dist.all_gather_into_tensor(dst[lo:hi], y.contiguous(), group=group)
            else:
                dst[lo:hi].copy_(y)
    return outs[0] if len(outs) == 1 else outs
