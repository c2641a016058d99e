"""world_size-2 `gloo` test of the multi-GPU host logic on CPU: shard bounds, the single final
all-gather and the equality with the unsharded result.  The numpy oracle stands in for the kernel -- the
sharding code is backend- and compute-agnostic, the kernel itself is covered by the -m gpu tests."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO, golden_weights


def test_shard_bounds_cover_everything():
    from posendf_amd.sharding import shard_bounds
    for total in (0, 1, 7, 64, 65, 65536, 524288):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, total, steps, outdir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    from oracle import posendf_np as onp
    from posendf_amd import synth
    from posendf_amd import sharding
    from posendf_amd.sharding import project_sharded, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = golden_weights("live")
    q_all = synth.make_poses(total, seed=77)
    lo, hi = shard_bounds(total, rank, world)

    def project_fn(q, steps):
        qo, d = onp.project(q.numpy(), sd, steps=steps)
        return torch.from_numpy(qo), torch.from_numpy(d)

    before = sharding.COLLECTIVES["all_gather_into_tensor"]
    q, d = project_sharded(project_fn, torch.from_numpy(q_all[lo:hi]), steps, total)
    # ONE collective per projection pass -- poses and distances travel together, 85 floats per pose
    assert sharding.COLLECTIVES["all_gather_into_tensor"] - before == 1
    assert q.shape == (total, 21, 4) and d.shape == (total, 1) and q.stride() == (85, 4, 1)      # views into the one receive buffer
    np.save(os.path.join(outdir, f"q_{rank}.npy"), q.numpy())
    np.save(os.path.join(outdir, f"d_{rank}.npy"), d.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("total,world", [(64, 2), (37, 2), (64, 4), (43, 4)])
def test_gloo_ranks_match_unsharded(tmp_path, total, world):
    from oracle import posendf_np as onp
    from posendf_amd import synth
    steps = 3
    port = 29500 + (os.getpid() % 2000) + total + 100 * world
    mp.spawn(_worker, args=(world, port, total, steps, str(tmp_path)), nprocs=world, join=True)
    q_ref, d_ref = onp.project(synth.make_poses(total, seed=77), golden_weights("live"), steps=steps)
    for r in range(world):
        q = np.load(tmp_path / f"q_{r}.npy")
        d = np.load(tmp_path / f"d_{r}.npy")
        assert q.shape == (total, 21, 4) and d.shape == (total, 1)
        # numpy's BLAS is not bitwise batch-size invariant (the HIP kernel is: test_full_size_properties)
        assert np.allclose(q, q_ref, rtol=1e-5, atol=1e-6) and np.allclose(d, d_ref, rtol=1e-5, atol=1e-7)


def _denoise_worker(rank, world, port, S, T, outdir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    from test_motion_denoise import _OraclePrior, _noisy_sequences
    from posendf_amd.motion_denoise import MotionDenoise
    from posendf_amd.sharding import denoise_sharded, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    md = MotionDenoise(_OraclePrior("lrelu", golden_weights("live")), device="cpu")
    theta = _noisy_sequences(S, T, seed=4)
    lo, hi = shard_bounds(S, rank, world)
    out = denoise_sharded(lambda th: md.denoise(th, iterations=2, steps_per_iter=2, record=False)[0], theta[lo:hi], S)
    np.save(os.path.join(outdir, f"th_{rank}.npy"), out.numpy())
    dist.destroy_process_group()


def test_two_rank_gloo_denoise_shards_whole_sequences(tmp_path):
    """Sequences are independent problems: 2 ranks x (2 + 1) sequences == one process with all 3."""
    from test_motion_denoise import _OraclePrior, _noisy_sequences
    from posendf_amd.motion_denoise import MotionDenoise
    S, T, world = 3, 6, 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_denoise_worker, args=(world, port, S, T, str(tmp_path)), nprocs=world, join=True)
    md = MotionDenoise(_OraclePrior("lrelu", golden_weights("live")), device="cpu")
    ref, _ = md.denoise(_noisy_sequences(S, T, seed=4), iterations=2, steps_per_iter=2, record=False)
    for r in range(world):
        got = np.load(tmp_path / f"th_{r}.npy")
        assert got.shape == (S, T, 69)
        assert np.allclose(got, ref.numpy(), atol=1e-5)
