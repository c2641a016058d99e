"""BASELINE.json configs[3] and configs[4] at FULL size on the HIP path in the only form a one-GPU box allows
(SURVEY.md section 4 "8 virtual shards", section 8e): the global batch goes through posendf_amd.sharding exactly as the
8 ranks of the real job would see it -- shard_bounds -> one project() / fused optimize() per shard -> the final
all_gather_into_tensor through an RCCL process group (one forced rank) into the preallocated receive buffer -- and must
equal the unsharded launch bit for bit (the kernels are batch-size invariant) and an oracle sample."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_weights, outlier_gate, rel_err_rows, traj_envelope

SHARDS = 8


def test_run_virtual_shards_host_logic_cpu():
    """rank-order placement, ragged blocks and tuple outputs, without a process group (CPU)."""
    from posendf_amd.sharding import run_virtual_shards, shard_bounds
    x = torch.arange(37 * 3, dtype=torch.float32).reshape(37, 3)
    seen = []

    def fn(block):
        seen.append(len(block))
        return block * 2, block.sum(dim=1, keepdim=True)

    a, b = run_virtual_shards(fn, x, SHARDS)
    assert seen == [hi - lo for lo, hi in (shard_bounds(37, r, SHARDS) for r in range(SHARDS))]
    assert torch.equal(a, x * 2) and torch.equal(b, x.sum(dim=1, keepdim=True))
    y = run_virtual_shards(lambda blk: blk + 1, x[:3], SHARDS)          # more shards than rows: empty blocks are skipped
    assert torch.equal(y, x[:3] + 1)


@pytest.fixture(scope="module")
def rccl_single_rank():
    """The process group the N > 1 job uses (backend nccl = RCCL), with the one rank this box has."""
    import torch.distributed as dist
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29600 + os.getpid() % 1000)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def _net(act, precision):
    from posendf_amd import PoseNDF, amass_config
    cfg = amass_config(act, "cuda:0")
    cfg["engine"] = {"precision": precision}
    net = PoseNDF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in golden_weights("live").items()})
    net.eval()
    return net


@pytest.mark.gpu
@pytest.mark.parametrize("act,precision", [("lrelu", "f16x3"), ("lrelu", "fp32"), ("softplus", "f16x3")])
def test_config3_batch_524288_as_eight_virtual_shards(rccl_single_rank, act, precision):
    """configs[3]: batch = 524,288 poses sharded 8 ways, 100-step project(), RCCL gather."""
    from oracle import posendf_np as onp
    from posendf_amd import synth
    from posendf_amd.sharding import run_virtual_shards
    B, steps = 524288, 100
    net = _net(act, precision)
    # the global batch as bench.py --gpus 8 builds it: rank r holds the window `offset = r` of the seeded stream
    q_all = torch.cat([torch.from_numpy(synth.make_poses(B // SHARDS, seed=1234, offset=r)) for r in range(SHARDS)]).cuda()
    recv_q = torch.empty_like(q_all)                                   # receive buffers of the final gather, allocated once
    recv_d = torch.empty(B, 1, device="cuda")
    calls = []

    def project_fn(q_shard):
        calls.append(q_shard.shape[0])
        return net.project(q_shard, steps=steps)

    q_sh, d_sh = run_virtual_shards(project_fn, q_all, SHARDS, out=(recv_q, recv_d))
    assert calls == [B // SHARDS] * SHARDS and q_sh.data_ptr() == recv_q.data_ptr()
    q_one, d_one = net.project(q_all, steps=steps)                     # the unsharded launch: 8,192 workgroups
    assert torch.equal(q_sh, q_one) and torch.equal(d_sh, d_one)
    assert torch.isfinite(q_sh).all() and torch.isfinite(d_sh).all()
    # oracle sample: 16 poses of every shard, the full 100 steps, fp32 envelope against the fp64 trajectory
    rng = np.random.default_rng(3)
    idx = np.concatenate([r * (B // SHARDS) + rng.choice(B // SHARDS, 16, replace=False) for r in range(SHARDS)])
    sd = golden_weights("live")
    q0 = q_all[idx].cpu().numpy()
    q64, _ = onp.project(q0, sd, steps=steps, act=act, dtype=np.float64)
    q32, _ = onp.project(q0, sd, steps=steps, act=act)
    outlier_gate(rel_err_rows(q_sh[idx].cpu().numpy(), q64), rel_err_rows(q32, q64), 1e-4, "config3 project100",
                 **traj_envelope(q0, sd, act, steps, q64))
    # the projection is a descent on d^2 / 2 for every shard
    d0 = net(q_all[: B // SHARDS], train=False)["dist_pred"]
    assert d_sh[: B // SHARDS].mean() < d0.mean()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_config4_512_sequences_as_eight_virtual_shards(rccl_single_rank, precision):
    """configs[4]: motion_denoise path, 512 sequences x 300 frames, whole sequences sharded 8 ways (64 per rank), the
    fused Adam step (engine launch + pndf_denoise_update) per shard, final gather of the denoised poses."""
    from oracle import denoise_np
    from posendf_amd.motion_denoise import MotionDenoise
    from posendf_amd.sharding import run_virtual_shards
    from test_motion_denoise import _noisy_sequences
    S, T, iters, per = 512, 300, 2, 3
    net = _net("lrelu", precision)
    md = MotionDenoise(net, device="cuda:0")
    theta = _noisy_sequences(S, T, seed=11).cuda()
    calls = []

    def optimize_fn(th):
        calls.append(tuple(th.shape))
        return md.denoise(th, iterations=iters, steps_per_iter=per, fused=True)[0]

    out_sh = run_virtual_shards(optimize_fn, theta, SHARDS)
    assert calls == [(S // SHARDS, T, 69)] * SHARDS
    out_one = md.denoise(theta, iterations=iters, steps_per_iter=per, fused=True)[0]     # all 153,600 frames per launch
    assert torch.equal(out_sh, out_one) and torch.isfinite(out_sh).all()
    # one sequence of the LAST shard against the numpy oracle of the loop (Adam's first steps are +-lr: compare the bulk)
    s = S - 5
    ref = denoise_np.optimize(theta[s].cpu().numpy(), golden_weights("live"), iterations=iters, steps_per_iter=per,
                              dtype=np.float64)
    diff = np.abs(out_sh[s].cpu().numpy() - ref)
    moved = np.abs(ref - theta[s].cpu().numpy()).max()
    assert np.median(diff) < 1e-5 and (diff > 1e-3).mean() < 0.01 and diff.max() < 0.5 * moved, (np.median(diff), diff.max(), moved)


@pytest.mark.gpu
def test_config4_reference_objective_as_eight_virtual_shards(rccl_single_rank):
    """configs[4] with the REFERENCE's objective (pose prior + SMPL vertex temporal term + joint data term,
    motion_denoise.py:74-99): 512 sequences x 300 frames of a synthetic SMPL-shaped body model (6,890 vertices), whole
    sequences sharded 8 ways, per shard one engine launch + fused body-model pass + Adam kernel per step.  A shard of 64
    sequences splits the vertex range over four workgroups where the full batch does not: the partial sums are added in
    another order, so sharded and unsharded runs agree to rounding, not bit for bit (Adam turns a rounding difference in a
    near-zero gradient into a step of +-lr for single entries: compare the bulk)."""
    from posendf_amd import BodyModel, synth
    from posendf_amd.motion_denoise import MotionDenoise
    from posendf_amd.sharding import run_virtual_shards
    from test_motion_denoise import _noisy_sequences
    S, T, iters, per = 512, 300, 2, 2
    net = _net("lrelu", "f16x3")
    bm = BodyModel(synth.make_body_model(seed=11), device="cuda:0")
    md = MotionDenoise(net, body_model=bm, device="cuda:0")
    theta = _noisy_sequences(S, T, seed=13).cuda()
    calls = []

    def optimize_fn(th):
        calls.append(tuple(th.shape))
        return md.denoise(th, iterations=iters, steps_per_iter=per, fused=True)[0]

    out_sh = run_virtual_shards(optimize_fn, theta, SHARDS)
    assert calls == [(S // SHARDS, T, 69)] * SHARDS and torch.isfinite(out_sh).all()
    out_one = md.denoise(theta, iterations=iters, steps_per_iter=per, fused=True)[0]
    d = (out_sh - out_one).abs().flatten()
    moved = (out_one - theta).abs().max().item()
    assert moved > 1e-3 and d.median().item() < 1e-6 and (d > 1e-3).float().mean().item() < 0.01 and d.max().item() < 0.5 * moved
    # deterministic: the same shards again give the same bits
    assert torch.equal(run_virtual_shards(optimize_fn, theta, SHARDS), out_sh)


def _gloo_single_rank(port, outdir):
    import sys
    import torch.distributed as dist
    from conftest import REPO
    sys.path.insert(0, REPO)
    from posendf_amd.sharding import run_virtual_shards
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    x = torch.arange(41 * 5, dtype=torch.float32).reshape(41, 5)
    recv = torch.empty(41, 5)
    out = run_virtual_shards(lambda blk: blk * 3, x, SHARDS, out=recv)          # every block through all_gather_into_tensor
    ok = torch.equal(out, x * 3) and out.data_ptr() == recv.data_ptr()
    dist.destroy_process_group()
    np.save(os.path.join(outdir, "ok.npy"), np.array([ok]))


def test_run_virtual_shards_through_a_process_group_cpu(tmp_path):
    """the collective branch of run_virtual_shards (one forced rank, here gloo on CPU; the -m gpu tests use RCCL)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_gloo_single_rank, args=(30500 + os.getpid() % 2000, str(tmp_path)))
    p.start()
    p.join(120)
    assert p.exitcode == 0 and bool(np.load(tmp_path / "ok.npy")[0])
