"""Motion-denoise driver: CPU checks of the host logic, GPU parity of the optimisation loop
against the same loop around the PyTorch-CPU oracle network."""
import numpy as np
import pytest
import torch

from conftest import golden_weights


def test_axis_angle_to_quaternion_convention():
    from posendf_amd.motion_denoise import axis_angle_to_quaternion
    aa = torch.tensor([[0.0, 0.0, 0.0], [np.pi, 0.0, 0.0], [0.0, np.pi / 2, 0.0], [1e-8, 0.0, 0.0]], dtype=torch.float64)
    q = axis_angle_to_quaternion(aa)
    assert torch.allclose(q[0], torch.tensor([1.0, 0, 0, 0], dtype=torch.float64))
    assert torch.allclose(q[1], torch.tensor([0.0, 1, 0, 0], dtype=torch.float64), atol=1e-12)
    s = np.sin(np.pi / 4)
    assert torch.allclose(q[2], torch.tensor([s, 0, s, 0], dtype=torch.float64))
    assert torch.allclose(q[3], torch.tensor([1.0, 5e-9, 0, 0], dtype=torch.float64))
    x = torch.randn(100, 3, dtype=torch.float64)
    assert torch.allclose(axis_angle_to_quaternion(x).norm(dim=-1), torch.ones(100, dtype=torch.float64))
    # gradient through the small-angle branch is finite
    z = torch.zeros(2, 3, dtype=torch.float64, requires_grad=True)
    axis_angle_to_quaternion(z).sum().backward()
    assert torch.isfinite(z.grad).all()


def test_weight_schedule_matches_reference_formulas():
    from posendf_amd.motion_denoise import loss_weights
    w = loss_weights()
    c = torch.tensor(0.03)
    for it in (0, 1, 4, 9):
        assert torch.isclose(w["temp"](c, it), 10.0 * c * (1 + it))
        assert torch.isclose(w["data"](c, it), 100.0 * c / (1 + it))
        assert torch.isclose(w["pose_pr"](c, it), 1e7 * c * c / (1 + it))


class _OraclePrior(torch.nn.Module):
    """The oracle network behind the reference's call signature (checker only)."""

    def __init__(self, act, sd):
        super().__init__()
        from oracle.posendf_torch import RefNet
        self.net = RefNet(act)
        self.net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})

    def forward(self, pose, train=False):
        return {"dist_pred": self.net(pose)}


def _noisy_sequences(S, T, seed=0):
    g = torch.Generator().manual_seed(seed)
    walk = torch.cumsum(0.02 * torch.randn(S, T, 69, generator=g), dim=1) + 0.3 * torch.randn(S, 1, 69, generator=g)
    return walk + 0.1 * torch.randn(S, T, 69, generator=g)


def test_loop_runs_on_cpu_with_oracle_prior():
    from posendf_amd.motion_denoise import MotionDenoise
    md = MotionDenoise(_OraclePrior("lrelu", golden_weights("live")), device="cpu")
    noisy = _noisy_sequences(2, 8)
    out, hist = md.optimize(noisy, iterations=2, steps_per_iter=3)
    assert out.shape == noisy.shape and len(hist) == 6
    assert "data" not in hist[0] and "data" in hist[-1]                # data term only for it > 0 (:92)
    assert hist[2]["pose_pr"] < hist[0]["pose_pr"]                     # the (dominant) prior term is descended


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_denoise_matches_oracle_loop(precision):
    """Same optimisation loop, pose prior from the HIP engine vs from the PyTorch-CPU oracle network."""
    from posendf_amd import PoseNDF, amass_config
    from posendf_amd.motion_denoise import MotionDenoise
    sd = golden_weights("live")
    cfg = amass_config("lrelu", "cuda:0")
    cfg["engine"] = {"precision": precision}
    net = PoseNDF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    noisy = _noisy_sequences(3, 12, seed=1)
    got, h_gpu = MotionDenoise(net, device="cuda:0").optimize(noisy, iterations=2, steps_per_iter=10)
    ref, h_cpu = MotionDenoise(_OraclePrior("lrelu", sd), device="cpu").optimize(noisy, iterations=2, steps_per_iter=10)
    assert abs(h_gpu[0]["pose_pr"] - h_cpu[0]["pose_pr"]) < 1e-5 * abs(h_cpu[0]["pose_pr"]) + 1e-8
    # Adam normalises gradients, so tiny gradient differences are not amplified; 20 steps stay close
    err = (got.cpu() - ref).abs().max().item()
    assert err < 2e-3, err
    moved = (ref - noisy).abs().max().item()
    assert moved > 0.05 and err < 0.05 * moved


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_fused_step_matches_autograd_driver(precision):
    """optimize(fused=True) -- engine launch + pndf_denoise_update, no PyTorch in the loop -- against the autograd
    driver around the same engine: same terms, same Adam; differences are rounding order only."""
    from posendf_amd import PoseNDF, amass_config
    from posendf_amd.motion_denoise import MotionDenoise
    sd = golden_weights("live")
    cfg = amass_config("lrelu", "cuda:0")
    cfg["engine"] = {"precision": precision}
    net = PoseNDF(cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    noisy = _noisy_sequences(5, 31, seed=3)
    noisy[0, 3, 6:9] = 0.0                                   # a zero rotation: small-angle branch and its Jacobian
    md = MotionDenoise(net, device="cuda:0")
    # one step: the update is lr * sign-like, so compare the step itself tightly
    a1, _ = md.optimize(noisy, iterations=1, steps_per_iter=1, record=False)
    f1, _ = md.optimize(noisy, iterations=1, steps_per_iter=1, fused=True)
    assert (a1 - f1).abs().max().item() < 2e-4               # Adam's first step is lr * g / (|g| + eps): +-lr unless g ~ 0
    ref, _ = md.optimize(noisy, iterations=2, steps_per_iter=10, record=False)
    got, _ = md.optimize(noisy, iterations=2, steps_per_iter=10, fused=True)
    assert torch.isfinite(got).all()
    # 20 steps: d d / d q is discontinuous at activation kinks and Adam turns a flipped gradient component into a
    # +-lr step, so a few elements drift apart; the bulk must agree to rounding
    diff = (got - ref).abs().flatten()
    moved = (ref.cpu() - noisy).abs().max().item()
    print(f"fused vs autograd driver after 20 steps: median {diff.median().item():.2e} p99 "
          f"{diff.kthvalue(int(0.99 * diff.numel())).values.item():.2e} max {diff.max().item():.2e} (moved {moved:.2f})")
    assert diff.median().item() < 1e-5
    assert (diff > 1e-3).float().mean().item() < 0.01
    assert diff.max().item() < 0.1 * moved
    assert torch.equal(got[..., 63:].cpu(), noisy[..., 63:])  # hand joints: no term, Adam leaves them in place


@pytest.mark.gpu
def test_aa2quat_kernel_matches_restatement():
    import ctypes
    from posendf_amd.engine import load_library
    from posendf_amd.motion_denoise import axis_angle_to_quaternion
    lib = load_library()
    theta = (0.8 * torch.randn(257, 69)).cuda()
    theta[5, 0:3] = 0.0
    theta[6, 3:6] = 1e-8
    q = torch.empty(257, 21, 4, device="cuda")
    assert lib.pndf_aa2quat(theta.data_ptr(), q.data_ptr(), 257, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    want = axis_angle_to_quaternion(theta.reshape(257, 23, 3)[:, :21])
    assert torch.allclose(q, want, atol=2e-7, rtol=1e-6)
