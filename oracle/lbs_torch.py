"""ORACLE (test infrastructure, NOT product code) -- smplx's `lbs()` / `SMPL.forward` restated with torch ops, for autograd:
checks the analytic reverse pass of oracle/lbs_np.py (tests/test_lbs_oracle.py) and serves as the PyTorch-ROCm comparator
of the body-model pass in bench.py (`motion_denoise_config4.gpu_torch_baseline`).  Parity status: UNPINNED, as oracle/lbs_np.py
(smplx is third-party and absent from /root/reference; experiments/body_model.py:27-40 calls it): the structure follows the
published smplx/lbs.py -- batch_rodrigues, pose blend shapes as one matmul, batch_rigid_transform as a chain of 4x4
products, skinning as W @ A and a batched 4x4 @ 4x1 product -- i.e. the launch sequence a stock PyTorch run executes."""
import torch


def torch_lbs(theta, m, dtype=torch.float64, device=None):
    """theta [N,69] (may require grad) -> vertices [N,V,3], joints [N, 24 + n_extra, 3]"""
    dt = dtype
    device = theta.device if device is None else device
    pad = torch.nn.functional.pad
    vt, sd, b = (torch.as_tensor(m[k], dtype=dt, device=device) for k in ("v_template", "shapedirs", "betas"))
    v_shaped = vt + torch.einsum("l,mkl->mk", b, sd)
    Jr = torch.as_tensor(m["J_regressor"], dtype=dt, device=device) @ v_shaped
    N = theta.shape[0]
    full = torch.cat([torch.zeros(N, 3, dtype=dt, device=device), theta], 1).reshape(-1, 3)
    angle = torch.norm(full + 1e-8, dim=1, keepdim=True)
    rd = full / angle
    cos, sin = torch.cos(angle)[:, None], torch.sin(angle)[:, None]
    rx, ry, rz = rd[:, 0:1], rd[:, 1:2], rd[:, 2:3]
    z = torch.zeros_like(rx)
    K = torch.cat([z, -rz, ry, rz, z, -rx, -ry, rx, z], 1).view(-1, 3, 3)
    eye = torch.eye(3, dtype=dt, device=device)
    R = (eye[None] + sin * K + (1 - cos) * torch.bmm(K, K)).view(N, 24, 3, 3)
    pf = (R[:, 1:] - eye).reshape(N, -1)
    v_posed = (pf @ torch.as_tensor(m["posedirs"], dtype=dt, device=device)).view(N, -1, 3) + v_shaped[None]
    parents = [int(x) for x in m["parents"]]
    rel = Jr.clone()
    rel[1:] = rel[1:] - Jr[parents[1:]]
    M = torch.cat([pad(R.reshape(-1, 3, 3), [0, 0, 0, 1]),
                   pad(rel[None].expand(N, -1, -1).reshape(-1, 3, 1), [0, 0, 0, 1], value=1)], dim=2).reshape(N, 24, 4, 4)
    chain = [M[:, 0]]
    for i in range(1, 24):
        chain.append(chain[parents[i]] @ M[:, i])
    Gm = torch.stack(chain, 1)
    jh = pad(Jr[None, :, :, None].expand(N, -1, -1, -1), [0, 0, 0, 1])
    A = Gm - pad(Gm @ jh, [3, 0, 0, 0, 0, 0, 0, 0])
    Tm = (torch.as_tensor(m["lbs_weights"], dtype=dt, device=device)[None] @ A.view(N, 24, 16)).view(N, -1, 4, 4)
    vh = torch.cat([v_posed, torch.ones(N, v_posed.shape[1], 1, dtype=dt, device=device)], 2)
    verts = (Tm @ vh[..., None])[:, :, :3, 0]
    joints = torch.cat([Gm[:, :, :3, 3], verts[:, torch.as_tensor(m["extra_joint_vertex"], device=device).long()]], 1)
    return verts, joints


def body_terms_loss(theta, joints0, m, it, dtype=torch.float64):
    """ONE sequence theta [T,69]: 10 (1 + it) temp + [it > 0] 100 / (1 + it) data (motion_denoise.py:29-45,86-94)"""
    verts, joints = torch_lbs(theta, m, dtype)
    loss = 10.0 * (1 + it) * torch.mean(torch.sqrt(torch.sum((verts[:-1] - verts[1:]) ** 2, dim=2)))
    if it > 0:
        loss = loss + 100.0 / (1 + it) * torch.mean(torch.sqrt(torch.sum((joints - joints0) ** 2, dim=2)))
    return loss
