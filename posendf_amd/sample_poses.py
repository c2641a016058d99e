"""Pose generation by projection -- mirror of the reference's `SamplePose` (experiments/sample_poses.py:37-105): random unit
quaternion poses are projected onto the manifold by the loop of :67-74 (here: ONE persistent kernel launch,
`PoseNDF.project`), and the poses before / after are turned into SMPL meshes for inspection (:59-62, :78-83:
`quaternion_to_axis_angle` + the body model) -- on the HIP body model of posendf_amd.BodyModel when one is given.
Rendering / mesh files (pytorch3d, :47-55) are out of scope; the vertices are returned instead.

`quaternion_to_axis_angle` is pytorch3d's (third-party, absent: parity unpinned), restated from its documented
convention (real part first; angle = 2 atan2(|v|, w); the small-angle series of sin(x/2)/x below 1e-6) and self-tested as
the inverse of motion_denoise.axis_angle_to_quaternion.
"""
from __future__ import annotations

import torch


def quaternion_to_axis_angle(quaternions: torch.Tensor) -> torch.Tensor:
    """[..., 4] (real part first, need not be normalised for the angle's sign convention) -> [..., 3] axis * angle."""
    norms = torch.norm(quaternions[..., 1:], p=2, dim=-1, keepdim=True)
    half_angles = torch.atan2(norms, quaternions[..., :1])
    angles = 2 * half_angles
    small = angles.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    sin_half_over_angle = torch.where(small, 0.5 - angles * angles / 48, torch.sin(half_angles) / safe)
    return quaternions[..., 1:] / sin_half_over_angle


def random_poses(batch_size: int, device="cuda:0", generator=None) -> torch.Tensor:
    """experiments/sample_poses.py:96-97: normalize(torch.rand(B, 21, 4), dim=2)."""
    q = torch.rand((batch_size, 21, 4), generator=generator)
    return torch.nn.functional.normalize(q, dim=2).to(device=device)


class SamplePose:
    def __init__(self, posendf, body_model=None, device="cuda:0"):
        self.pose_prior = posendf
        self.body_model = body_model
        self.device = device

    def _mesh(self, poses):
        """:59-61 / :78-80: quaternions -> axis-angle body pose [B,69] (hands zero) -> SMPL vertices, joints"""
        aa = torch.zeros((len(poses), 23, 3), device=poses.device, dtype=torch.float32)
        aa[:, :21] = quaternion_to_axis_angle(poses.detach().float())
        out = self.body_model(pose_body=aa.view(-1, 69))
        return aa.view(-1, 69), out.vertices.detach(), out.Jtr.detach()

    @torch.no_grad()
    def project(self, noisy_poses, steps=10):
        """:57-83 with the ten iterations of :70 as `steps`.  Returns the projected poses [B,21,4], dist_pred of the last
        iteration [B,1] and, with a body model, the meshes before / after ({'pose_init', 'vertices_init', 'pose', 'vertices',
        'joints'})."""
        noisy_poses = noisy_poses.to(self.device)
        meshes = {}
        if self.body_model is not None:
            meshes["pose_init"], meshes["vertices_init"], _ = self._mesh(noisy_poses)
        poses, dist = self.pose_prior.project(noisy_poses, steps=steps)
        if self.body_model is not None:
            meshes["pose"], meshes["vertices"], meshes["joints"] = self._mesh(poses)
        return poses, dist, meshes


def sample_pose(net, batch_size=10, steps=10, body_model=None, device="cuda:0", generator=None):
    """experiments/sample_poses.py:86-105 after the checkpoint is loaded: draw random poses, project them."""
    sampler = SamplePose(net, body_model=body_model, device=device)
    return sampler.project(random_poses(batch_size, device, generator), steps=steps)
