"""Motion denoising around the engine -- the second caller of the hot path.

Mirrors the optimiser structure of the reference's `MotionDenoise.optimize`
(experiments/motion_denoise.py:58-121): Adam(lr=0.02, betas=(0.9, 0.999)) over SMPL axis-angle body poses,
10 x 50 steps, loss = sum_k w_k(loss_k, it) with the reference's iteration-dependent weights (:29-35):
    temp   : 10    * c * (1 + it)
    data   : 100   * c / (1 + it)          (only for it > 0, :92)
    pose_pr: 1e7   * c * c / (1 + it)      with c = mean_t PoseNDF(q_t)  (:81-83)
The pose-prior term -- the part on the hot path -- runs on the HIP engine through `PoseNDF.forward(train=False)`
and its first-order autograd contract.  The reference's temporal / data terms need the SMPL body model
(vertices, joints: third-party code + licensed model files, parity unpinned); here they are
pluggable: pass `body_model(pose_body[T,69]) -> (vertices[T,V,3], joints[T,J,3])`, or leave it None to use
pose-space surrogates (per-joint axis-angle differences), which keep the objective's structure.

`denoise(fused=True)` runs the same loop without PyTorch in it: per Adam step one engine launch (distances and
d d / d q for all S x T frames) and one HIP kernel (`pndf_denoise_update`, posendf_amd/csrc/pndf_denoise.hip) that
does the per-sequence mean, the weights, the axis-angle Jacobian, the pose-space terms, Adam and the next step's
quaternions -- 3 launches per step instead of ~60.  With `body_model is None` it optimises the surrogate terms; with a
`posendf_amd.BodyModel` (the HIP linear-blend-skinning kernels, csrc/pndf_lbs.hip) it optimises the REFERENCE's objective:
per Adam step the engine launch, the fused body-model pass (`pndf_lbs_terms_grad`: vertices, the vertex temporal term and the
joint data term of motion_denoise.py:86-94 and their gradient, nothing per-vertex in HBM) and `pndf_denoise_update_body`.

Sequences are independent problems (one `main()` per sequence in the reference, :171-188): a batch [S, T, 69] is
optimised with per-sequence means, one engine launch per Adam step for all S x T frames.
"""
from __future__ import annotations

import torch


def axis_angle_to_quaternion(axis_angle: torch.Tensor) -> torch.Tensor:
    """pytorch3d.transforms.axis_angle_to_quaternion restated from its documented convention (real part first;
    q = [cos(|a|/2), a * sin(|a|/2) / |a|], Taylor series 1/2 - |a|^2/48 below 1e-6).  Third-party arithmetic that
    is not under /root/reference: parity unpinned; self-tested in tests/test_motion_denoise.py."""
    angles = torch.norm(axis_angle, p=2, dim=-1, keepdim=True)
    half = 0.5 * angles
    small = angles.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    k = torch.where(small, 0.5 - angles * angles / 48.0, torch.sin(half) / safe)
    return torch.cat([torch.cos(half), axis_angle * k], dim=-1)


# The reference ships the loop twice with different weights: experiments/motion_denoise.py:29-35 and its copy
# experiments/partial_observation.py:29-35 (whose pose prior is LINEAR in the mean distance).  (coefficient, power of the
# loss, exponent of (1 + it)) per term:
SCHEDULES = {
    "motion_denoise": {"temp": (10.0 ** 1, 1, +1), "data": (10.0 ** 2, 1, -1), "pose_pr": (10.0 ** 7, 2, -1)},
    "partial_observation": {"temp": (10.0 ** 2, 1, +1), "data": (10.0 ** 1, 1, -1), "pose_pr": (10.0 ** 2, 1, -1)},
}


def loss_weights(schedule="motion_denoise"):
    """experiments/motion_denoise.py:29-35 (default) or experiments/partial_observation.py:29-35: {term: f(loss, it)}."""
    def make(coef, power, e):
        return lambda cst, it: coef * cst ** power * (1 + it) ** e
    return {k: make(*v) for k, v in SCHEDULES[schedule].items()}


def iteration_coefs(schedule, it):
    """(prior_coef, prior_power, temp_coef, data_coef) of outer iteration `it` (the data term only for it > 0, :92): what the
    C ABI takes (pndf_denoise_weights, pndf_lbs_terms_grad_w)."""
    s = SCHEDULES[schedule]
    ev = lambda k: s[k][0] * (1 + it) ** s[k][2]
    return ev("pose_pr"), s["pose_pr"][1], ev("temp"), (ev("data") if it > 0 else 0.0)


class MotionDenoise:
    """Positional arguments are the reference's (experiments/motion_denoise.py:21: `MotionDenoise(posendf, body_model,
    out_path, debug, device, batch_size, gender)`), so `MotionDenoise(net, body_model=bm, batch_size=len(poses),
    out_path=path)` (:151) constructs this class unchanged.  `out_path`, `debug` and `gender` only feed the reference's mesh
    export / renderer (:47-56, SURVEY.md section 2: out of scope) and are kept as attributes; `batch_size` sized the
    reference's zero betas (:27) -- the betas are fixed at the body model's construction here.  `schedule` (keyword only)
    selects the weights of experiments/partial_observation.py:29-35 instead.

    Two entry points: `optimize(noisy_poses, gt_poses=None, iterations=10, steps_per_iter=50)` is the reference's (:58):
    one sequence [T,69] (or a batch [S,T,69]) in, the vertex-to-vertex error in cm out; `denoise(...)` is the batched
    form the rest of this package uses (poses and loss history out, autograd or fused driver)."""

    def __init__(self, posendf, body_model=None, out_path="./experiment_results/motion_denoise", debug=False, device="cuda:0",
                 batch_size=1, gender="male", *, schedule="motion_denoise"):
        if schedule not in SCHEDULES:
            raise ValueError(f"unknown weight schedule {schedule!r} ({', '.join(SCHEDULES)})")
        self.pose_prior = posendf
        self.body_model = body_model
        self.out_path, self.debug, self.batch_size, self.gender = out_path, debug, batch_size, gender
        self.device = device
        self.schedule = schedule          # "partial_observation": the weights of experiments/partial_observation.py:29-35
        self.last_poses = None            # denoised poses of the last optimize() call

    # ---- loss terms -----------------------------------------------------------------------------
    def pose_prior_term(self, body_pose):
        """[S,T,69] -> per-sequence mean distance [S]  (motion_denoise.py:81-83)."""
        S, T = body_pose.shape[:2]
        quat = axis_angle_to_quaternion(body_pose.reshape(S * T, 23, 3)[:, :21])
        dist = self.pose_prior(quat, train=False)["dist_pred"]
        return dist.reshape(S, T).mean(dim=1)

    def _geometry(self, body_pose):
        S, T = body_pose.shape[:2]
        if self.body_model is None:
            pts = body_pose.reshape(S, T, 23, 3)[:, :, :21]          # pose-space surrogate "vertices" = "joints"
            return pts, pts
        from .body_model import BodyModel
        if isinstance(self.body_model, BodyModel):          # the reference's call (motion_denoise.py:86): vertices and Jtr
            out = self.body_model(pose_body=body_pose.reshape(S * T, 69))
            v, j = out.vertices, out.Jtr
        else:
            v, j = self.body_model(body_pose.reshape(S * T, 69))
        return v.reshape(S, T, *v.shape[1:]), j.reshape(S, T, *j.shape[1:])

    def _mean_norm(self, x):
        """mean over (t, point) of the Euclidean norm, per sequence (motion_denoise.py:89,94).  With a body model this
        is the reference's formula as written; the pose-space surrogate adds 1e-20 under the root, so that two frames
        with an identical joint rotation have a zero instead of a NaN gradient (the reference skips its data term at
        it = 0 for the same reason, :92 "for nans")."""
        eps = 1e-20 if self.body_model is None else 0.0
        return torch.sqrt((x * x).sum(dim=-1) + eps).mean(dim=(1, 2))

    def losses(self, body_pose, init_joints, it):
        loss = {"pose_pr": self.pose_prior_term(body_pose)}
        verts, joints = self._geometry(body_pose)
        if verts.shape[1] > 1:       # a one-frame "sequence" has no temporal term (the reference's mean over nothing is NaN)
            loss["temp"] = self._mean_norm(verts[:, :-1] - verts[:, 1:])             # :88-89
        if it > 0:                                                                    # :92 ("for nans")
            loss["data"] = self._mean_norm(joints - init_joints)                      # :93-94
        return loss

    def total(self, loss, it):
        w = loss_weights(self.schedule)
        return torch.stack([w[k](v, it) for k, v in loss.items()]).sum(dim=0)         # backward_step, :37-45

    # ---- optimiser ------------------------------------------------------------------------------
    def _optimize_fused(self, pose, iterations, steps_per_iter, lr):
        """The loop of `optimize` on the engine + pndf_denoise_update, no autograd (module docstring)."""
        import ctypes
        from .body_model import BodyModel
        bm = self.body_model
        if bm is not None and not isinstance(bm, BodyModel):
            raise ValueError("fused=True needs body_model None (pose-space surrogates) or a posendf_amd.BodyModel (HIP LBS); "
                             "an arbitrary callable runs through the autograd driver (fused=False)")
        S, T = pose.shape[:2]
        N = S * T
        dev = pose.device
        if dev.type != "cuda":
            raise ValueError("fused=True is the HIP driver (engine launch + pndf_denoise_update per Adam step): poses on "
                             f"{dev}; the autograd driver (fused=False) runs anywhere the PoseNDF model does")
        if bm is not None and torch.device(dev.type, dev.index or 0) != bm.device:
            raise ValueError(f"poses on {dev} but the body model lives on {bm.device}: the fused step hands raw pointers to both")
        eng = self.pose_prior._engine_for(dev)
        lib = eng.lib
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        theta0 = pose.contiguous()
        bufs = [theta0.clone(), torch.empty_like(theta0)]
        m, v = torch.zeros_like(theta0), torch.zeros_like(theta0)
        q = torch.empty(N, 21, 4, device=dev, dtype=torch.float32)
        d = torch.empty(N, device=dev, dtype=torch.float32)
        dq = torch.empty(N, 21, 4, device=dev, dtype=torch.float32)
        if lib.pndf_aa2quat(bufs[0].data_ptr(), q.data_ptr(), N, stream) != 0:
            raise RuntimeError("pndf_aa2quat failed")
        if bm is not None:
            joints0 = bm.joints_of(theta0)                  # smpl_init.Jtr of the noisy poses (motion_denoise.py:60,63)
            g_body = torch.empty_like(theta0)
        k = 0
        from .engine import DenoiseWeights
        for it in range(iterations):
            pc, pp, tc, dc = iteration_coefs(self.schedule, it)
            w = DenoiseWeights(pc, pp, tc, dc)
            for _ in range(steps_per_iter):
                k += 1
                eng.forward_grad(q.data_ptr(), None, d.data_ptr(), dq.data_ptr(), N, stream.value or 0)
                if bm is not None:
                    bm.terms_grad(bufs[0], joints0, it, out=g_body, coefs=(w.temp_coef, w.data_coef))
                rc = lib.pndf_denoise_update_w(bufs[0].data_ptr(), bufs[1].data_ptr(), theta0.data_ptr(), d.data_ptr(),
                                               dq.data_ptr(), None if bm is None else g_body.data_ptr(), m.data_ptr(),
                                               v.data_ptr(), q.data_ptr(), S, T, ctypes.byref(w), k, float(lr), stream)
                if rc != 0:
                    raise RuntimeError(f"pndf_denoise_update failed ({rc})")
                bufs.reverse()
        return bufs[0]

    def optimize(self, noisy_poses, gt_poses=None, iterations=10, steps_per_iter=50, *, fused=None):
        """The reference's entry point, argument for argument (experiments/motion_denoise.py:58): optimise the poses and
        return the mean vertex-to-vertex error in cm as a 0-d numpy array -- against the ground-truth poses when given
        (:113-117), else against the meshes of the noisy input (:110).  Needs a body model with `.vertices` (the
        reference's always has one); the denoised poses are left in `self.last_poses`.  The HIP body model takes the fused
        driver (engine launch + fused LBS pass + Adam kernel per step), any other callable the autograd driver."""
        from .body_model import BodyModel
        if self.body_model is None:
            raise ValueError("optimize() returns the reference's v2v error and needs a body model; denoise() runs without one")
        if fused is None:
            fused = isinstance(self.body_model, BodyModel)
        noisy = noisy_poses.to(self.device, torch.float32)
        out, _ = self.denoise(noisy.reshape(-1, noisy.shape[-2], 69) if noisy.dim() > 2 else noisy.reshape(-1, 69),
                              iterations=iterations, steps_per_iter=steps_per_iter, record=False, fused=fused)
        self.last_poses = out.reshape(noisy.shape)
        ref = noisy if gt_poses is None else gt_poses.to(self.device, torch.float32)
        with torch.no_grad():
            def verts(p):      # the same two calling conventions as _geometry: keyword for BodyModel, positional for a plain callable
                flat = p.reshape(-1, 69)
                if isinstance(self.body_model, BodyModel):
                    return self.body_model(pose_body=flat).vertices
                res = self.body_model(flat)
                return res.vertices if hasattr(res, "vertices") else res[0]
            d = verts(self.last_poses) - verts(ref)
            v2v = torch.mean(torch.sqrt(torch.sum(d * d, dim=2))) * 100.0                # :118
        return v2v.detach().cpu().numpy()                                               # :120

    def denoise(self, noisy_poses, iterations=10, steps_per_iter=50, lr=0.02, record=True, fused=False):
        """noisy_poses: [T,69] or [S,T,69] axis-angle.  Returns (denoised poses, history of per-step mean losses;
        empty with fused=True)."""
        if fused:
            single = noisy_poses.dim() == 2
            pose = noisy_poses.to(self.device, torch.float32)
            out = self._optimize_fused(pose[None] if single else pose, iterations, steps_per_iter, lr)
            return (out[0] if single else out), []
        single = noisy_poses.dim() == 2
        pose = noisy_poses.to(self.device, torch.float32)
        if single:
            pose = pose[None]
        body_pose = pose.clone().requires_grad_(True)
        with torch.no_grad():
            _, init_joints = self._geometry(pose)
        opt = torch.optim.Adam([body_pose], lr, betas=(0.9, 0.999))                   # :70
        history = []
        for it in range(iterations):                                                  # :74
            for _ in range(steps_per_iter):                                           # :77
                opt.zero_grad()
                loss = self.losses(body_pose, init_joints, it)
                tot = self.total(loss, it).sum()          # sequences are independent: sum of per-sequence objectives
                tot.backward()                            # :98
                opt.step()                                # :99
                if record:                                # one host sync per step; off for throughput
                    history.append({k: float(v.detach().mean()) for k, v in loss.items()})
        out = body_pose.detach()
        return (out[0] if single else out), history


# ---- the script level of the reference (experiments/motion_denoise.py:108-121,124-153): motion files in, v2v error out
def load_motion_npz(path, device="cuda:0"):
    """`np.load(motion_file)['pose_body']` ([T,63] axis-angle of the 21 body joints) padded with the two (zero) hand joints
    to the body model's [T,69] (motion_denoise.py:134-138)."""
    import numpy as np
    pose_body = np.load(path)["pose_body"].astype(np.float32)
    if pose_body.ndim != 2 or pose_body.shape[1] not in (63, 69):
        raise ValueError(f"{path}: pose_body must be [T,63] or [T,69], got {pose_body.shape}")
    poses = torch.zeros((len(pose_body), 69), dtype=torch.float32)
    poses[:, :pose_body.shape[1]] = torch.from_numpy(pose_body)
    return poses.to(device)


@torch.no_grad()
def v2v_error_cm(body_model, poses, reference_poses):
    """mean vertex-to-vertex distance in cm between the meshes of two pose sequences (motion_denoise.py:111,117-119)."""
    v = body_model(pose_body=poses.reshape(-1, 69)).vertices
    r = body_model(pose_body=reference_poses.reshape(-1, 69)).vertices
    d = v - r
    return float(torch.mean(torch.sqrt(torch.sum(d * d, dim=2))) * 100.0)


def denoise_motion_file(posendf, body_model, motion_file, gt_file=None, device="cuda:0", iterations=10, steps_per_iter=50,
                        fused=True, schedule="motion_denoise"):
    """`main()` of experiments/motion_denoise.py:124-153 after the model is loaded: read the noisy motion, optimise it,
    return (denoised poses [T,69], v2v error in cm against the ground truth if given, else against the noisy input)."""
    noisy = load_motion_npz(motion_file, device)
    md = MotionDenoise(posendf, body_model, device=device, batch_size=len(noisy), schedule=schedule)
    gt = load_motion_npz(gt_file, device) if gt_file is not None else None
    v2v = md.optimize(noisy, gt, iterations, steps_per_iter, fused=fused)          # the reference's call (:152)
    return md.last_poses, float(v2v)
