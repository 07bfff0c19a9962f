"""Trajectory alignment used by the DTU pose evaluation: mirror of the functions the
reference's model/barf_inn_dtu.py imports from align_trajectories.py (`align_ate_c2b_use_a2b` :89-140,
`backtrack_from_aligning_the_trajectory` :62-69) and of the Umeyama solver they call
(third_party/ATE/align_trajectory.py `align_umeyama`, method 'sim3' of `alignTrajectory`).

Host-side algebra on [N,3,4] pose matrices; nothing here touches the per-sample path.
"""
import numpy as np
import torch

from . import camera
from .util import edict


def align_umeyama(model, data, known_scale=False):
    """Least-squares similarity  model ~ s * R @ data + t  (Umeyama 1991) on [n,3] numpy arrays.
    Follows the reference solver's conventions: population covariance, degenerate spread (variance < 1e-5)
    falls back to unit scale, scale denominator regularised by 1e-6."""
    mu_m, mu_d = model.mean(0), data.mean(0)
    mc, dc = model - mu_m, data - mu_d
    n = model.shape[0]
    cov = mc.T @ dc / n
    var_d = (dc * dc).sum() / n
    degenerate = var_d < 1e-5
    if degenerate:
        var_d = 1.0
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt.T) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    s = 1 if (known_scale or degenerate) else np.trace(np.diag(D) @ S) / (var_d + 1e-6)
    t = mu_m - s * R @ mu_d
    return s, R, t


def convert3x4_4x4(x):
    """[N,3,4] -> [N,4,4] (numpy)"""
    out = np.concatenate([x, np.zeros_like(x[:, 0:1])], axis=1)
    out[:, 3, 3] = 1.0
    return out


def align_ate_c2b_use_a2b(traj_a_c2w, traj_b_c2w, traj_c=None, method="sim3"):
    """Align trajectory c to b with the sim3 that maps a onto b -> ([N1,4,4] aligned c2w, edict(R [1,3,3],
    t [1,3,1], s float)).  Only the 'sim3' method is used by the models (barf_inn_dtu.py:195)."""
    if method != "sim3":
        raise ValueError("align_ate_c2b_use_a2b: only method='sim3' is used on this path")
    device = traj_a_c2w.device
    if traj_c is None:
        traj_c = traj_a_c2w.clone()
    a, b, c = (x.float().cpu().numpy() for x in (traj_a_c2w, traj_b_c2w, traj_c))
    s, R, t = align_umeyama(b[:, :3, 3], a[:, :3, 3])      # gt = s R est + t
    R = R[None].astype(np.float32)
    t = t[None, :, None].astype(np.float32)
    s = float(s)
    R_c = R @ c[:, :3, :3]
    t_c = s * (R @ c[:, :3, 3:4]) + t
    aligned = convert3x4_4x4(np.concatenate([R_c, t_c], axis=2))
    return torch.from_numpy(aligned).to(device), edict(R=torch.from_numpy(R).to(device), t=torch.from_numpy(t).to(device), s=s)


def backtrack_from_aligning_the_trajectory(pose_GT_w2c, ssim_est_gt_c2w):
    """Bring ground-truth (test) w2c poses into the optimised poses' frame by undoing the est->gt
    similarity (reference align_trajectories.py:62-69)."""
    pose_GT_c2w = camera.pose.invert(pose_GT_w2c)
    Rt = ssim_est_gt_c2w.R.transpose(-2, -1)
    R_al = Rt @ pose_GT_c2w[:, :3, :3]
    t_al = Rt / ssim_est_gt_c2w.s @ (pose_GT_c2w[:, :3, 3:4] - ssim_est_gt_c2w.t)
    return camera.pose.invert(camera.pose(R=R_al, t=t_al.reshape(-1, 3)))
