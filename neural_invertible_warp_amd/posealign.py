"""Relating the learnt camera frame to the ground-truth frame (host algebra on [N,3,4] matrices).

The INN models learn poses up to a similarity of the world, so pose errors and held-out renders need the similarity
`sim3 = edict(t0, s0, t1, s1, R)` that `camera.procrustes_analysis(centres_gt, centres_learnt)` fits to the camera CENTRES:

        x_gt = s0 * ((x_learnt - t1) / s1) R^T + t0                                    (row vectors)

Everything the reference does with it (model/barf_inn_llff.py:171-197 pre-alignment of the learnt poses, :385-396 bringing a
ground-truth test pose into the learnt frame) is one operation, `transfer_poses`: move the camera centre through the map (or its
inverse), rotate the world axes by R (or R^T), and rebuild the world-to-camera translation t = -R_cam c.
"""
import torch

from . import camera
from .util import edict


def camera_centers(pose_w2c):
    """[...,3,4] world-to-camera -> camera centres [...,3] = -R^T t"""
    R, t = pose_w2c[..., :3], pose_w2c[..., 3]
    return -(R.transpose(-1, -2) @ t[..., None])[..., 0]


def identity_sim3(device):
    return edict(t0=torch.zeros(3, device=device), t1=torch.zeros(3, device=device), s0=torch.tensor(1.0, device=device),
                 s1=torch.tensor(1.0, device=device), R=torch.eye(3, device=device))


def fit_sim3(pose_learnt_w2c, pose_gt_w2c):
    """similarity between the two sets of camera centres; the identity when the Procrustes SVD fails to converge
    (reference barf_inn_llff.py:177-181)"""
    try:
        return camera.procrustes_analysis(camera_centers(pose_gt_w2c), camera_centers(pose_learnt_w2c))
    except Exception:      # torch.linalg.svd raises on non-convergence
        return identity_sim3(pose_gt_w2c.device)


def map_points(sim3, x, to_gt=True):
    if to_gt:
        return (x - sim3.t1) / sim3.s1 @ sim3.R.t() * sim3.s0 + sim3.t0
    return (x - sim3.t0) / sim3.s0 @ sim3.R * sim3.s1 + sim3.t1


def transfer_poses(sim3, pose_w2c, to_gt=True):
    """world-to-camera poses of one frame expressed in the other (to_gt: learnt -> ground truth; else ground truth -> learnt)"""
    centers = map_points(sim3, camera_centers(pose_w2c), to_gt=to_gt)
    R = pose_w2c[..., :3] @ (sim3.R.t() if to_gt else sim3.R)
    return camera.pose(R=R, t=-(R @ centers[..., None])[..., 0])


def pose_errors(pose_a_w2c, pose_b_w2c):
    """per-view rotation angle (rad) between the two rotations and distance between the two w2c translations"""
    angle = camera.rotation_distance(pose_a_w2c[..., :3], pose_b_w2c[..., :3])
    return edict(R=angle, t=(pose_a_w2c[..., 3] - pose_b_w2c[..., 3]).norm(dim=-1))
