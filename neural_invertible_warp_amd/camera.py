"""Host mirror of the reference's camera.py for the functions on the render path.

Ray generation runs in libniw_hip.so (niw_raygen / niw_convert_ndc).  The pose algebra kept
here (Pose, cam2world on [B,N,3] point sets) is host glue on [B,3,4] matrices used by the
alignment loss and evaluation poses, not part of the per-sample path.
"""
import torch

from . import ops


def to_hom(X):
    """reference camera.py:330-333"""
    return torch.cat([X, torch.ones_like(X[..., :1])], dim=-1)


class Pose:
    """[R|t] operations, reference camera.py:64-112."""

    def __call__(self, R=None, t=None):
        assert R is not None or t is not None
        if R is None:
            t = torch.as_tensor(t)
            R = torch.eye(3, device=t.device).repeat(*t.shape[:-1], 1, 1)
        elif t is None:
            R = torch.as_tensor(R)
            t = torch.zeros(R.shape[:-1], device=R.device)
        R, t = torch.as_tensor(R).float(), torch.as_tensor(t).float()
        assert R.shape[:-1] == t.shape and R.shape[-2:] == (3, 3)
        return torch.cat([R, t[..., None]], dim=-1)

    def invert(self, pose, use_inverse=False):
        R, t = pose[..., :3], pose[..., 3:]
        R_inv = R.inverse() if use_inverse else R.transpose(-1, -2)
        return self(R=R_inv, t=(-R_inv @ t)[..., 0])

    def compose_pair(self, pose_a, pose_b):
        R_a, t_a = pose_a[..., :3], pose_a[..., 3:]
        R_b, t_b = pose_b[..., :3], pose_b[..., 3:]
        return self(R=R_b @ R_a, t=(R_b @ t_a + t_b)[..., 0])

    def compose(self, pose_list):
        out = pose_list[0]
        for p in pose_list[1:]:
            out = self.compose_pair(out, p)
        return out


pose = Pose()


def cam2world(X, pose_w2c):
    """reference camera.py:343-346 (pose is world->camera)."""
    return to_hom(X) @ pose.invert(pose_w2c).transpose(-1, -2)


def world2cam(X, pose_w2c):
    """reference camera.py:335-337"""
    return to_hom(X) @ pose_w2c.transpose(-1, -2)


def get_unwarped_center_and_ray(opt, intr=None, ray_idx=None, pose_init=None):
    """reference camera.py:359-390 -> (center_3D, grid_3D), each [B,R,3] (R = H*W without ray_idx).
    Only the requested pixels are generated (the reference builds all H*W and indexes)."""
    assert opt.camera.model == "perspective"
    return ops.raygen(intr, pose_init, ray_idx, opt.H, opt.W, 0)


def get_center_and_ray(opt, pose, intr=None, ray_idx=None):
    """reference camera.py:419-443 -> (center_3D, ray), each [B,R,3].  `ray_idx` (an extension
    of the reference signature) restricts generation to the pixels the caller would index."""
    assert opt.camera.model == "perspective"
    return ops.raygen(intr, pose, ray_idx, opt.H, opt.W, 1)


def get_3D_points_from_depth(opt, center, ray, depth, multi_samples=False):
    """reference camera.py:517-521.  Kept for interface parity; the render path generates the
    sample points inside the field-MLP kernel (niw_mlp_fwd) and never materialises them."""
    if multi_samples:
        center, ray = center[:, :, None], ray[:, :, None]
    return center + ray * depth


def convert_NDC(opt, center, ray, intr, near=1):
    """reference camera.py:523-540"""
    return ops.convert_ndc(center, ray, intr, near)
