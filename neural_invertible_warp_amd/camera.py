"""Host mirror of the reference's camera.py for the functions on the render path.

Ray generation runs in libniw_hip.so (niw_raygen / niw_convert_ndc).  The pose algebra kept
here (Pose, cam2world on [B,N,3] point sets) is host glue on [B,3,4] matrices used by the
alignment loss and evaluation poses, not part of the per-sample path.
"""
import math

import torch

from . import ops
from .util import edict


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


class Lie:
    """so(3) / se(3) exponential and logarithm with the closed-form coefficients evaluated as truncated
    power series in theta^2 (terms up to theta^20), reference camera.py:193-274.  Host-side [.,6]-sized
    algebra (test-time pose refinement, noisy initial poses); not on the per-sample path."""
    _TERMS = 11

    @staticmethod
    def _series(theta, first_factorial):
        # sum_i (-1)^i theta^(2i) / (2i + first_factorial)!   evaluated by Horner in theta^2
        x2 = theta * theta
        coef = [(-1.0) ** i / math.factorial(2 * i + first_factorial) for i in range(Lie._TERMS)]
        acc = torch.full_like(theta, coef[-1])
        for c in reversed(coef[:-1]):
            acc = acc * x2 + c
        return acc

    def taylor_A(self, x):   # sin(x)/x
        return self._series(x, 1)

    def taylor_B(self, x):   # (1-cos(x))/x^2
        return self._series(x, 2)

    def taylor_C(self, x):   # (x-sin(x))/x^3
        return self._series(x, 3)

    def skew_symmetric(self, w):
        w0, w1, w2 = w.unbind(dim=-1)
        z = torch.zeros_like(w0)
        return torch.stack([torch.stack([z, -w2, w1], -1), torch.stack([w2, z, -w0], -1), torch.stack([-w1, w0, z], -1)], -2)

    def so3_to_SO3(self, w):
        wx = self.skew_symmetric(w)
        theta = w.norm(dim=-1)[..., None, None]
        eye = torch.eye(3, device=w.device, dtype=torch.float32)
        return eye + self.taylor_A(theta) * wx + self.taylor_B(theta) * wx @ wx

    def SO3_to_so3(self, R, eps=1e-7):
        trace = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
        theta = ((trace - 1) / 2).clamp(-1 + eps, 1 - eps).acos()[..., None, None] % math.pi
        lnR = 1 / (2 * self.taylor_A(theta) + 1e-8) * (R - R.transpose(-2, -1))
        return torch.stack([lnR[..., 2, 1], lnR[..., 0, 2], lnR[..., 1, 0]], dim=-1)

    def se3_to_SE3(self, wu):
        w, u = wu.split([3, 3], dim=-1)
        wx = self.skew_symmetric(w)
        theta = w.norm(dim=-1)[..., None, None]
        eye = torch.eye(3, device=w.device, dtype=torch.float32)
        A, B, C = self.taylor_A(theta), self.taylor_B(theta), self.taylor_C(theta)
        wx2 = wx @ wx
        R = eye + A * wx + B * wx2
        V = eye + B * wx + C * wx2
        return torch.cat([R, V @ u[..., None]], dim=-1)

    def SE3_to_se3(self, Rt, eps=1e-8):
        R, t = Rt.split([3, 1], dim=-1)
        w = self.SO3_to_so3(R)
        wx = self.skew_symmetric(w)
        theta = w.norm(dim=-1)[..., None, None]
        eye = torch.eye(3, device=w.device, dtype=torch.float32)
        A, B = self.taylor_A(theta), self.taylor_B(theta)
        invV = eye - 0.5 * wx + (1 - A / (2 * B)) / (theta ** 2 + eps) * wx @ wx
        return torch.cat([w, (invV @ t)[..., 0]], dim=-1)


pose = Pose()
lie = Lie()


def pad_poses(p):
    """[...,3,4] -> [...,4,4] with the homogeneous row (reference utils/camera.py:24-29)"""
    bottom = torch.tensor([0, 0, 0, 1.0], device=p.device, dtype=p.dtype).expand(p[..., :1, :4].shape)
    return torch.cat((p[..., :3, :4], bottom), dim=-2)


def pose_inverse_4x4(mat, use_inverse=False):
    """Rigid inverse of [B,4,4] / [4,4] matrices (reference camera.py:34-61)"""
    R, t = mat[..., :3, :3], mat[..., :3, 3:]
    R_inv = R.inverse() if use_inverse else R.transpose(-1, -2)
    out = torch.zeros_like(mat)
    out[..., :3, :3] = R_inv
    out[..., :3, 3:] = -R_inv @ t
    out[..., 3, 3] = 1
    return out


def rotation_distance(R1, R2, eps=1e-7):
    """Geodesic angle between rotations (reference camera.py:542-547)"""
    R_diff = R1 @ R2.transpose(-2, -1)
    trace = R_diff[..., 0, 0] + R_diff[..., 1, 1] + R_diff[..., 2, 2]
    return ((trace - 1) / 2).clamp(-1 + eps, 1 - eps).acos()


def procrustes_analysis(X0, X1):
    """Similarity transform between two [N,3] point sets, X1to0 = (X1-t1)/s1 @ R.t() * s0 + t0
    (reference camera.py:549-566; rotation from a double-precision SVD)."""
    t0, t1 = X0.mean(dim=0, keepdim=True), X1.mean(dim=0, keepdim=True)
    X0c, X1c = X0 - t0, X1 - t1
    s0 = (X0c ** 2).sum(dim=-1).mean().sqrt()
    s1 = (X1c ** 2).sum(dim=-1).mean().sqrt()
    U, _, Vh = torch.linalg.svd(((X0c / s0).t() @ (X1c / s1)).double(), full_matrices=False)
    R = (U @ Vh).float()
    if R.det() < 0:
        R[2] *= -1
    return edict(t0=t0[0], t1=t1[0], s0=s0, s1=s1, R=R)


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


def get_center_and_ray(opt, pose, intr=None, ray_idx=None, pixel_range=None):
    """reference camera.py:419-443 -> (center_3D, ray), each [B,R,3].  `ray_idx` / `pixel_range` = (first, count) (extensions
    of the reference signature) restrict generation to the pixels the caller would index."""
    assert opt.camera.model == "perspective"
    if torch.is_grad_enabled() and pose.requires_grad:
        # pose optimisation (test-time photometric refinement, barf_inn_llff.py:218-234): camera-frame grid
        # from the kernel, then the reference's cam2world algebra so that autograd reaches the pose
        center_cam, grid_cam = ops.raygen(intr, None, ray_idx, opt.H, opt.W, 0, pixel_range=pixel_range)
        center, grid = cam2world(center_cam, pose), cam2world(grid_cam, pose)
        return center, grid - center
    return ops.raygen(intr, pose, ray_idx, opt.H, opt.W, 1, pixel_range=pixel_range)


def get_3D_points_from_depth(opt, center, ray, depth, multi_samples=False):
    """reference camera.py:517-521.  Kept for interface parity; the render path generates the
    sample points inside the field-MLP kernel (niw_mlp_fwd) and never materialises them."""
    if multi_samples:
        center, ray = center[:, :, None], ray[:, :, None]
    return center + ray * depth


def convert_NDC(opt, center, ray, intr, near=1):
    """reference camera.py:523-540: rays re-parametrised in normalised device coordinates (near plane at z = `near`, +z forward).
    One launch (niw_convert_ndc); when a gradient can flow into `center` / `ray` (warped rays in training, a refined pose at test time)
    its reverse pass is niw_convert_ndc_bwd (round 6; rounds 1-5 ran the formulas as torch algebra there) -- autograd reaches the warp /
    the pose either way.  Inputs that are not float32 device tensors, or intrinsics that want a gradient, keep the torch algebra."""
    on_device = center.is_cuda and center.dtype == torch.float32 and ray.dtype == torch.float32 and not intr.requires_grad
    if torch.is_grad_enabled() and (center.requires_grad or ray.requires_grad) and not on_device:
        sx = (intr[:, 0, 0] / intr[:, 0, 2])[:, None]                   # focal / principal point, per view
        sy = (intr[:, 1, 1] / intr[:, 1, 2])[:, None]
        cx, cy, cz = center.unbind(dim=-1)
        rx, ry, rz = ray.unbind(dim=-1)
        shift = (near - cz) / rz                                         # slide every origin onto the near plane
        cx, cy, cz = cx + shift * rx, cy + shift * ry, cz + shift * rz
        center_ndc = torch.stack([sx * (cx / cz), sy * (cy / cz), 1 - 2 * near / cz], dim=-1)
        ray_ndc = torch.stack([sx * (rx / rz - cx / cz), sy * (ry / rz - cy / cz), 2 * near / cz], dim=-1)
        return center_ndc, ray_ndc
    return ops.convert_ndc(center, ray, intr, near)
