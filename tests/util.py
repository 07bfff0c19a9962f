"""Shared helpers for the test-suite."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def t(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


def check_grad_summary(grad, g, key, rtol=1e-3):
    """Gradient fixtures hold the L2 norm and a strided sample (see make_golden.gsum).
    Tolerance: rtol * max|g| absolute on the sample, rtol relative on the norm."""
    assert grad is not None, key
    f = grad.detach().reshape(-1).double().cpu()
    norm, sample, stride = float(g[key + ".norm"]), g[key + ".sample"], int(g[key + ".stride"])
    mine = f[::stride].numpy()
    scale = max(float(np.abs(sample).max()), 1e-12)
    err = float(np.abs(mine - sample).max())
    assert err <= rtol * scale + 1e-7, f"{key}: sample err {err:.3e} scale {scale:.3e}"
    assert abs(float(f.norm()) - norm) <= rtol * max(norm, 1e-12) + 1e-7, f"{key}: norm {float(f.norm())} vs {norm}"


class TorchAlign:
    """torch restatement of the fused alignment operations (csrc/niw_align.hip), plugged into
    `nerf_inn_llff.ALIGN_BACKEND` by the CPU-only tests of the rank-sharded alignment term: same three stages (fp64 per-view
    moments -> optional reduction over ranks -> Kabsch solve; residual loss with the pose held constant)."""

    @staticmethod
    def rigid_registration(target, source, reduce_moments=None):
        x, y = target.detach().double(), source.detach().double()
        B, N = x.shape[:2]
        mom = torch.cat([torch.full((B, 1), float(N), dtype=torch.float64), x.sum(1), y.sum(1), (y.transpose(1, 2) @ x).reshape(B, 9)], dim=1)
        if reduce_moments is not None:
            reduce_moments(mom)
        n, xm, ym = mom[:, :1], mom[:, 1:4] / mom[:, :1], mom[:, 4:7] / mom[:, :1]
        M = mom[:, 7:].reshape(B, 3, 3) - n[:, :, None] * ym[:, :, None] * xm[:, None, :]
        U, _, Vt = torch.linalg.svd(M)
        det = torch.det(U @ Vt)
        R = U @ torch.diag_embed(torch.stack([torch.ones_like(det), torch.ones_like(det), det], dim=-1)) @ Vt
        t = ym - (R @ xm[:, :, None])[..., 0]
        return torch.cat([R, t[..., None]], dim=-1).float()

    @staticmethod
    def alignment_residual(target, source, poses, n_norm=None):
        from neural_invertible_warp_amd import camera
        e = target - camera.cam2world(source, poses.detach())
        return (e ** 2).sum() / float(n_norm if n_norm is not None else e.numel())
