"""Evaluation metrics of the val / eval path: rendered-depth errors (mirror of the
reference's core/metrics.py:4-119), PSNR as the engines compute it (nerf_inn_llff.py:214) and the SSIM of
external/pohsun_ssim (Gaussian 11x11 window, sigma 1.5) that evaluate_full reports (:215).

LPIPS (nerf_inn_llff.py:216) needs the pretrained AlexNet/LPIPS weights, which are not available offline:
not provided.  Everything here is image-sized host-side torch arithmetic on rendered outputs.
"""
import math

import torch
import torch.nn.functional as F


def compute_rmse(prediction, target):
    return torch.sqrt((prediction - target).pow(2).mean())


def _abs_rmse(depth_gt, depth):
    abs_e = torch.abs(depth_gt - depth)
    return abs_e.sum() / (abs_e.nelement() + 1e-6), compute_rmse(depth_gt, depth)


def compute_depth_error_on_rays(var, scaling_factor_for_pred_depth=1.):
    """Depth error at the rendered rays (reference core/metrics.py:4-59).  var: idx [B], depth_gt /
    valid_depth_gt [N,H,W], depth [B,R,1], optional ray_idx [R] or [B,R]."""
    B = len(var.idx)
    depth_gt = var.depth_gt[var.idx].view(B, -1, 1)
    valid = var.valid_depth_gt[var.idx].view(B, -1, 1)
    if "ray_idx" in var.keys():
        ray_idx = var.ray_idx
        if ray_idx.dim() == 2 and ray_idx.shape[0] == B:          # a different pixel set per image
            gather = ray_idx.long().unsqueeze(-1)
            depth_gt, valid = depth_gt.gather(1, gather), valid.gather(1, gather)
        else:
            depth_gt, valid = depth_gt[:, ray_idx], valid[:, ray_idx]
    return _abs_rmse(depth_gt[valid], var.depth[valid] * scaling_factor_for_pred_depth)


def compute_depth_error(var, scaling_factor_for_pred_depth=1.):
    """Full-image depth error of the first rendered view -> (abs, rmse) floats; with a scaling factor the
    better of scaled / unscaled is reported (reference core/metrics.py:64-111)."""
    pred = var.depth.view(1, -1, 1)
    depth_gt = var.depth_gt[0].view(1, -1, 1)
    valid = var.valid_depth_gt[0].view(1, -1)
    depth_gt, pred = depth_gt[valid], pred[valid]
    abs_e, rmse = (x.item() for x in _abs_rmse(depth_gt, pred))
    if scaling_factor_for_pred_depth != 1.:
        abs_s, rmse_s = (x.item() for x in _abs_rmse(depth_gt, pred * scaling_factor_for_pred_depth))
        abs_e, rmse = min(abs_e, abs_s), min(rmse, rmse_s)
    return abs_e, rmse


def compute_depth_metrics(var, scaling_factor_for_pred_depth):
    """reference core/metrics.py:114-119"""
    a0, r0 = compute_depth_error(var, scaling_factor_for_pred_depth=1)
    a1, r1 = compute_depth_error(var, scaling_factor_for_pred_depth=scaling_factor_for_pred_depth)
    return min(a0, a1), min(r0, r1)


def psnr(pred, target):
    """-10 log10(mean squared error) over all elements (nerf_inn_llff.py:214)"""
    return -10 * ((pred.contiguous() - target) ** 2).mean().log10()


def _gaussian_window(size, sigma, channels, like):
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)])
    g = (g / g.sum()).unsqueeze(1)
    return (g @ g.t()).float()[None, None].expand(channels, 1, size, size).contiguous().to(like)


def ssim(img1, img2, window_size=11, size_average=True):
    """Structural similarity of [B,C,H,W] images in [0,1] (Wang et al. 2004; constants and window of the
    reference's external/pohsun_ssim/pytorch_ssim/__init__.py:7-40)."""
    C = img1.shape[1]
    w = _gaussian_window(window_size, 1.5, C, img1)
    blur = lambda x: F.conv2d(x, w, padding=window_size // 2, groups=C)
    mu1, mu2 = blur(img1), blur(img2)
    s11 = blur(img1 * img1) - mu1 * mu1
    s22 = blur(img2 * img2) - mu2 * mu2
    s12 = blur(img1 * img2) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))
    return m.mean() if size_average else m.mean(dim=(1, 2, 3))
