#!/usr/bin/env python3
"""End-to-end sanity demo on a consistent synthetic scene (not a benchmark): images of an analytic scene (Gaussian
density blobs with position-dependent colour) are rendered from perturbed ground-truth poses with the HIP compositing
kernel; a barf_inn_llff model then trains from IDENTITY poses through the reference's call sequence.  Prints the
held-in PSNR and the Procrustes-aligned pose errors as training proceeds.

    python tools/teacher_student_demo.py [--steps 1500] [--views 8] [--size 48 64]
"""
import argparse
import sys
import time

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from neural_invertible_warp_amd import camera, configs, engine, evaluation, ops
from neural_invertible_warp_amd.util import edict


def analytic_scene(pts):
    """pts [..., 3] -> (rgb [..., 3], sigma [...]): a few Gaussian blobs in front of the cameras (+z)"""
    centers = torch.tensor([[0.0, 0.0, 3.0], [0.7, -0.3, 3.6], [-0.8, 0.4, 4.2], [0.2, 0.6, 2.6], [-0.3, -0.7, 3.2]], device=pts.device)
    radii = torch.tensor([0.55, 0.4, 0.5, 0.3, 0.35], device=pts.device)
    cols = torch.tensor([[0.9, 0.2, 0.2], [0.2, 0.8, 0.3], [0.2, 0.3, 0.9], [0.9, 0.8, 0.2], [0.7, 0.3, 0.8]], device=pts.device)
    d2 = ((pts[..., None, :] - centers) ** 2).sum(-1)                       # [..., 5]
    w = torch.exp(-0.5 * d2 / radii ** 2)
    sigma = 8.0 * w.sum(-1)
    tex = 0.5 + 0.5 * torch.sin(6.0 * pts[..., None, :] + torch.arange(5, device=pts.device)[:, None])      # [..., 5, 3] mild texture
    rgb = ((w[..., None] * cols * (0.6 + 0.4 * tex)).sum(-2) / (w.sum(-1, keepdim=True) + 1e-6)).clamp(0, 1)
    return rgb, sigma


@torch.no_grad()
def render_teacher(opt, pose, intr, S=192):
    if pose.shape[0] > 1 and pose.shape[0] * opt.H * opt.W * S > (1 << 27):    # image-scale scenes: one view at a time (the analytic scene
        return torch.cat([render_teacher(opt, pose[i:i + 1], intr[i:i + 1], S) for i in range(pose.shape[0])])    # materialises ~60 floats per sample)
    B = pose.shape[0]
    center, ray = camera.get_center_and_ray(opt, pose, intr=intr)          # [B,HW,3]
    depth = torch.linspace(1.0, 7.0, S, device=pose.device)
    pts = center[:, :, None] + ray[:, :, None] * depth[None, None, :, None]
    rgb_s, sigma = analytic_scene(pts)
    n = B * opt.H * opt.W
    rgb, _, opacity, _ = ops.composite(ray.reshape(n, 3), rgb_s.reshape(n, S, 3), sigma.reshape(n, S), depth.expand(n, S).contiguous())
    rgb = rgb + (1 - opacity[:, None]) * 0.1                                # dark grey background
    return rgb.view(B, opt.H, opt.W, 3).permute(0, 3, 1, 2).contiguous()


def run(steps=1500, views=8, size=(48, 64), device="cuda:0", log_every=250, seed=0, quiet=False, ga=2):
    H, W = size
    opt = configs.cfg3_barf_inn_llff(device=device, global_alignment=ga)
    opt.H, opt.W, opt.data.image_size = H, W, [H, W]
    opt.max_iter = steps
    opt.nerf.rand_rays, opt.nerf.sample_intvs = 2048, 64
    opt.inn.real_nvp.max_pe_iter = steps // 2
    opt.optim.test_photo = False
    gen = torch.Generator().manual_seed(seed)
    pose_GT = camera.lie.se3_to_SE3(torch.randn(views, 6, generator=gen) * torch.tensor([0.06, 0.06, 0.03, 0.15, 0.15, 0.05])).to(device)
    intr = torch.tensor([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]], dtype=torch.float32).repeat(views, 1, 1).to(device)
    image = render_teacher(opt, pose_GT, intr)
    var0 = edict(idx=torch.arange(views), image=image, intr=intr, pose=torch.eye(3, 4, device=device).repeat(views, 1, 1))
    tr = engine.INNTrainer(opt, views, seed=seed)
    ev = evaluation.LLFFEvaluator(opt, tr.graph, pose_GT)

    def report(it):
        pose, gt = ev.get_all_training_poses(opt)
        aligned, _ = ev.prealign_cameras(opt, pose, gt)
        err = ev.evaluate_camera_alignment(opt, aligned, gt)
        # gauge-free check: relative rotations between all pairs of views (independent of the centre-based sim3 alignment,
        # which is ill-conditioned for the small baselines of this scene)
        Rp, Rg = pose[:, :, :3], gt[:, :, :3]
        rel_p = Rp[:, None] @ Rp[None].transpose(-1, -2)
        rel_g = Rg[:, None] @ Rg[None].transpose(-1, -2)
        rel = camera.rotation_distance(rel_p, rel_g)
        n = pose.shape[0]
        hist_rel = float(rel.sum() / (n * n - n)) * 57.2958
        return float(err.R.mean()) * 57.2958, float(err.t.mean()), hist_rel

    hist = []
    t0 = time.perf_counter()
    for it in range(1, steps + 1):
        loss = tr.train_iteration(edict(var0))
        if it % log_every == 0 or it == 1:
            psnr = -10 * torch.log10(loss.render.detach()).item()
            r, t, rel = report(it)
            hist.append((it, psnr, r, t, rel))
            if not quiet:
                print(f"it {it:5d}  train PSNR {psnr:6.2f} dB  rot err {r:7.3f} deg  trans err {t:7.4f}  pairwise relative rot err {rel:6.3f} deg"
                      f"  ({time.perf_counter() - t0:.1f} s)", flush=True)
    return hist


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--size", type=int, nargs=2, default=[48, 64])
    ap.add_argument("--ga", type=float, default=2, help="loss_weight.global_alignment (log10); scripts/train_llff.sh uses 4")
    a = ap.parse_args()
    run(a.steps, a.views, tuple(a.size), ga=a.ga)
