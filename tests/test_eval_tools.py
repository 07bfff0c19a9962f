"""Val / eval tooling against golden vectors captured from the reference
(tests/golden/make_golden_eval.py -> eval_tools.npz): pose algebra, similarity alignment of the optimised
poses (LLFF Procrustes, DTU pairwise / Umeyama), test-pose back-alignment, depth and image metrics.
Host-side logic: runs without a GPU."""
import math
import os
import types

import numpy as np
import pytest
import torch

from neural_invertible_warp_amd import camera, evaluation, metrics
from neural_invertible_warp_amd.align_trajectories import backtrack_from_aligning_the_trajectory
from neural_invertible_warp_amd.model import barf_inn_llff
from neural_invertible_warp_amd.util import edict

G = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_tools.npz")).items()}


def close(a, b, tol=1e-5):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), err


def test_lie_exp_log():
    close(camera.lie.se3_to_SE3(G["lie_wu"]), G["lie_SE3"], 1e-6)
    close(camera.lie.so3_to_SO3(G["lie_wu"][:, :3]), G["lie_SO3"], 1e-6)
    close(camera.lie.SE3_to_se3(G["lie_SE3"]), G["lie_log"], 1e-5)
    close(camera.lie.SE3_to_se3(camera.lie.se3_to_SE3(G["lie_wu"]))[2:], G["lie_wu"][2:], 1e-5)      # round trip


def test_procrustes_and_rotation_distance():
    s = camera.procrustes_analysis(G["proc_X0"], G["proc_X1"])
    for k in ("t0", "t1", "s0", "s1", "R"):
        close(s[k], G["proc_" + k], 1e-6)
    close(camera.rotation_distance(G["rot_Ra"], G["rot_Rb"]), G["rot_dist"], 1e-6)


def test_llff_pose_evaluation():
    opt = edict(device="cpu", optim=edict(test_photo=False))
    ev = evaluation.LLFFEvaluator(opt, graph=None, pose_GT=G["llff_pose_GT"])
    aligned, sim3 = ev.prealign_cameras(opt, G["llff_pose_pred"], G["llff_pose_GT"])
    close(aligned, G["llff_pose_aligned"], 1e-5)
    for k in ("t0", "t1", "s0", "s1", "R"):
        close(sim3[k], G["llff_sim3_" + k], 1e-6)
    err = ev.evaluate_camera_alignment(opt, aligned, G["llff_pose_GT"])
    close(err.R, G["llff_err_R"], 1e-5)
    close(err.t, G["llff_err_t"], 1e-5)
    assert err.R.mean() < 0.05 and err.t.mean() < 0.05            # the similarity was recovered
    # val / eval branch of Graph.get_pose consumes the stored sim3 (barf_inn_llff.py:385-396)
    g = types.SimpleNamespace(sim3=sim3)
    pose = barf_inn_llff.Graph.get_pose(g, opt, edict(pose=G["llff_test_pose"]), mode="val")
    close(pose, G["llff_val_pose"], 1e-5)


@pytest.mark.parametrize("tag", ["dtu3", "dtu12"])
def test_dtu_pose_evaluation(tag):
    opt = edict(device="cpu", pose=edict(n_first_fixed_poses=0), optim=edict(test_photo=False))
    gt, pred = G[tag + "_gt"], G[tag + "_pred"]
    ev = evaluation.DTUEvaluator(opt, graph=None, pose_GT=gt)
    e0 = ev.evaluate_camera_alignment(opt, pred, gt)
    close(e0.R, G[tag + "_err0_R"], 1e-5)
    close(e0.t, G[tag + "_err0_t"], 1e-5)
    fn = ev.prealign_w2c_small_camera_systems if gt.shape[0] <= 10 else ev.prealign_w2c_large_camera_systems
    aligned, sim = fn(opt, pred, gt)
    close(aligned, G[tag + "_aligned"], 2e-5)
    close(sim.R, G[tag + "_sim_R"], 1e-5)
    close(sim.t, G[tag + "_sim_t"], 1e-5)
    close(float(sim.s), G[tag + "_sim_s"], 1e-5)
    e1 = ev.evaluate_camera_alignment(opt, aligned, gt)
    close(e1.R, G[tag + "_err1_R"], 1e-4)
    close(e1.t, G[tag + "_err1_t"], 1e-4)
    stats = ev.evaluate_any_poses(opt, pred, gt)
    close(torch.stack([torch.as_tensor(float(stats[k])) for k in ("error_R_before_align", "error_t_before_align", "error_R", "error_t")]),
          G[tag + "_stats"], 1e-4)
    assert float(stats["error_t"]) < float(stats["error_t_before_align"])
    close(backtrack_from_aligning_the_trajectory(G[tag + "_test_pose"], sim), G[tag + "_test_back"], 1e-5)


def test_depth_metrics():
    base = dict(idx=G["dm_idx"], depth_gt=G["dm_depth_gt"], valid_depth_gt=G["dm_valid"], depth=G["dm_depth"])
    a1, r1 = metrics.compute_depth_error_on_rays(edict(ray_idx=G["dm_ray1"], **base), 1.3)
    a2, r2 = metrics.compute_depth_error_on_rays(edict(ray_idx=G["dm_ray2"], **base), 1.0)
    close(torch.stack([a1, r1, a2, r2]), G["dm_on_rays"], 1e-6)
    v = edict(depth=G["dm_full"], depth_gt=G["dm_depth_gt"][None, 1], valid_depth_gt=G["dm_valid"][None, 1])
    got = [*metrics.compute_depth_error(v, 1.), *metrics.compute_depth_error(v, 0.9), *metrics.compute_depth_metrics(v, 1.1)]
    close(torch.tensor(got, dtype=torch.float64), G["dm_full_out"], 1e-6)


def test_image_metrics():
    close(metrics.ssim(G["im_a"], G["im_b"]), G["im_ssim"], 1e-6)
    close(metrics.ssim(G["im_a"], G["im_b"], size_average=False), G["im_ssim_each"], 1e-6)
    close(metrics.psnr(G["im_a"], G["im_b"]), G["im_psnr"], 1e-6)


def test_dtu_small_system_alignment_is_the_best_pairwise_proposal():
    """the batched [pairs, N] evaluation picks the same proposal as trying the ordered pairs one at a time"""
    torch.manual_seed(3)
    n = 6
    gt = camera.pose(R=camera.lie.so3_to_SO3(torch.randn(n, 3) * 0.4), t=torch.randn(n, 3))
    sim_R, sim_t = camera.lie.so3_to_SO3(torch.randn(3) * 0.7), torch.randn(3)
    c2w = camera.pose.invert(gt)
    moved = torch.cat([sim_R @ c2w[..., :3] @ camera.lie.so3_to_SO3(torch.randn(n, 3) * 0.02),
                       (0.6 * (sim_R @ c2w[..., 3:]) + sim_t[:, None]) + 0.01 * torch.randn(n, 3, 1)], dim=-1)
    pred = camera.pose.invert(moved)
    opt = edict(device="cpu", pose=edict(n_first_fixed_poses=0), optim=edict(test_photo=False))
    ev = evaluation.DTUEvaluator(opt, graph=None, pose_GT=gt)
    aligned, sim = ev.prealign_w2c_small_camera_systems(opt, pred, gt)
    best = None
    E, T = camera.pad_poses(camera.pose.invert(pred)), camera.pad_poses(camera.pose.invert(gt))
    for a in range(n):
        for b in range(n):
            if a != b:
                s = (T[a, :3, 3] - T[b, :3, 3]).norm() / (E[a, :3, 3] - E[b, :3, 3]).norm()
                Es = E.clone()
                Es[:, :3, 3] *= s
                cand = camera.pose_inverse_4x4((T[a] @ camera.pose_inverse_4x4(Es[a]))[None] @ Es)[:, :3]
                err = ev.evaluate_camera_alignment(opt, cand, gt)
                score = err.t.mean().item() * math.degrees(err.R.mean().item())
                if best is None or score < best[0]:
                    best = (score, cand, s)
    close(aligned, best[1], 1e-5)
    close(float(sim.s), float(best[2]), 1e-6)
    assert abs(float(sim.s) - 1 / 0.6) < 0.1
