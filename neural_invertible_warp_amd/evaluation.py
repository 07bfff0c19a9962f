"""Val / eval tooling around the Graph.

What the reference spreads over its `Model` classes (LLFF: model/barf_inn_llff.py:156-234 + nerf_inn_llff.py:193-228; DTU:
model/barf_inn_dtu.py:116-299, 370-382, 437-483 + nerf_inn_dtu.py:205-262) is organised here around three questions, with the
reference's method names kept as the entry points its `Model`s (and ours) call:

  1. where are the learnt training cameras?          `learnt_poses`            (get_all_training_poses)
  2. how do they relate to the ground truth?         `align` -> aligned poses + a similarity record the graph stores, from which
                                                     `Graph.get_pose` brings held-out ground-truth poses into the learnt frame
                                                     (prealign_cameras / prealign_w2c_*_camera_systems, validate)
  3. how well does a held-out view render?           `_score_view`: optional test-time pose refinement, full-image render on the HIP
                                                     path (graph.forward(mode="eval"): a sweep over pixel ranges), image / depth metrics
                                                     (evaluate_full, evaluate_test_time_photometric_optim)

The pose algebra is [N,3,4]-sized host work (posealign.py, align_trajectories.py); every render runs through libniw_hip.so.
"""
import math

import torch
from numpy.linalg import LinAlgError

from . import camera, metrics, posealign
from .align_trajectories import align_ate_c2b_use_a2b
from .util import edict


_detached = torch.no_grad()            # pose bookkeeping: never part of a graph


class _Evaluator:
    def __init__(self, opt, graph, pose_GT):
        """pose_GT: ground-truth world-to-camera poses of the training views [N,3,4] (train_data.get_all_camera_poses)"""
        self.opt, self.graph = opt, graph
        self.pose_GT = torch.as_tensor(pose_GT).to(opt.device)

    # ---- 3. held-out views
    def evaluate_test_time_photometric_optim(self, opt, var):
        """Test-time pose refinement (reference barf_inn_llff.py:218-234, barf_inn_dtu.py:467-483): the aligned ground-truth pose
        of a held-out view is still slightly off in the learnt frame; a 6-vector `var.se3_refine_test`, composed in front of it as
        exp(se3) by `Graph.get_pose(mode="test-optim")`, is fitted to the view's own pixels for `optim.test_iter` Adam steps
        (random rays, photometric loss only; the networks stay fixed)."""
        correction = torch.zeros(1, 6, device=opt.device, requires_grad=True)
        var.se3_refine_test = correction
        adam = getattr(torch.optim, opt.optim.algo)([correction], lr=opt.optim.lr_pose)
        for _ in range(opt.optim.test_iter):
            with torch.enable_grad():
                var.pose_refine_test = camera.lie.se3_to_SE3(correction)
                var = self.graph.forward(opt, var, mode="test-optim")
                photometric = self.graph.compute_loss(opt, var, mode="test-optim").render
                (correction.grad,) = torch.autograd.grad(photometric, correction)
            adam.step()
        return var

    def _score_view(self, opt, view):
        """-> (var after the eval render, rgb map [1,3,H,W], edict(psnr, ssim))"""
        var = edict(view)
        if opt.optim.test_photo:
            var = self.evaluate_test_time_photometric_optim(opt, var)
        with torch.no_grad():
            var = self.graph.forward(opt, var, mode="eval")
            rgb_map = var.rgb.reshape(-1, opt.H, opt.W, 3).movedim(-1, 1).contiguous()
            score = edict(psnr=metrics.psnr(rgb_map, var.image).item(), ssim=metrics.ssim(rgb_map, var.image).item())
        return var, rgb_map, score

    def _pose_errors_after_alignment(self, opt):
        self.graph.eval()
        return self.evaluate_camera_alignment(opt, self.validate(opt), self.get_all_training_poses(opt)[1])


# ------------------------------------------------------------------------------------------ LLFF
class LLFFEvaluator(_Evaluator):
    @_detached
    def get_all_training_poses(self, opt):
        """-> (learnt w2c poses, ground truth): the per-view rigid motions registered by the alignment loss (`global_rigid`),
        applied on top of the initial pose every view starts from (the identity: barf_inn_llff.py:156-169)"""
        learnt = self.graph.global_rigid.weight.detach().reshape(-1, 3, 4).clone()
        start = self.graph.pose_eye.expand_as(learnt)
        return camera.pose.compose([learnt, start]), self.pose_GT

    @_detached
    def prealign_cameras(self, opt, pose, pose_GT):
        """-> (learnt poses expressed in the ground-truth frame, sim3) (barf_inn_llff.py:171-187)"""
        pose, pose_GT = torch.as_tensor(pose).to(opt.device), torch.as_tensor(pose_GT).to(opt.device)
        sim3 = posealign.fit_sim3(pose, pose_GT)
        return posealign.transfer_poses(sim3, pose, to_gt=True), sim3

    @_detached
    def evaluate_camera_alignment(self, opt, pose_aligned, pose_GT):
        """-> edict(R [N] rad, t [N]) (barf_inn_llff.py:189-197)"""
        return posealign.pose_errors(torch.as_tensor(pose_aligned), torch.as_tensor(pose_GT))

    def validate(self, opt):
        """fit the similarity, leave it on the graph for `Graph.get_pose` -> the aligned learnt poses"""
        aligned, self.graph.sim3 = self.prealign_cameras(opt, *self.get_all_training_poses(opt))
        return aligned

    def evaluate_full(self, opt, test_views, eps=1e-10):
        """test_views: iterable of var dicts (idx, image [1,3,H,W], intr, pose) -> edict(error = per-view pose errors,
        res = [edict(psnr, ssim)], maps = rgb / inverse-depth images of the last view) (barf_inn_llff.py:199-216)"""
        error = self._pose_errors_after_alignment(opt)
        scores, maps = [], None
        for view in test_views:
            var, rgb_map, score = self._score_view(opt, view)
            # expected inverse depth of the opaque part of each ray (NDC depth runs 0..1 towards infinity)
            disparity = (1 - var.depth) / var.opacity if opt.camera.ndc else (var.depth / var.opacity + eps).reciprocal()
            maps = edict(rgb=rgb_map, invdepth=disparity.reshape(-1, 1, opt.H, opt.W))
            scores.append(score)
        return edict(error=error, res=scores, maps=maps)


# ------------------------------------------------------------------------------------------ DTU
def _frame_is_pinned(opt):
    """two or more poses held at their ground truth fix the gauge: the learnt frame IS the ground-truth frame"""
    return opt.pose.n_first_fixed_poses >= 2


def _no_alignment(device):
    return edict(R=torch.eye(3, device=device)[None], t=torch.zeros(1, 3, 1, device=device), s=1.0, type="traj_align")


class DTUEvaluator(_Evaluator):
    """graph: barf_inn_dtu.Graph (the poses live on graph.pose_net)"""

    @_detached
    def get_all_training_poses(self, opt):
        net = self.graph.pose_net
        return camera.pose.compose([net.get_w2c_poses(), net.initial_poses_w2c]), self.pose_GT

    @_detached
    def evaluate_camera_alignment(self, opt, pose_aligned_w2c, pose_GT_w2c):
        """-> edict(R: rotation angle [N] rad, t: distance between the camera CENTRES [N]) (barf_inn_dtu.py:139-170)"""
        centres = posealign.camera_centers(pose_aligned_w2c) - posealign.camera_centers(pose_GT_w2c)
        return edict(R=camera.rotation_distance(pose_aligned_w2c[..., :3], pose_GT_w2c[..., :3]), t=centres.norm(dim=-1))

    @_detached
    def prealign_w2c_large_camera_systems(self, opt, pose_w2c, pose_GT_w2c):
        """Umeyama similarity between the two trajectories of camera centres (barf_inn_dtu.py:173-201)
        -> (aligned w2c poses, est->gt similarity record)"""
        if _frame_is_pinned(opt):
            return pose_w2c, _no_alignment(opt.device)
        try:
            moved_c2w, record = align_ate_c2b_use_a2b(camera.pose.invert(pose_w2c), camera.pose.invert(pose_GT_w2c), method="sim3")
        except LinAlgError:                                # the SVD inside Umeyama did not converge
            return pose_w2c, _no_alignment(opt.device)
        record.type = "traj_align"
        return camera.pose.invert(moved_c2w[:, :3]), record

    @_detached
    def prealign_w2c_small_camera_systems(self, opt, pose_w2c, pose_GT_w2c):
        """Few cameras (< 10) do not determine a similarity robustly, so every ORDERED PAIR (a, b) of views proposes one -- scale from
        the ratio of the a-b baselines, rotation and translation by pinning camera a onto its ground truth -- and the proposal with the
        smallest (mean rotation error in degrees) x (mean centre error) wins (barf_inn_dtu.py:203-299).  All proposals are evaluated
        at once: [pairs, N] batched 4x4 algebra."""
        if _frame_is_pinned(opt):
            record = _no_alignment(opt.device)
            record.pop("type")
            return pose_w2c, record
        est, truth = camera.pad_poses(camera.pose.invert(pose_w2c)), camera.pad_poses(camera.pose.invert(pose_GT_w2c))
        n = min(est.shape[0], 10)
        views = torch.arange(n, device=est.device)
        a, b = (x.reshape(-1) for x in torch.meshgrid(views, views, indexing="ij"))
        a, b = a[a != b], b[a != b]
        centre = lambda T: T[..., :3, 3]
        scale = (centre(truth)[a] - centre(truth)[b]).norm(dim=-1) / (centre(est)[a] - centre(est)[b]).norm(dim=-1)        # [pairs]
        scaled = est[None].repeat(len(a), 1, 1, 1)
        scaled[..., :3, 3] *= scale[:, None, None]
        pin = truth[a] @ camera.pose_inverse_4x4(scaled[torch.arange(len(a), device=est.device), a])                                           # [pairs,4,4]
        moved_w2c = camera.pose_inverse_4x4(pin[:, None] @ scaled)[..., :3, :]                                              # [pairs,N,3,4]
        err = self.evaluate_camera_alignment(opt, moved_w2c, pose_GT_w2c[None])
        score = err.t.mean(dim=-1) * torch.rad2deg(err.R.mean(dim=-1))
        k = int(score.argmin())                                          # first minimum: the reference keeps the earliest best pair
        return moved_w2c[k], edict(R=pin[k, :3, :3][None], type="traj_align", t=pin[k, :3, 3].reshape(1, 3, 1), s=scale[k])

    def _prealign(self, opt, pose, pose_GT, large_above):
        many = pose.shape[0] > large_above
        return (self.prealign_w2c_large_camera_systems if many else self.prealign_w2c_small_camera_systems)(opt, pose, pose_GT)

    @_detached
    def evaluate_any_poses(self, opt, pose_w2c, pose_GT_w2c):
        """-> dict of mean rotation (degrees) / centre errors before and after the alignment (barf_inn_dtu.py:116-137)"""
        pose_w2c = pose_w2c.detach()
        report = {}
        for suffix, poses in (("_before_align", pose_w2c), ("", self._prealign(opt, pose_w2c, pose_GT_w2c, large_above=10)[0])):
            err = self.evaluate_camera_alignment(opt, poses, pose_GT_w2c)
            report["error_R" + suffix], report["error_t" + suffix] = torch.rad2deg(err.R.mean()), err.t.mean()
        return report

    def evaluate_poses(self, opt):
        return self.evaluate_any_poses(opt, *self.get_all_training_poses(opt))

    @_detached
    def validate(self, opt):
        """fit the est->gt similarity and leave it on the pose network, where the val / eval branch of Graph.get_pose reads it
        (barf_inn_dtu.py:370-382) -> the aligned learnt poses"""
        aligned, self.graph.pose_net.sim3_est_to_gt_c2w = self._prealign(opt, *self.get_all_training_poses(opt), large_above=9)
        return aligned

    def evaluate_full(self, opt, test_views):
        """-> edict(error, res = [edict(psnr, ssim, abs_err, rms_err)]); depth errors for views that carry depth_gt / valid_depth_gt,
        with the rendered depth brought to ground-truth units by the similarity's scale (nerf_inn_dtu.py:205-262)"""
        error = self._pose_errors_after_alignment(opt)
        scale = float(self.graph.pose_net.sim3_est_to_gt_c2w.s)
        scores = []
        for view in test_views:
            var, _, score = self._score_view(opt, view)
            score.abs_err = score.rms_err = math.nan
            if "depth_gt" in var and scale != 1.0:
                score.abs_err, score.rms_err = metrics.compute_depth_metrics(var, scale)
            scores.append(score)
        return edict(error=error, res=scores)
