"""Val / eval tooling around the Graph: pose evaluation up to a similarity transform,
the test-pose alignment the val / eval branch of `Graph.get_pose` consumes, full-image rendering through
`render_by_slices` and the image / depth metrics.  Mirrors the evaluation methods of the reference's
`Model` classes, taking the graph and the ground-truth poses explicitly instead of a dataset object:

  LLFF (model/barf_inn_llff.py):  get_all_training_poses :156-169, prealign_cameras :171-187,
        evaluate_camera_alignment :189-197, evaluate_full :199-216 (+ nerf_inn_llff.py:193-228),
        evaluate_test_time_photometric_optim :218-234
  DTU  (model/barf_inn_dtu.py):   evaluate_any_poses :116-137, evaluate_camera_alignment :139-170,
        prealign_w2c_large_camera_systems :173-201, prealign_w2c_small_camera_systems :203-299,
        validate :370-382, evaluate_full :437-465 (+ nerf_inn_dtu.py:205-262)

The renders run on the HIP path (graph.forward(mode="eval")); the pose algebra is [N,3,4]-sized host work.
"""
import numpy as np
import torch

from . import camera, metrics, posealign
from .align_trajectories import align_ate_c2b_use_a2b
from .util import edict


class _TestTimeRefinement:
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


# ------------------------------------------------------------------------------------------ LLFF
class LLFFEvaluator(_TestTimeRefinement):
    def __init__(self, opt, graph, pose_GT):
        """pose_GT: ground-truth w2c poses of the training views [N,3,4] (train_data.get_all_camera_poses)"""
        self.opt, self.graph = opt, graph
        self.pose_GT = torch.as_tensor(pose_GT).to(opt.device)

    @torch.no_grad()
    def get_all_training_poses(self, opt):
        """-> (learnt w2c poses, ground truth): the per-view rigid motions registered by the alignment loss (`global_rigid`),
        applied on top of the initial pose every view starts from (the identity: barf_inn_llff.py:156-169)"""
        learnt = self.graph.global_rigid.weight.detach().reshape(-1, 3, 4).clone()
        start = self.graph.pose_eye.expand_as(learnt)
        return camera.pose.compose([learnt, start]), self.pose_GT

    @torch.no_grad()
    def prealign_cameras(self, opt, pose, pose_GT):
        """-> (learnt poses expressed in the ground-truth frame, sim3) (barf_inn_llff.py:171-187)"""
        pose, pose_GT = torch.as_tensor(pose).to(opt.device), torch.as_tensor(pose_GT).to(opt.device)
        sim3 = posealign.fit_sim3(pose, pose_GT)
        return posealign.transfer_poses(sim3, pose, to_gt=True), sim3

    @torch.no_grad()
    def evaluate_camera_alignment(self, opt, pose_aligned, pose_GT):
        """-> edict(R [N] rad, t [N]) (barf_inn_llff.py:189-197)"""
        return posealign.pose_errors(torch.as_tensor(pose_aligned), torch.as_tensor(pose_GT))

    def evaluate_full(self, opt, test_views, eps=1e-10):
        """test_views: iterable of var dicts (idx, image [1,3,H,W], intr, pose) -> edict(error=pose errors,
        res=[edict(psnr, ssim)], rgb / invdepth maps of the last view)"""
        self.graph.eval()
        pose, pose_GT = self.get_all_training_poses(opt)
        pose_aligned, self.graph.sim3 = self.prealign_cameras(opt, pose, pose_GT)
        error = self.evaluate_camera_alignment(opt, pose_aligned, pose_GT)
        res, maps = [], None
        for var in test_views:
            var = edict(var)
            if opt.optim.test_photo:
                var = self.evaluate_test_time_photometric_optim(opt, var)
            with torch.no_grad():
                var = self.graph.forward(opt, var, mode="eval")
                invdepth = (1 - var.depth) / var.opacity if opt.camera.ndc else 1 / (var.depth / var.opacity + eps)
                rgb_map = var.rgb.view(-1, opt.H, opt.W, 3).permute(0, 3, 1, 2)
                invdepth_map = invdepth.view(-1, opt.H, opt.W, 1).permute(0, 3, 1, 2)
                res.append(edict(psnr=metrics.psnr(rgb_map, var.image).item(), ssim=metrics.ssim(rgb_map.contiguous(), var.image).item()))
                maps = edict(rgb=rgb_map, invdepth=invdepth_map)
        return edict(error=error, res=res, maps=maps)


# ------------------------------------------------------------------------------------------ DTU
class DTUEvaluator(_TestTimeRefinement):
    def __init__(self, opt, graph, pose_GT):
        """graph: barf_inn_dtu.Graph (poses live on graph.pose_net); pose_GT: training w2c poses [N,3,4]"""
        self.opt, self.graph, self.pose_GT = opt, graph, pose_GT.to(opt.device)

    @torch.no_grad()
    def get_all_training_poses(self, opt):
        net = self.graph.pose_net
        return camera.pose.compose([net.get_w2c_poses(), net.initial_poses_w2c]), self.pose_GT

    @torch.no_grad()
    def evaluate_camera_alignment(self, opt, pose_aligned_w2c, pose_GT_w2c):
        """Rotation (rad) and camera-centre (world frame) errors, not averaged"""
        a, g = camera.pose.invert(pose_aligned_w2c), camera.pose.invert(pose_GT_w2c)
        return edict(R=camera.rotation_distance(a[..., :3], g[..., :3]), t=(a[..., 3] - g[..., 3]).norm(dim=-1))

    @torch.no_grad()
    def prealign_w2c_large_camera_systems(self, opt, pose_w2c, pose_GT_w2c):
        """sim3 trajectory alignment (Umeyama on the camera centres) -> (aligned w2c poses, est->gt similarity)"""
        identity = edict(R=torch.eye(3, device=opt.device).unsqueeze(0), t=torch.zeros(1, 3, 1, device=opt.device), s=1., type="traj_align")
        if opt.pose.n_first_fixed_poses > 1:
            return pose_w2c, identity
        try:
            aligned_c2w, sim = align_ate_c2b_use_a2b(camera.pose.invert(pose_w2c), camera.pose.invert(pose_GT_w2c), method="sim3")
            sim.type = "traj_align"
            return camera.pose.invert(aligned_c2w[:, :3]), sim
        except np.linalg.LinAlgError:
            return pose_w2c, identity

    @torch.no_grad()
    def prealign_w2c_small_camera_systems(self, opt, pose_w2c, pose_GT_w2c):
        """For < 10 views: try every ordered pair (a, b) - scale from the a-b baseline, rigid transform from
        view a - and keep the candidate with the smallest (mean rot deg x mean trans) error."""
        if opt.pose.n_first_fixed_poses > 1:
            return pose_w2c, edict(R=torch.eye(3, device=opt.device).unsqueeze(0), t=torch.zeros(1, 3, 1, device=opt.device), s=1.)
        src = camera.pad_poses(camera.pose.invert(pose_w2c))
        dst = camera.pad_poses(camera.pose.invert(pose_GT_w2c))
        n = min(src.shape[0], 10)
        best = None
        for a in range(n):
            for b in range(n):
                if a == b:
                    continue
                scale = torch.norm(dst[a, :3, 3] - dst[b, :3, 3]) / torch.norm(src[a, :3, 3] - src[b, :3, 3])
                scaled = src.clone()
                scaled[:, :3, 3] = scaled[:, :3, 3] * scale
                T = dst[a] @ camera.pose_inverse_4x4(scaled[a])
                aligned_w2c = camera.pose_inverse_4x4(T[None] @ scaled)[:, :3]
                err = self.evaluate_camera_alignment(opt, aligned_w2c, pose_GT_w2c)
                score = err.t.mean().item() * (err.R.mean().item() * 180. / np.pi)
                if best is None or score < best[0]:
                    best = (score, aligned_w2c, edict(R=T[:3, :3].unsqueeze(0), type="traj_align", t=T[:3, 3].reshape(1, 3, 1), s=scale))
        return best[1], best[2]

    def _prealign(self, opt, pose, pose_GT, large_above):
        fn = self.prealign_w2c_large_camera_systems if pose.shape[0] > large_above else self.prealign_w2c_small_camera_systems
        return fn(opt, pose, pose_GT)

    @torch.no_grad()
    def evaluate_any_poses(self, opt, pose_w2c, pose_GT_w2c):
        stats = {}
        error = self.evaluate_camera_alignment(opt, pose_w2c.detach(), pose_GT_w2c)
        stats["error_R_before_align"] = error.R.mean() * 180. / np.pi
        stats["error_t_before_align"] = error.t.mean()
        pose_aligned, _ = self._prealign(opt, pose_w2c.detach(), pose_GT_w2c, large_above=10)
        error = self.evaluate_camera_alignment(opt, pose_aligned, pose_GT_w2c)
        stats["error_R"] = error.R.mean() * 180. / np.pi
        stats["error_t"] = error.t.mean()
        return stats

    def evaluate_poses(self, opt):
        return self.evaluate_any_poses(opt, *self.get_all_training_poses(opt))

    @torch.no_grad()
    def validate(self, opt):
        """Install the est->gt similarity the val / eval branch of Graph.get_pose needs (barf_inn_dtu.py:370-382)"""
        pose, pose_GT = self.get_all_training_poses(opt)
        pose_aligned, self.graph.pose_net.sim3_est_to_gt_c2w = self._prealign(opt, pose, pose_GT, large_above=9)
        return pose_aligned

    def evaluate_full(self, opt, test_views):
        """-> edict(error, res=[edict(psnr, ssim, abs_err, rms_err)]); test views carry depth_gt /
        valid_depth_gt / depth_range when the dataset has them (nerf_inn_dtu.py:205-262)"""
        self.graph.eval()
        pose, pose_GT = self.get_all_training_poses(opt)
        pose_aligned = self.validate(opt)
        error = self.evaluate_camera_alignment(opt, pose_aligned, pose_GT)
        scale = self.graph.pose_net.sim3_est_to_gt_c2w.s
        res = []
        for var in test_views:
            var = edict(var)
            if opt.optim.test_photo:
                var = self.evaluate_test_time_photometric_optim(opt, var)
            with torch.no_grad():
                var = self.graph.forward(opt, var, mode="eval")
                rgb_map = var.rgb.view(-1, opt.H, opt.W, 3).permute(0, 3, 1, 2).contiguous()
                r = edict(psnr=metrics.psnr(rgb_map, var.image).item(), ssim=metrics.ssim(rgb_map, var.image).item(),
                          abs_err=float("nan"), rms_err=float("nan"))
                if "depth_gt" in var and float(scale) != 1.:
                    r.abs_err, r.rms_err = metrics.compute_depth_metrics(var, float(scale))
                res.append(r)
        return edict(error=error, res=res)
