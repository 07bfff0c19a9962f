"""Mirror of the reference's model/barf_inn_dtu.py `Graph` (524-599) and `NeRF` (601-622): DTU,
learnable poses through `INNPoseParams` (the warp lives on the pose network that the engine
passes into the graph, barf_inn_dtu.py:325-336) + BARF coarse-to-fine encoding."""
import torch

from .. import camera
from ..util import edict
from ..align_trajectories import backtrack_from_aligning_the_trajectory
from . import barf_inn_llff, nerf_inn_dtu


class NeRF(barf_inn_llff.NeRF):
    """reference barf_inn_dtu.py:601-622 -- identical c2f override to the LLFF model."""


class Graph(nerf_inn_dtu.Graph):

    def __init__(self, opt, pose_net):
        torch.nn.Module.__init__(self)
        self.pose_net = pose_net
        self.nerf = NeRF(opt)
        if opt.nerf.fine_sampling:
            self.nerf_fine = NeRF(opt)
        self.pose_eye = torch.eye(3, 4).to(opt.device)

    def get_pose(self, opt, var, mode=None, iter=None):
        return self.get_w2c_pose(opt, var, mode, iter)

    def get_w2c_pose(self, opt, var, mode=None, iter=None):
        """reference barf_inn_dtu.py:538-567.  val / eval: the ground-truth test pose is brought into the
        frame of the optimised poses with the est->gt similarity stored on the pose network by the
        evaluator (evaluation.DTUEvaluator.validate, reference :370-382)."""
        if mode == "train":
            assert iter is not None, "ERROR: Iteration is needed for the c2f embedding in INN"
            return self.pose_net.get_warped_rays_in_world(var, mode, iter)
        if mode in ["val", "eval", "test-optim", "test"]:
            assert hasattr(self.pose_net, "sim3_est_to_gt_c2w")
            sim = self.pose_net.sim3_est_to_gt_c2w
            if sim.type != "traj_align":
                raise ValueError(sim.type)
            pose = backtrack_from_aligning_the_trajectory(var.pose, sim)
            if opt.optim.test_photo and mode != "val":
                pose = camera.pose.compose([var.pose_refine_test, pose])
            return pose
        raise ValueError(mode)

    def get_c2w_pose(self, opt, var, mode=None):
        return camera.pose.invert(self.get_w2c_pose(opt, var, mode))


from . import nerf as _nerf  # noqa: E402  (engine interface of the vanilla model: data, nerf optimizer, loop, checkpoints)


class Model(_nerf.Model):
    """The reference's engine interface for DTU with learnable poses (model/barf_inn_dtu.py:302-465): initial poses
    (`pose.init`: noisy_gt / given / identity, :31-66), INNPoseParams, Adam + ExponentialLR for the NeRF and a second
    pair for the pose network (:341-355), c2f progress after every step (:367-369), validation = pose errors after the
    pairwise / trajectory alignment + held-out PSNR from the back-aligned test poses (:370-382)."""

    def set_initial_poses(self, opt):
        gt = self.train_data.get_all_camera_poses(opt).to(opt.device)
        n = len(self.train_data)
        if opt.pose.init == "noisy_gt":
            noise = camera.lie.se3_to_SE3(torch.randn(n, 6, device=opt.device) * opt.pose.noise)
            return camera.pose.compose([noise, gt])[:, :3]
        if opt.pose.init == "given":
            return self.train_data.all.pose.to(opt.device)[:, :3]
        if opt.pose.init == "identity":
            return torch.eye(3, 4, device=opt.device).repeat(n, 1, 1)
        raise ValueError("pose.init={} (COLMAP / PDC-Net initialisation is outside the render path)".format(opt.pose.init))

    def build_networks(self, opt):
        """Adam on a GPU (the shipped options/barf_inn_dtu.yaml): engine.INNTrainer (family "dtu": INNPoseParams + Graph as below, one
        niw_train_step call per iteration, one fused Adam launch over the field network(s), the pose network and the latent table, both
        ExponentialLR schedules, c2f progress; round 6 -- rounds 1-5 ran the mirror under two torch.optim.Adam here).  Any other
        optimizer: the reference's own structure, on the mirror."""
        from .pose_models.inn import INNPoseParams
        torch.manual_seed(opt.seed or 0)
        init = self.set_initial_poses(opt)
        self.trainer = None
        if opt.optim.get("algo", "Adam") == "Adam" and torch.device(opt.device).type == "cuda":
            from .. import engine, parallel
            rank, world, _ = parallel.init_from_env()
            self.trainer = engine.INNTrainer(opt, len(self.train_data), rank=rank, world=world, seed=opt.seed or 0, initial_poses_w2c=init)
            self.graph, self.pose_net = self.trainer.graph, self.trainer.pose_net
            return
        self.pose_net = INNPoseParams(opt, num_poses=len(self.train_data), initial_poses_w2c=init, device=opt.device)
        self.graph = Graph(opt, self.pose_net).to(opt.device)

    def setup_optimizer(self, opt):
        if self.trainer is not None:
            return                                    # the trainer owns both Adam states and both schedules
        super().setup_optimizer(opt)
        optimizer = getattr(torch.optim, opt.optim.algo)
        self.optim_pose = optimizer([dict(params=self.pose_net.pose_embedding.parameters(), lr=opt.optim.lr_pose)])
        self.optim_pose.add_param_group(dict(params=self.pose_net.pose_latent.parameters(), lr=opt.optim.lr_pose))
        gamma = (opt.optim.lr_pose_end / opt.optim.lr_pose) ** (1. / opt.max_iter) if opt.optim.get("lr_pose_end") else 1.0
        self.sched_pose = torch.optim.lr_scheduler.ExponentialLR(self.optim_pose, gamma=gamma)

    def train_iteration(self, opt, var, loader=None):
        if self.trainer is not None:
            loss = self.trainer.train_iteration(edict(var))
            self.it = self.trainer.it
            return loss
        self.optim.zero_grad(set_to_none=True)
        self.optim_pose.zero_grad(set_to_none=True)
        var = self.graph.forward(opt, edict(var), mode="train", iter=self.it)
        loss = self.summarize_loss(opt, var, self.graph.compute_loss(opt, var, mode="train"))
        loss.all.backward()
        self.optim.step()
        self.optim_pose.step()
        self.sched_pose.step()
        self.it += 1
        for net in (self.graph.nerf, getattr(self.graph, "nerf_fine", None)):
            if net is not None:
                net.set_progress(self.it / opt.max_iter)
        return loss

    def _evaluator(self, opt):
        from .. import evaluation
        return evaluation.DTUEvaluator(opt, self.graph, self.train_data.get_all_camera_poses(opt).to(opt.device))

    @torch.no_grad()
    def validate(self, opt, ep=None):
        if self.trainer is not None:
            self.trainer.sync_state()                 # every rank: c2f progress Parameter, per-view poses of the owning ranks
        ev = self._evaluator(opt)
        stats = ev.evaluate_poses(opt)
        ev.validate(opt)                               # installs the est->gt similarity the val / eval pose branch needs
        out = super().validate(opt, ep)
        out.update(error_R=float(stats["error_R"]), error_t=float(stats["error_t"]))
        print("[val it {}] rot {:.3f} deg  trans {:.5f}".format(ep, out.error_R, out.error_t), flush=True)
        return out

    def evaluate_full(self, opt):
        if self.trainer is not None:
            self.trainer.sync_state()
        allv = self.test_data.all
        return self._evaluator(opt).evaluate_full(opt, [edict({k: v[i:i + 1] for k, v in allv.items()}) for i in range(len(self.test_data))])

    def save_checkpoint(self, opt, ep=0, it=0, latest=False):
        import os
        import shutil
        if self.trainer is not None:
            from .. import checkpoint
            self.trainer.sync_state()                 # every rank (a collective under ray sharding), then the rank gate
            if self.trainer.rank == 0:
                checkpoint.save_checkpoint(opt, self.trainer, ep=ep, it=it, latest=latest)
            return
        os.makedirs("{0}/model".format(opt.output_path), exist_ok=True)
        ck = dict(epoch=ep, iter=it, graph=self.graph.state_dict(), optim=self.optim.state_dict(), sched=self.sched.state_dict(),
                  optim_pose=self.optim_pose.state_dict(), sched_pose=self.sched_pose.state_dict())
        torch.save(ck, "{0}/model.ckpt".format(opt.output_path))
        if not latest:
            shutil.copy("{0}/model.ckpt".format(opt.output_path), "{0}/model/{1}.ckpt".format(opt.output_path, ep or it))

    def restore_checkpoint(self, opt):
        super().restore_checkpoint(opt)               # (with the engine's trainer: both optimizers, both schedules, the counters)
        if opt.resume and self.trainer is None:
            name = "{0}/model.ckpt".format(opt.output_path) if opt.resume is True else "{0}/model/{1}.ckpt".format(opt.output_path, opt.resume)
            ck = torch.load(name, map_location=opt.device, weights_only=False)
            self.optim_pose.load_state_dict(ck["optim_pose"])
            self.sched_pose.load_state_dict(ck["sched_pose"])
            for net in (self.graph.nerf, getattr(self.graph, "nerf_fine", None)):
                if net is not None:
                    net.set_progress(float(net.progress.data))
