"""Mirror of the reference's model/barf_inn_llff.py `Graph` (273-419) and `NeRF` (421-442):
learnable poses through the NVP warp + BARF coarse-to-fine positional encoding.
`attach_warp` builds the sub-modules the reference's Model.build_networks attaches
(barf_inn_llff.py:41-75) with the same attribute names and state-dict keys.
"""
import numpy as np
import torch

from .. import camera, posealign
from ..util import edict
from . import nerf_inn_llff
from .nvp import nvp_ndr


class NeRF(nerf_inn_llff.NeRF):

    def __init__(self, opt):
        super().__init__(opt)
        # a Parameter so that the c2f state is checkpointed (reference barf_inn_llff.py:425)
        self.progress = torch.nn.Parameter(torch.tensor(0., device=opt.device))

    def set_progress(self, value, device_copy=True):
        """Engine hook replacing `progress.data.fill_(it/max_iter)` (barf_inn_llff.py:117): keeps a host copy so that the band
        weights are formed without a device->host sync per step; device_copy=False defers the write of the Parameter itself
        (a launch per network and step) until state is read (engine.INNTrainer.sync_state)."""
        if device_copy:
            self.progress.data.fill_(value)
            self._progress_seen = self.progress._version
        self.progress_host = float(value)

    def band_weights(self, opt, L):
        """reference barf_inn_llff.py:427-442: w_k = (1 - cos(pi * clamp(alpha - k, 0, 1))) / 2 with
        alpha = (progress - start) / (end - start) * L; evaluated in fp32 like the reference."""
        if opt.barf_c2f is None:
            return [1.0] * L
        start, end = opt.barf_c2f
        # one source of truth: the host copy, unless the Parameter was written behind its back (reference-style
        # `nerf.progress.data.fill_()`, a checkpoint load) -- its version counter tells
        if self.progress_host is None or getattr(self, "_progress_seen", None) != self.progress._version:
            self.progress_host, self._progress_seen = float(self.progress.data), self.progress._version
        prog = self.progress_host
        alpha = (np.float32(prog) - np.float32(start)) / np.float32(end - start) * np.float32(L)
        k = np.arange(L, dtype=np.float32)
        w = (1 - np.cos(np.clip(alpha - k, 0, 1).astype(np.float32) * np.float32(np.pi))) / 2
        return [float(x) for x in w.astype(np.float32)]


class Graph(nerf_inn_llff.Graph):

    def __init__(self, opt):
        torch.nn.Module.__init__(self)
        self.nerf = NeRF(opt)
        if opt.nerf.fine_sampling:
            self.nerf_fine = NeRF(opt)
        self.pose_eye = torch.eye(3, 4).to(opt.device)

    def attach_warp(self, opt, n_views):
        """What reference Model.build_networks adds to the graph (barf_inn_llff.py:41-75, l2fbarf
        latent): warp_latent Embedding(n,128), warp_mlp DeformNetwork, global_rigid Embedding(n,12)."""
        assert opt.warp_latent.enc_type == "l2fbarf"
        self.warp_latent = torch.nn.Embedding(n_views, opt.warp_latent.embed_dim).to(opt.device)
        self.warp_mlp = nvp_ndr.build_warp_network(opt, opt.warp_latent.embed_dim)
        pose = self.pose_eye[None].repeat(n_views, 1, 1)
        self.global_rigid = torch.nn.Embedding(n_views, 12, _weight=pose.reshape(-1, 12).clone()).to(opt.device)
        return self

    def get_pose_init(self, opt, var, mode=None, ind=None, iter=None):
        """reference barf_inn_llff.py:282-302 (non-blender datasets: identity)"""
        if mode == "train":
            return self.pose_eye[None].repeat(len(var.idx), 1, 1)

    def get_pose(self, opt, var, mode=None, ind=None, iter=None):
        """reference barf_inn_llff.py:305-364 (train) and :385-399 (val / eval / test-optim)."""
        if mode == "train":
            center_cam = var.center_cam if "center_cam" in var else None
            if center_cam is None:
                center_cam, grid_cam = camera.get_unwarped_center_and_ray(opt, intr=var.intr, ray_idx=var.ray_idx)
            else:
                grid_cam = var.grid_cam
            center_cam, grid_cam = center_cam.detach(), grid_cam.detach()
            # the whole latent table is used, not var.idx rows of it (reference :334); under ray sharding the rows of this rank's views
            win = var.get("view_window")
            latent = self.warp_latent.weight if win is None else self.warp_latent.weight[win.views]
            self.warp_mlp.code_rows = None if win is None else (win.v0, win.v1)
            alpha_ratio = nvp_ndr.embedding_anneal_ratio(opt, iter)
            ray, center_3D, grid_3D = nvp_ndr.warp_grid_and_center(self.warp_mlp, latent, grid_cam, center_cam, alpha_ratio)
            return ray, center_3D, grid_3D, alpha_ratio
        if mode in ["val", "eval", "test-optim"]:
            # a held-out view has no learnt pose: its ground-truth pose is brought into the learnt frame with the similarity the
            # validation fitted to the training cameras (reference :385-396), plus, at test time, the photometric refinement
            pose = posealign.transfer_poses(self.sim3, var.pose, to_gt=False)
            if opt.optim.test_photo and mode != "val":
                pose = camera.pose.compose([var.pose_refine_test, pose])
            return pose
        return var.pose


class Model:
    """The reference's engine interface for this model (`train.py:21-32` drives it by name: Model(opt);
    load_dataset; build_networks; setup_optimizer; restore_checkpoint; setup_visualizer; train), reduced to what
    touches the render path: data to device, graph + warp modules, fused Adam / schedules (engine.INNTrainer),
    checkpoints in the reference's wire format, pose / view-synthesis evaluation (evaluation.LLFFEvaluator).
    Logging back-ends (tensorboard, visdom) are outside the path: `log_scalars` prints."""

    def __init__(self, opt):
        import os
        self.opt = opt
        os.makedirs(opt.output_path, exist_ok=True)
        self.it = self.iter_start = 0

    # ---- reference model/base.py:24-33 / nerf_inn_llff.py:22-32
    def load_dataset(self, opt, eval_split="val"):
        """reference model/base.py:24-33 + nerf_inn_llff.py:22-32: both splits pre-loaded, tensors resident on the device"""
        from .. import data
        self.train_data, self.test_data = data.open_splits(opt, eval_split=eval_split)

    # ---- barf_inn_llff.py:41-75 (+ setup_optimizer :84-104: the trainer owns the flat Adam state and the schedules)
    def build_networks(self, opt):
        from .. import engine, parallel
        rank, world, _ = parallel.init_from_env()
        self.trainer = engine.INNTrainer(opt, len(self.train_data), rank=rank, world=world, seed=opt.seed or 0)
        self.graph = self.trainer.graph

    def setup_optimizer(self, opt):
        assert hasattr(self, "trainer"), "build_networks first"

    def restore_checkpoint(self, opt):
        from .. import checkpoint
        ep = it = None
        if opt.resume:
            ep, it = checkpoint.restore_checkpoint(opt, self.trainer, resume=opt.resume)
        elif opt.load is not None:
            ep, it = checkpoint.restore_checkpoint(opt, self.trainer, load_name=opt.load)
        self.epoch_start, self.iter_start = ep or 0, it or 0

    def setup_visualizer(self, opt):
        pass

    # ---- nerf_inn_llff.py:49-100, barf_inn_llff.py:106-120
    def train(self, opt):
        self.graph.train()
        var = self.train_data.all
        self.it = self.iter_start
        if self.iter_start == 0:
            self.validate(opt, 0)
        while self.it < opt.max_iter:
            loss = self.train_iteration(opt, var, None)
            if self.it % opt.freq.scalar == 0:
                self.log_scalars(opt, var, loss, step=self.it, split="train")
            if self.it % opt.freq.val == 0:
                self.validate(opt, self.it)
            if self.it % opt.freq.ckpt == 0:
                self.save_checkpoint(opt, ep=None, it=self.it)
        return self

    def train_iteration(self, opt, var, loader=None):
        loss = self.trainer.train_iteration(edict(var))
        self.it = self.trainer.it
        return loss

    def summarize_loss(self, opt, var, loss):
        return self.trainer.summarize_loss(loss)

    def log_scalars(self, opt, var, loss, metric=None, step=0, split="train"):
        msg = " ".join("{}={:.5f}".format(k, float(torch.as_tensor(v).detach())) for k, v in loss.items())
        print("[{} it {}] {}".format(split, step, msg), flush=True)

    def _evaluator(self, opt):
        from .. import evaluation
        return evaluation.LLFFEvaluator(opt, self.graph, self.train_data.get_all_camera_poses(opt).to(opt.device))

    @torch.no_grad()
    def validate(self, opt, ep=None):
        """pose errors after the Procrustes pre-alignment + PSNR of the held-out views rendered from their
        aligned ground-truth poses (nerf_inn_llff.py:130-160, barf_inn_llff.py:122-145)"""
        self.trainer.sync_state()                  # (under ray sharding: collect the per-view poses the ranks hold)
        self.graph.eval()
        ev = self._evaluator(opt)
        pose, pose_GT = ev.get_all_training_poses(opt)
        pose_aligned, self.graph.sim3 = ev.prealign_cameras(opt, pose, pose_GT)
        error = ev.evaluate_camera_alignment(opt, pose_aligned, pose_GT)
        psnr = []
        allv = self.test_data.all
        for i in range(len(self.test_data)):
            var = edict({k: v[i:i + 1] for k, v in allv.items()})
            var = self.graph.forward(opt, var, mode="val")
            rgb_map = var.rgb.view(-1, opt.H, opt.W, 3).permute(0, 3, 1, 2)
            psnr.append(-10 * self.graph.MSE_loss(rgb_map, var.image).log10().item())
        self.graph.train()
        out = edict(error_R=float(error.R.mean()), error_t=float(error.t.mean()), psnr=sum(psnr) / max(len(psnr), 1))
        print("[val it {}] rot {:.4f} rad  trans {:.5f}  PSNR {:.2f}".format(ep, out.error_R, out.error_t, out.psnr), flush=True)
        return out

    def evaluate_full(self, opt):
        self.trainer.sync_state()                  # every rank (a collective under ray sharding)
        allv = self.test_data.all
        views = [edict({k: v[i:i + 1] for k, v in allv.items()}) for i in range(len(self.test_data))]
        out = self._evaluator(opt).evaluate_full(opt, views)
        with open("{}/quant_pose.txt".format(opt.output_path), "w") as f:
            for i, (err_R, err_t) in enumerate(zip(out.error.R, out.error.t)):
                f.write("{} {} {}\n".format(i, err_R.item(), err_t.item()))
        with open("{}/quant.txt".format(opt.output_path), "w") as f:
            for i, r in enumerate(out.res):
                f.write("{} {} {}\n".format(i, r.psnr, r.ssim))
        return out

    def save_checkpoint(self, opt, ep=0, it=0, latest=False):
        from .. import checkpoint
        # every rank: the per-view pose table is collected with an all-reduce (engine.INNTrainer.sync_state); only then the rank gate
        self.trainer.sync_state()
        if self.trainer.rank == 0:
            checkpoint.save_checkpoint(opt, self.trainer, ep=ep, it=it, latest=latest)
