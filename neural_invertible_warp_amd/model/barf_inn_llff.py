"""Mirror of the reference's model/barf_inn_llff.py `Graph` (273-419) and `NeRF` (421-442):
learnable poses through the NVP warp + BARF coarse-to-fine positional encoding.
`attach_warp` builds the sub-modules the reference's Model.build_networks attaches
(barf_inn_llff.py:41-75) with the same attribute names and state-dict keys.
"""
import numpy as np
import torch

from .. import camera
from . import nerf_inn_llff
from .nvp import nvp_ndr


class NeRF(nerf_inn_llff.NeRF):

    def __init__(self, opt):
        super().__init__(opt)
        # a Parameter so that the c2f state is checkpointed (reference barf_inn_llff.py:425)
        self.progress = torch.nn.Parameter(torch.tensor(0., device=opt.device))

    def set_progress(self, value):
        """Engine hook replacing `progress.data.fill_(it/max_iter)` (barf_inn_llff.py:117): also keeps
        a host copy so that the band weights are formed without a device->host sync per step."""
        self.progress.data.fill_(value)
        self.progress_host = float(value)

    def band_weights(self, opt, L):
        """reference barf_inn_llff.py:427-442: w_k = (1 - cos(pi * clamp(alpha - k, 0, 1))) / 2 with
        alpha = (progress - start) / (end - start) * L; evaluated in fp32 like the reference."""
        if opt.barf_c2f is None:
            return [1.0] * L
        start, end = opt.barf_c2f
        prog = self.progress_host if self.progress_host is not None else float(self.progress.data)
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
        self.warp_mlp = nvp_ndr.DeformNetwork(d_feature=opt.warp_latent.embed_dim, d_in=3, d_out_1=1, d_out_2=3, n_blocks=3,
                                              d_hidden=opt.inn.real_nvp.d_hidden, n_layers=1, skip_in=[],
                                              multires=opt.inn.real_nvp.multires, weight_norm=True, actfn=opt.inn.actfn).to(opt.device)
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
            feat = self.warp_latent.weight                                   # whole table, not indexed by var.idx (:334)
            if opt.inn.real_nvp.c2f == True:  # noqa: E712  (reference :351)
                alpha_ratio = max(min(iter / opt.inn.real_nvp.max_pe_iter, 1), 0)
            else:
                alpha_ratio = 1
            n = grid_cam.shape[1]
            camera_coords_3D = torch.cat([grid_cam, center_cam], dim=1)
            warped = self.warp_mlp.forward(feat, camera_coords_3D.unsqueeze(2), alpha_ratio=alpha_ratio)
            grid_3D, center_3D = warped[:, :n], warped[:, n:]
            ray = grid_3D - center_3D
            return ray.squeeze(2), center_3D.squeeze(2), grid_3D.squeeze(2), alpha_ratio
        if mode in ["val", "eval", "test-optim"]:
            sim3 = self.sim3
            center = torch.zeros(1, 1, 3, device=opt.device)
            center = camera.cam2world(center, var.pose)[:, 0]
            center_aligned = (center - sim3.t0) / sim3.s0 @ sim3.R * sim3.s1 + sim3.t1
            R_aligned = var.pose[..., :3] @ sim3.R
            t_aligned = (-R_aligned @ center_aligned[..., None])[..., 0]
            pose = camera.pose(R=R_aligned, t=t_aligned)
            if opt.optim.test_photo and mode != "val":
                pose = camera.pose.compose([var.pose_refine_test, pose])
            return pose
        return var.pose
