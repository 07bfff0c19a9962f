"""Mirror of the reference's INNPoseParams (model/pose_models/inn.py:9-102): the DTU wrapper of
the NVP warp.  Camera-frame grid / centre points are first moved to the world by the (noisy)
initial poses (camera.py:382-384), then warped per view; the Kabsch registration of the result
is stored, detached, in `pose_global` (inn.py:96-102)."""
import torch

from ... import camera
from ..nerf_inn_llff import rigid_points_registration
from ..nvp import nvp_ndr


class INNPoseParams(torch.nn.Module):
    def __init__(self, opt, num_poses, initial_poses_w2c, device="cuda"):
        super().__init__()
        self.opt = opt
        self.num_poses = num_poses
        self.device = opt.device
        self.initial_poses_w2c = initial_poses_w2c
        self.init_poses_embed()

    def init_poses_embed(self):
        """reference inn.py:19-31"""
        o = self.opt
        self.pose_latent = torch.nn.Embedding(self.num_poses, o.inn.real_nvp.latent_dim).to(o.device)
        self.pose_embedding = nvp_ndr.DeformNetwork(d_feature=o.inn.real_nvp.latent_dim, d_in=3, d_out_1=1, d_out_2=3, n_blocks=3,
                                                    d_hidden=o.inn.real_nvp.d_hidden, n_layers=1, skip_in=[],
                                                    multires=o.inn.real_nvp.multires, weight_norm=True, actfn=o.inn.actfn).to(o.device)
        self.pose_global = torch.nn.Embedding(self.num_poses, 12).to(o.device)

    def get_w2c_poses(self):
        """reference inn.py:34-43"""
        return self.pose_global.weight.data.detach().clone().view(-1, 3, 4)

    def get_warped_rays_in_world(self, var, mode=None, iter=None):
        """reference inn.py:63-77 -> ray, center_3D, grid_3D, each [B,R,3]"""
        assert mode == "train"
        self.center_init, self.grid_init = camera.get_unwarped_center_and_ray(self.opt, intr=var.intr, ray_idx=var.ray_idx,
                                                                              pose_init=self.initial_poses_w2c)
        center_init, grid_init = self.center_init.detach(), self.grid_init.detach()
        out = self.forward_inn(center_init, grid_init, iter)
        n = len(var.ray_idx)
        grid_3D_pred = out[:, :n].squeeze(2)
        center_3D_pred = out[:, n:].squeeze(2)
        ray_pred = grid_3D_pred - center_3D_pred
        self.solve_for_global_transformation(grid_3D_pred, center_3D_pred)
        return ray_pred, center_3D_pred, grid_3D_pred

    def forward_inn(self, centers, grids, iter):
        """reference inn.py:81-93"""
        feat = self.pose_latent.weight
        o = self.opt.inn.real_nvp
        alpha_ratio = max(min(iter / o.max_pe_iter, 1), 0) if o.c2f == True else 1  # noqa: E712
        input_coords = torch.cat([grids, centers], dim=1).unsqueeze(2)
        return self.pose_embedding.forward(feat, input_coords, alpha_ratio=alpha_ratio)

    def solve_for_global_transformation(self, grid_pred, center_pred):
        """reference inn.py:96-102"""
        source = torch.cat([self.grid_init, self.center_init], dim=1)
        target = torch.cat([grid_pred, center_pred], dim=1)
        R_global, t_global = rigid_points_registration(target, source)
        svd_poses = torch.cat((R_global, t_global[..., None]), -1)
        self.pose_global.weight.data = svd_poses.detach().clone().view(-1, 12)
