"""Pose parameterisation of the DTU INN model: the interface of the reference's `INNPoseParams`
(model/pose_models/inn.py:9-102 -- constructor arguments, `pose_latent` / `pose_embedding` / `pose_global` modules and their
state-dict keys, `get_w2c_poses`, `get_warped_rays_in_world`, `forward_inn`, `solve_for_global_transformation`,
`grid_init` / `center_init`), expressed with the helpers the LLFF graph uses too (..nvp.nvp_ndr).

Camera-frame pixel grid and centre points are first placed in the world by the (noisy) initial poses inside the ray
generator (camera.py:382-384), then warped view by view; the rigid registration of the warped onto the initial points is
kept, detached, as the current world-to-camera correction `pose_global` (what pose evaluation and the alignment loss read).
"""
import torch

from ... import camera
from ..nerf_inn_llff import rigid_points_registration
from ..nvp import nvp_ndr


class INNPoseParams(torch.nn.Module):
    def __init__(self, opt, num_poses, initial_poses_w2c, device="cuda"):
        super().__init__()
        dim = opt.inn.real_nvp.latent_dim
        self.opt, self.num_poses, self.device = opt, num_poses, opt.device
        self.initial_poses_w2c = initial_poses_w2c
        self.grid_init = self.center_init = None          # refreshed by every get_warped_rays_in_world (the DTU graph's loss reads them)
        self.pose_latent = torch.nn.Embedding(num_poses, dim).to(self.device)          # per-view code of the warp
        self.pose_embedding = nvp_ndr.build_warp_network(opt, dim)                    # the warp itself
        self.pose_global = torch.nn.Embedding(num_poses, 12).to(self.device)           # registered [R|t] per view, not trained

    def _alpha(self, it):
        return nvp_ndr.embedding_anneal_ratio(self.opt, it)

    def get_w2c_poses(self):
        """[N,3,4] copy of the registered global correction"""
        return self.pose_global.weight.detach().clone().reshape(self.num_poses, 3, 4)

    def forward_inn(self, centers, grids, iter):
        """[B,R,3] x 2 -> warped [B,2R,1,3], grid points first"""
        stacked = torch.cat([grids, centers], dim=1)[:, :, None]
        return self.pose_embedding(self.pose_latent.weight, stacked, alpha_ratio=self._alpha(iter))

    def get_warped_rays_in_world(self, var, mode=None, iter=None):
        """-> ray, center_3D, grid_3D, each [B,R,3] (training only).  Under ray sharding (`var.view_window`, ...parallel.ViewWindow)
        B is this rank's window of whole views."""
        if mode != "train":
            raise AssertionError("INNPoseParams renders warped rays in training mode only")
        win = var.get("view_window")
        views = slice(None) if win is None else win.views
        self.center_init, self.grid_init = camera.get_unwarped_center_and_ray(self.opt, intr=var.intr[views], ray_idx=var.ray_idx,
                                                                              pose_init=self.initial_poses_w2c[views])
        self.pose_embedding.code_rows = None if win is None else (win.v0, win.v1)
        ray, center_3D, grid_3D = nvp_ndr.warp_grid_and_center(self.pose_embedding, self.pose_latent.weight[views], self.grid_init.detach(),
                                                               self.center_init.detach(), self._alpha(iter))
        self.solve_for_global_transformation(grid_3D, center_3D, views)
        return ray, center_3D, grid_3D

    def solve_for_global_transformation(self, grid_pred, center_pred, views=slice(None)):
        """Kabsch registration of the warped onto the initial points, kept detached in pose_global (reference :96-102).  Under ray
        sharding a rank warps WHOLE views (...parallel), so the registration of each of them is complete without a collective; the
        rows of the views it handles are refreshed (every rank that touches a view computes the same row)."""
        stack = nvp_ndr.stacked_points(grid_pred, center_pred)
        warped, initial = stack if stack is not None else (torch.cat([grid_pred, center_pred], dim=1), torch.cat([self.grid_init, self.center_init], dim=1))
        R, t = rigid_points_registration(warped, initial)
        rows = torch.cat([R, t.unsqueeze(-1)], dim=-1).reshape(-1, 12)
        if views == slice(None):
            self.pose_global.weight.data = rows.clone()
        else:
            self.pose_global.weight.data[views] = rows
