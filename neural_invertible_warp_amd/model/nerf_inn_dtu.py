"""Mirror of the reference's model/nerf_inn_dtu.py `Graph` (363-567) and `NeRF` (569-680): the DTU
copy of the INN graph.  Differences from the LLFF copy: the depth range of metric sampling comes
from the data (`var.depth_range[0]`, :373-376) and is threaded through render / render_local /
sample_depth; the alignment loss uses the detached Kabsch pose kept by the pose network
(:410-414); the inverse-CDF bins still use the yaml range (:549), as in the reference."""
import torch

from .. import ops
from . import nerf, nerf_inn_llff


class NeRF(nerf.NeRF):
    """reference nerf_inn_dtu.py:569-680 (duplicate of model/nerf.py's NeRF)"""


class Graph(nerf_inn_llff.Graph):

    def __init__(self, opt):
        torch.nn.Module.__init__(self)
        self.nerf = NeRF(opt)
        if opt.nerf.fine_sampling:
            self.nerf_fine = NeRF(opt)

    def forward(self, opt, var, mode=None, iter=None):
        """reference nerf_inn_dtu.py:371-396"""
        batch_size = len(var.idx)
        depth_range = opt.nerf.depth.range if opt.nerf.depth.param == "inverse" else self._host_depth_range(var.depth_range)
        if opt.nerf.rand_rays and mode == "train":
            var.ray_idx = self.draw_ray_idx(opt, batch_size, draw=iter)
            var.view_window = self._last_window = self.view_window(opt, batch_size, len(var.ray_idx))      # ray sharding (..parallel)
            ray, center, grid_3d = self.get_pose(opt, var, mode=mode, iter=iter)
            ret = self.render_share(opt, var, ray, center, mode=mode, depth_range=depth_range)
            ret.update(grid_local=grid_3d, center_local=center, grid_init=self.pose_net.grid_init, center_init=self.pose_net.center_init)
        elif opt.nerf.rand_rays and mode == "test-optim":
            # test-time pose refinement: random rays of the back-aligned (and refined) test pose.  The reference's DTU forward
            # unpacks that pose as (ray, center, grid) in this mode (nerf_inn_dtu.py:385) and cannot run; rendered properly here.
            var.ray_idx = self.draw_ray_idx(opt, batch_size)
            ret = self.render(opt, self.get_pose(opt, var, mode=mode), intr=var.intr, ray_idx=var.ray_idx, mode=mode, depth_range=depth_range)
        else:
            pose_w2c = self.get_pose(opt, var, mode=mode)
            ret = self.render_by_slices(opt, pose_w2c, intr=var.intr, mode=mode, depth_range=depth_range) if opt.nerf.rand_rays else \
                self.render(opt, pose_w2c, intr=var.intr, mode=mode, depth_range=depth_range)
        var.update(ret)
        return var

    def _host_depth_range(self, depth_range):
        """var.depth_range[0] as two Python floats (the kernels take the range by value).  The device->host read is a
        synchronisation point, so it is done once per tensor OBJECT and content version (the range is a per-scene constant);
        the tensor is held by a weak reference, so an unrelated tensor that the allocator later places at the same address is
        never mistaken for it."""
        import weakref
        cached = getattr(self, "_dr_cache", None)
        if cached is None or cached[0]() is not depth_range or cached[1] != depth_range._version:
            self._dr_cache = cached = (weakref.ref(depth_range), depth_range._version, [float(x) for x in depth_range[0]])
        return cached[2]

    def compute_loss(self, opt, var, mode=None):
        """reference nerf_inn_dtu.py:398-415: as the LLFF graph, except that the registered pose is the DETACHED one the pose network
        keeps (refreshed by every get_warped_rays_in_world) and the un-warped points are the initial-pose world points."""
        loss = nerf.Graph.compute_loss(self, opt, var, mode=mode)
        if mode != "train" or opt.loss_weight.global_alignment is None:
            return loss
        from .nvp import nvp_ndr
        stack = nvp_ndr.stacked_points(var.grid_local, var.center_local)
        if stack is not None:
            warped, initial = stack
        else:
            warped = torch.cat([var.grid_local, var.center_local], dim=1)
            initial = torch.cat([var.grid_init, var.center_init], dim=1)
        win = var.get("view_window")
        backend = nerf_inn_llff.ALIGN_BACKEND or ops
        poses = self.pose_net.get_w2c_poses()
        if win is None:
            loss.global_alignment = backend.alignment_residual(warped, initial, poses, n_norm=warped.numel())
        else:
            # this rank's window of views: the alignment terms of the views it OWNS, normalised by the global point count (..parallel)
            own = win.owned_in_window
            if own.stop > own.start:
                loss.global_alignment = backend.alignment_residual(warped[own], initial[own], poses[win.own0:win.own1], n_norm=3 * win.B * warped.shape[1])
            else:
                loss.global_alignment = warped.sum() * 0.0
        return loss

    def render(self, opt, pose, intr=None, ray_idx=None, mode=None, depth_range=None):
        """reference nerf_inn_dtu.py:472-509"""
        return self._render_pixels(opt, pose, intr, mode, ray_idx=ray_idx, depth_range=depth_range)

    def render_by_slices(self, opt, pose, intr=None, mode=None, depth_range=None):
        """reference nerf_inn_dtu.py:511-522"""
        if not torch.is_grad_enabled():
            return self._render_image(opt, pose, intr, depth_range=depth_range)
        return self._sweep_image(opt, lambda first, count: self._render_pixels(opt, pose, intr, mode, pixel_range=(first, count),
                                                                               depth_range=depth_range))

    def sample_depth(self, opt, batch_size, num_rays=None, depth_range=None):
        """reference nerf_inn_dtu.py:524-546 (explicit depth_range)"""
        return nerf.Graph.sample_depth(self, opt, batch_size, num_rays=num_rays,
                                       depth_range=opt.nerf.depth.range if depth_range is None else depth_range)
