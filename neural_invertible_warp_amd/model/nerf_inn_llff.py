"""Mirror of the reference's model/nerf_inn_llff.py `Graph` (485-703) and `NeRF` (705-818):
the INN variant whose train-mode forward renders from warped rays (`render_local`) and whose
loss adds the Kabsch global-alignment term.
"""
import torch

from .. import camera, ops
from ..util import edict
from . import nerf


# Test seam: the CPU-only multi-process tests of the sharded alignment term plug a torch restatement of the two fused
# operations in here (an object with `rigid_registration` and `alignment_residual`).  None = libniw_hip.so (..ops).
ALIGN_BACKEND = None


def rigid_points_registration(x, y, sharded=False):
    """[R|t] minimising sum ||R x + t - y||^2 over points [B,N,3] (Kabsch with reflection fix) -> (R [B,3,3], t [B,3]).
    Stands in for roma.rigid_points_registration (roma==1.4.1, reference requirements.txt:1, call sites
    model/nerf_inn_llff.py:569, model/pose_models/inn.py:100); that package is not vendored in the reference, so this restates
    the published algorithm -- parity unpinned.  Two launches (niw_align_moments: fp64 per-view moments; niw_align_solve:
    Jacobi-SVD solve).  sharded=True (ray-shard data parallelism): every rank holds a slice of the points; the moments n, sum x,
    sum y, sum y x^T (16 doubles per view) are summed over ranks before the solve, so all ranks obtain the registration of the
    GLOBAL point set.  Gradient-free: see `Graph.compute_loss`."""
    from .. import parallel
    poses = (ALIGN_BACKEND or ops).rigid_registration(x, y, reduce_moments=parallel.all_reduce_sum_ if sharded else None)
    return poses[..., :3], poses[..., 3]


class NeRF(nerf.NeRF):
    """reference nerf_inn_llff.py:705-818 (a verbatim duplicate of model/nerf.py's NeRF)."""


class Graph(nerf.Graph):

    def __init__(self, opt):
        torch.nn.Module.__init__(self)
        self.nerf = NeRF(opt)
        if opt.nerf.fine_sampling:
            self.nerf_fine = NeRF(opt)

    def forward(self, opt, var, mode=None, iter=None):
        """reference nerf_inn_llff.py:493-546 (blender depth-range rescaling :495-502 and the dead
        render_train branch :531-538 are outside the supported configs)."""
        batch_size = len(var.idx)
        if opt.nerf.rand_rays and mode in ["train", "test-optim"]:
            var.ray_idx = self.draw_ray_idx(opt, batch_size, draw=iter)
            if mode == "train":
                pose_init = self.get_pose_init(opt, var, mode=mode, iter=iter)
                # under ray sharding (..parallel): this rank's window of whole views; every tensor from here to the renderer is the window's
                win = self.view_window(opt, batch_size, len(var.ray_idx))
                var.view_window = self._last_window = win
                views = slice(None) if win is None else win.views
                # camera-frame grid / centre kept for the alignment loss (:519); get_pose reuses them
                center_cam, grid_cam = camera.get_unwarped_center_and_ray(opt, intr=var.intr[views], ray_idx=var.ray_idx,
                                                                          pose_init=None if opt.data.dataset != "blender" else pose_init[views])
                var.center_cam, var.grid_cam = center_cam, grid_cam
                ray, center, grid_3D, alpha_ratio = self.get_pose(opt, var, mode=mode, iter=iter)
                ret = self.render_share(opt, var, ray, center, mode=mode)
                ret.update(grid_3D=grid_3D, center=center, grid_cam=grid_cam, center_cam=center_cam, inn_posenc_alpha=alpha_ratio)
            else:
                pose = self.get_pose(opt, var, mode=mode)
                ret = self.render(opt, pose, intr=var.intr, ray_idx=var.ray_idx, mode=mode)
        else:
            pose = self.get_pose(opt, var, mode=mode)
            ret = self.render_by_slices(opt, pose, intr=var.intr, mode=mode) if opt.nerf.rand_rays else \
                self.render(opt, pose, intr=var.intr, mode=mode)
        var.update(ret)
        return var

    def view_window(self, opt, n_views, n_rays_per_view):
        """..parallel.ViewWindow of this rank under ray sharding, else None"""
        shard = getattr(opt, "ray_shard", None)
        if shard is None:
            return None
        from .. import parallel
        return parallel.ViewWindow(n_views, n_rays_per_view, *shard)

    def render_share(self, opt, var, ray, center, mode=None, depth_range=None):
        """render_local on what this rank renders of the warped rays: all of them [B,R,3], or -- under ray sharding (..parallel) -- its
        contiguous share of the flattened view-major ray list as ONE [1, n, 3] batch, cut out of the window of whole views it has
        warped (`var.view_window`); `var.ray_share` = (lo, hi) tells the loss which rays of the global batch those are.  Autograd
        routes the share's gradient back into the window's warp."""
        win = var.get("view_window")
        if win is None:
            var.ray_share = None
            return self.render_local(opt, ray, center, intr=var.intr, mode=mode, depth_range=depth_range)
        if opt.camera.ndc:                       # per-view intrinsics: reparametrise while the rays still have their [views, R] shape
            center, ray = camera.convert_NDC(opt, center, ray, intr=var.intr[win.views])
        n = ray.shape[0] * ray.shape[1]
        a, b = win.local
        var.ray_share = (win.lo, win.hi)
        return self._render_rays(opt, center.reshape(1, n, 3)[:, a:b], ray.reshape(1, n, 3)[:, a:b], mode=mode, depth_range=depth_range)

    def draw_ray_idx(self, opt, batch_size, draw=None):
        """The pixel subset of a training step (reference :510): `nerf.rand_rays // batch_size` distinct pixels, the SAME set for
        every view.  Two samplers, equal in distribution (a uniformly random subset in random order):
          * "randperm" (default): `torch.randperm(H*W)[:n]`, the reference's own call (a device sort of H*W keys);
          * "feistel" (`opt.nerf.ray_sampler`, what the engine selects): niw_draw_ray_idx, one sort-free launch keyed by
            (opt.seed, `draw` = the training iteration -- a resumed run continues the same sequence; a running count of the
            calls when no iteration is given), also replayable from a captured HIP graph (`self.draw_dev`).
        Under ray sharding (..parallel) every rank draws the SAME n pixels (it warps all of them and renders a contiguous share of
        the B x n rays, render_share): "feistel" has that by construction; "randperm" then draws from a generator of its own, seeded
        alike on all ranks, because the ranks' default generators drift apart as soon as their stratified draws differ in size."""
        n = opt.nerf.rand_rays // batch_size
        self._ray_draws = getattr(self, "_ray_draws", 0) + 1
        number = self._ray_draws if draw is None else int(draw) + 1
        self._depth_draw = number                  # the in-kernel stratified draw of this forward uses the same number (nerf.Graph.sample_depth)
        self._depth_call_in_iter = 0               # ... and numbers its sample_depth calls from here (one stream per call)
        if opt.nerf.get("ray_sampler", "randperm") == "feistel":
            return ops.draw_ray_idx(opt.H * opt.W, n, int(getattr(opt, "seed", 0) or 0), number, opt.device, draw_dev=getattr(self, "draw_dev", None))
        if getattr(opt, "ray_shard", None) is None:
            return torch.randperm(opt.H * opt.W, device=opt.device)[:n]
        gen = getattr(self, "_ray_idx_gen", None)
        if gen is None:
            gen = self._ray_idx_gen = torch.Generator(device=opt.device)
            gen.manual_seed(1234567 + int(getattr(opt, "seed", 0) or 0))
        return torch.randperm(opt.H * opt.W, device=opt.device, generator=gen)[:n]

    def get_pose_init(self, opt, var, mode=None, ind=None, iter=None):
        return None

    def compute_loss(self, opt, var, mode=None):
        """reference nerf_inn_llff.py:548-573: photometric term(s) + the global-alignment term, which ties the per-point warp to
        ONE rigid motion per view: with [R|t] the rigid registration of the warped points (pixel grid ; camera centre) onto the
        un-warped ones, the loss is the mean squared distance between the warped points and the registered rigid image of the
        un-warped ones.  Fused (ops.rigid_registration + ops.alignment_residual: three launches).  The reference back-propagates
        through roma's SVD as well; that part of the gradient is identically zero (the loss is stationary in [R|t], which
        minimises it), so only the direct term is formed."""
        loss = super().compute_loss(opt, var, mode=mode)
        if opt.loss_weight.global_alignment is None or mode != "train":
            return loss
        backend = ALIGN_BACKEND or ops
        from .nvp import nvp_ndr
        stack = nvp_ndr.stacked_points(var.grid_3D, var.center)
        if stack is not None:                 # the warp's own [grid ; centre] input and output of this step
            warped, unwarped = stack
        else:
            unwarped = torch.cat([var.grid_cam, var.center_cam], dim=1)
            warped = torch.cat([var.grid_3D, var.center], dim=1)
        poses = backend.rigid_registration(warped, unwarped)                          # whole views: no collective under sharding either
        win = var.get("view_window")
        if win is None:
            if hasattr(self, "global_rigid"):
                self.global_rigid.weight.data = poses.reshape(-1, 12)              # what pose evaluation reads (:570); `poses` is a fresh tensor
            loss.global_alignment = backend.alignment_residual(warped, unwarped, poses, n_norm=warped.numel())
            return loss
        # under ray sharding the tensors are this rank's window of views: it refreshes their rows of global_rigid (evaluation gathers the
        # table, `gather_global_rigid`) and counts the alignment term of the views it OWNS, normalised by the global point count, so
        # that the ranks' losses and gradients sum to the unsharded term
        if hasattr(self, "global_rigid"):
            self.global_rigid.weight.data[win.views] = poses.reshape(-1, 12)
        own = win.owned_in_window
        n_global = 3 * win.B * warped.shape[1]
        if own.stop > own.start:
            loss.global_alignment = backend.alignment_residual(warped[own], unwarped[own], poses[own], n_norm=n_global)
        else:
            loss.global_alignment = warped.sum() * 0.0                              # (no view starts in this rank's share)
        return loss

    def gather_global_rigid(self, opt=None):
        """Under ray sharding every rank refreshes only the rows of `global_rigid` of the views it warps; before the table is read as a
        whole (validation, checkpoints) every view's row is taken from the rank that owns the view (one small all-reduce)."""
        from .. import parallel
        if hasattr(self, "global_rigid"):
            parallel.gather_owned_rows(self.global_rigid.weight.data, getattr(self, "_last_window", None))      # in place: the storage is stable

    def get_pose(self, opt, var, mode=None):
        return var.pose

    def render_local(self, opt, ray, center, intr=None, ray_idx=None, mode=None, depth_range=None):
        """reference nerf_inn_llff.py:581-612 (depth_range: the DTU copy's extra argument,
        nerf_inn_dtu.py:420)"""
        if ray_idx is not None:
            center, ray = center[:, ray_idx], ray[:, ray_idx]
        if opt.camera.ndc:
            center, ray = camera.convert_NDC(opt, center, ray, intr=intr)
        return self._render_rays(opt, center, ray, mode=mode, depth_range=depth_range)

    def render_by_slices_local(self, opt, ray, center, intr=None, mode=None):
        """reference nerf_inn_llff.py:614-625: a full image from given rays, in pixel ranges (views of `ray` / `center`)"""
        return self._sweep_image(opt, lambda first, count: self.render_local(opt, ray[:, first:first + count], center[:, first:first + count],
                                                                             intr=intr, mode=mode))
