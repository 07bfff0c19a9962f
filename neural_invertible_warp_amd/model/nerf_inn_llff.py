"""Mirror of the reference's model/nerf_inn_llff.py `Graph` (485-703) and `NeRF` (705-818):
the INN variant whose train-mode forward renders from warped rays (`render_local`) and whose
loss adds the Kabsch global-alignment term.
"""
import torch

from .. import camera, ops
from ..util import edict
from . import nerf
from .nerf import _slice_rays


def _all_reduce_sum(t):
    """Differentiable SUM over ranks (identity without a process group)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        import torch.distributed.nn.functional as dist_fn
        return dist_fn.all_reduce(t.contiguous(), op=dist.ReduceOp.SUM)
    return t


# M [B,3,3] -> R [B,3,3].  None = the HIP solver; the CPU-only tests of the rank-sharded moments plug in a torch one.
ROTATION_SOLVER = None


def rigid_points_registration(x, y, sharded=False):
    """R, t minimising sum ||R x + t - y||^2 over points [B,N,3] (Kabsch with reflection fix).
    Stands in for roma.rigid_points_registration (roma==1.4.1, reference requirements.txt:1, call
    site model/nerf_inn_llff.py:569); that package is not vendored in the reference, so this
    restates the published algorithm -- parity unpinned.  Differentiable: the
    reference does not detach the result (nerf_inn_llff.py:569-572).  Per-view 3x3 SVD on [B,3,3]
    moments: host-side glue, not part of the per-sample path.
    sharded=True (ray-shard data parallelism): every rank holds a slice of the points; the per-view
    moments n, sum x, sum y, sum y x^T (16 floats per view) are summed over ranks before the SVD, so
    all ranks obtain the registration of the GLOBAL point set."""
    if sharded:
        n = torch.full((x.shape[0], 1), float(x.shape[1]), device=x.device, dtype=x.dtype)
        mom = torch.cat([n, x.sum(dim=1), y.sum(dim=1), (y.transpose(1, 2) @ x).reshape(x.shape[0], 9)], dim=1)
        mom = _all_reduce_sum(mom)
        n, sx, sy, syx = mom[:, :1], mom[:, 1:4], mom[:, 4:7], mom[:, 7:].reshape(-1, 3, 3)
        xm, ym = (sx / n)[:, None], (sy / n)[:, None]
        M = syx - n[:, :, None] * ym.transpose(1, 2) @ xm
    else:
        xm, ym = x.mean(dim=1, keepdim=True), y.mean(dim=1, keepdim=True)
        M = (y - ym).transpose(1, 2) @ (x - xm)
    R = (ROTATION_SOLVER or ops.kabsch_rotation)(M)       # niw_kabsch_rotation_fwd / _bwd (torch.linalg.svd blocks the host ~2 ms per call)
    t = ym[:, 0] - (R @ xm.transpose(1, 2))[..., 0]
    return R, t


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
            var.ray_idx = self.draw_ray_idx(opt, batch_size)
            if mode == "train":
                pose_init = self.get_pose_init(opt, var, mode=mode, iter=iter)
                # camera-frame grid / centre kept for the alignment loss (:519); get_pose reuses them
                center_cam, grid_cam = camera.get_unwarped_center_and_ray(opt, intr=var.intr, ray_idx=var.ray_idx,
                                                                          pose_init=None if opt.data.dataset != "blender" else pose_init)
                var.center_cam, var.grid_cam = center_cam, grid_cam
                ray, center, grid_3D, alpha_ratio = self.get_pose(opt, var, mode=mode, iter=iter)
                ret = self.render_local(opt, ray, center, intr=var.intr, mode=mode)
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

    def draw_ray_idx(self, opt, batch_size):
        """reference :510 -- one pixel set shared by every view.  Under ray sharding (..parallel.shard_ray_idx) every
        rank draws the same permutation and keeps its slice; the permutation then comes from a generator of its own,
        seeded alike on all ranks, because the ranks' default generators drift apart as soon as their stratified
        draws differ in size (rays per rank differ by one when the ray count is not a multiple of the world size)."""
        n = opt.nerf.rand_rays // batch_size
        shard = getattr(opt, "ray_shard", None)
        if shard is None:
            return torch.randperm(opt.H * opt.W, device=opt.device)[:n]
        gen = getattr(self, "_ray_idx_gen", None)
        if gen is None:
            gen = self._ray_idx_gen = torch.Generator(device=opt.device)
            gen.manual_seed(1234567 + int(getattr(opt, "seed", 0) or 0))
        rank, world = shard
        return torch.randperm(opt.H * opt.W, device=opt.device, generator=gen)[:n][rank::world]

    def get_pose_init(self, opt, var, mode=None, ind=None, iter=None):
        return None

    def compute_loss(self, opt, var, mode=None):
        """reference nerf_inn_llff.py:548-573"""
        loss = super().compute_loss(opt, var, mode=mode)
        if opt.loss_weight.global_alignment is not None and mode == "train":
            source = torch.cat([var.grid_cam, var.center_cam], dim=1)
            target = torch.cat([var.grid_3D, var.center], dim=1)
            shard = getattr(opt, "ray_shard", None)
            R_global, t_global = rigid_points_registration(target, source, sharded=shard is not None)
            svd_poses = torch.cat((R_global, t_global[..., None]), -1)
            if hasattr(self, "global_rigid"):
                self.global_rigid.weight.data = svd_poses.detach().clone().view(-1, 12)
            if shard is None:
                loss.global_alignment = self.MSE_loss(target, camera.cam2world(source, svd_poses))
            else:
                # this rank's share of the global mean (the gradient all-reduce sums the shares)
                n_global = 3 * target.shape[0] * 2 * (opt.nerf.rand_rays // target.shape[0])   # grid + centre points of the global draw
                loss.global_alignment = ((target - camera.cam2world(source, svd_poses)) ** 2).sum() / n_global
        return loss

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
        """reference nerf_inn_llff.py:614-625"""
        ret_all = edict(rgb=[], depth=[], opacity=[])
        if opt.nerf.fine_sampling:
            ret_all.update(rgb_fine=[], depth_fine=[], opacity_fine=[])
        with self._hold_weights():
            step = _slice_rays(opt)
            for c in range(0, opt.H * opt.W, step):
                ray_idx = torch.arange(c, min(c + step, opt.H * opt.W), device=opt.device)
                ret = self.render_local(opt, ray, center, intr=intr, ray_idx=ray_idx, mode=mode)
                for k in ret: ret_all[k].append(ret[k])
        for k in ret_all: ret_all[k] = torch.cat(ret_all[k], dim=1)
        return ret_all
