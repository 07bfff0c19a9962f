"""Thin training step around the Graph, honouring the reference engine's call sequence for one
iteration (model/nerf_inn_llff.py:80-100 + model/barf_inn_llff.py:106-120 + model/base.py:130-142):
zero_grad -> graph.forward(mode="train", iter=it) -> compute_loss -> summarize_loss (sum of
10^w * loss) -> backward -> Adam(nerf) + Adam(warp_mlp, warp_latent) -> ExponentialLR ->
progress = it / max_iter.  The Adam updates run as fused launches over flat buffers
(niw_adam_step); under ray sharding the gradients of all groups are summed over ranks in one
flat all-reduce first (..parallel.GradBucket).
"""
import torch

from . import ops, parallel
from .model import barf_inn_llff
from .util import edict


class INNTrainer:
    def __init__(self, opt, n_views, rank=0, world=1, warp_perturb=0.0, seed=0):
        self.opt, self.rank, self.world = opt, rank, world
        torch.manual_seed(seed)                       # identical initial weights on every rank
        self.graph = barf_inn_llff.Graph(opt).attach_warp(opt, n_views)
        if warp_perturb:
            # the reference zero-initialises the last layer of every sub-network (identity warp);
            # a small perturbation makes the warp and its gradients non-trivial for benchmarking
            with torch.no_grad():
                for name, p in self.graph.warp_mlp.named_parameters():
                    if "_1." in name or "_c." in name:
                        p.normal_(0.0, warp_perturb)
        self.nets = [self.graph.nerf] + ([self.graph.nerf_fine] if opt.nerf.fine_sampling else [])
        # optimizer groups = the flat buffers the kernels read (NeRF nets, warp network) + the latent table
        latent = self.graph.warp_latent.weight
        dev = latent.device
        self.bucket = parallel.GradBucket([n.field_parameters() for n in self.nets] +
                                          [list(self.graph.warp_mlp.parameters()), [latent]], dev)
        self.flats = [n.flat_params for n in self.nets] + [self.graph.warp_mlp.flat_params, latent.data.view(-1)]
        self.m = [torch.zeros_like(f) for f in self.flats]
        self.v = [torch.zeros_like(f) for f in self.flats]
        o = opt.optim
        self.lrs = [(o.lr, o.lr_end)] * len(self.nets) + [(o.lr_pose, o.lr_pose_end)] * 2
        self.it = 0
        if world > 1:
            opt.ray_shard = (rank, world)
            opt.loss_norm_elements = parallel.global_loss_elements(n_views, opt.nerf.rand_rays // n_views)

    def summarize_loss(self, loss):
        """reference model/base.py:130-142"""
        total = 0.
        for key in loss:
            w = self.opt.loss_weight[key]
            if w is not None:
                total = total + 10 ** float(w) * loss[key]
        loss.update(all=total)
        return loss

    def train_iteration(self, var):
        opt = self.opt
        self.it += 1
        for g in self.bucket.groups:
            for p in g:
                p.grad = None
        var = self.graph.forward(opt, var, mode="train", iter=self.it)
        loss = self.summarize_loss(self.graph.compute_loss(opt, var, mode="train"))
        loss.all.backward()
        self.bucket.gather()
        self.bucket.all_reduce()
        for i, flat in enumerate(self.flats):
            lr0, lr_end = self.lrs[i]
            gamma = (lr_end / lr0) ** (1.0 / opt.max_iter) if lr_end else 1.0
            ops.adam_step(flat, self.bucket.segment(i), self.m[i], self.v[i], lr0 * gamma ** (self.it - 1), self.it)
        for n in self.nets:
            n.set_progress(self.it / opt.max_iter)
        return loss


def synthetic_scene(opt, n_views, seed=0):
    """SURVEY section 8(d) synthetic inputs: uniform random images, pin-hole intrinsics
    fx = fy = 0.8 W, principal point at the image centre, identity initial poses."""
    gen = torch.Generator().manual_seed(seed)
    image = torch.rand(n_views, 3, opt.H, opt.W, generator=gen)
    intr = torch.tensor([[0.8 * opt.W, 0, opt.W / 2], [0, 0.8 * opt.W, opt.H / 2], [0, 0, 1]], dtype=torch.float32).repeat(n_views, 1, 1)
    pose = torch.eye(3, 4).repeat(n_views, 1, 1)
    return edict(idx=torch.arange(n_views), image=image.to(opt.device), intr=intr.to(opt.device), pose=pose.to(opt.device))
