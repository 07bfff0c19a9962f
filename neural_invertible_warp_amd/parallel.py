"""Ray-sharded data parallelism (one process per GPU, torch.distributed over RCCL/xGMI).

The reference is single-GPU (options.py:103 asserts it); this is new work.  Rays are
independent units, so the only exchange is the gradient all-reduce:
  * every rank holds all B views and the full parameter set;
  * one global pixel set is drawn per iteration, identical on every rank ("same pixels for every view",
    reference nerf_inn_llff.py:510, holds for the global batch exactly as in the reference);
  * PER WHOLE VIEW: ray generation, the NVP warp and the alignment term -- 0.4 % of the step's FLOPs.  A rank warps ALL 2R points of
    every view its share of the rays touches (view_window: B / world + 1 views or fewer), whole views because the warp's embedder window
    acts on point INDICES inside a view: every point is warped at the index the reference gives it, so a sharded run is the
    same function as the unsharded one, and the per-view Kabsch registration has all of a view's points at hand -- no collective inside
    the forward.  A view cut by a share boundary is warped by both neighbours; its alignment term is counted once, by the rank that
    holds the view's first ray.  (Warping the whole batch on every rank, the first form of this round, cost N x the warp under weak scaling.)
  * SHARDED: everything per (ray, sample) -- depth sampling, the field MLPs, compositing, the photometric loss -- on the rank's
    CONTIGUOUS share of the flattened view-major [B][R] ray list (flat_share: equal shares to within one ray).  Round 2 gave rank r
    the pixels idx[r::world] of every view; at 113 rays per view and 8 ranks that is 15 rays on rank 0 = 34,560 samples, 1,792 more
    than one 32,768-sample round of the register-chained MLP kernels, i.e. two rounds where 255 rays (32,640 samples) take one;
  * every rank normalises its photometric loss by the GLOBAL element count, so that the SUM of
    the per-rank gradients is the gradient of the global-batch mean (reference base.py:209-211);
  * ONE flat fp32 bucket (NeRF + fine NeRF + warp + latents, ~4.9 MB) is all-reduced per step:
    at this size a ring over xGMI is latency-bound, so a single in-place call beats buckets.
No collective sits on the per-sample data path, and none inside the forward.
"""
import os

import torch
import torch.distributed as dist


FORCE_COLLECTIVES = False      # issue the collectives of a ONE-rank group too (bench.py --force-dist: RCCL exercised on a 1-GPU box)


def _collectives_live():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def init_from_env(backend=None, force=False):
    """-> (rank, world, local_rank).  Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* as set by
    torch.distributed.run; backend "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU.  `force`: create the group for a single
    rank as well and send its (identity) all-reduces through the back end."""
    global FORCE_COLLECTIVES
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    FORCE_COLLECTIVES = bool(force)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = backend or os.environ.get("NIW_DIST_BACKEND")           # (gloo: several ranks on ONE GPU -- logic tests on a 1-GPU box)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_ray_idx(ray_idx, rank, world):
    """Disjoint, exhaustive strided split of a shared pixel permutation (round 1-2 partition; kept as a helper)."""
    return ray_idx[rank::world]


def flat_share(n_rays, rank, world):
    """-> (lo, hi): rank's contiguous share of a list of n_rays rays; shares are disjoint, exhaustive and equal to within one ray."""
    return (n_rays * rank) // world, (n_rays * (rank + 1)) // world


class ViewWindow:
    """What one rank handles of a batch of B views x R rays (view-major flat ray list):
         lo, hi      its contiguous share of the rays (flat_share)
         v0, v1      the views that share touches, [v0, v1): these it generates rays for and warps WHOLE
         own0, own1  the views whose first ray lies in its share, [own0, own1): their alignment terms are its to count
         local       the share as a slice of the window's own flattened rays: [lo - v0 R, hi - v0 R)"""

    def __init__(self, n_views, n_rays_per_view, rank, world):
        B, R = int(n_views), int(n_rays_per_view)
        self.B, self.R = B, R
        self.lo, self.hi = flat_share(B * R, rank, world)
        if self.hi > self.lo:
            self.v0, self.v1 = self.lo // R, (self.hi - 1) // R + 1
            self.own0, self.own1 = -(-self.lo // R), (self.hi - 1) // R + 1
        else:                                   # more ranks than rays: nothing to do
            self.v0 = self.v1 = self.own0 = self.own1 = 0
        self.own1 = max(self.own1, self.own0)
        self.local = (self.lo - self.v0 * R, self.hi - self.v0 * R)

    @property
    def views(self):
        return slice(self.v0, self.v1)

    @property
    def owned_in_window(self):
        return slice(self.own0 - self.v0, self.own1 - self.v0)


def all_reduce_sum_(t):
    """In-place SUM over ranks (identity without a process group): the [B,16] Kabsch moments of the alignment loss."""
    if _collectives_live():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def gather_owned_rows(table, win):
    """Per-view table [B, k] of which every rank has refreshed the rows of the views it handles: every view's row is taken from the rank
    that OWNS the view (zeros elsewhere, summed over ranks) and written back INTO `table` -- the storage stays where it is, because a
    captured train iteration has the table's device address baked into its launches (engine.FusedStep: `d.poses`); a fresh tensor here
    would leave every later replay writing the poses into freed memory.  Returns `table`.  Identity without a process group or window."""
    if win is None or not _collectives_live():
        return table
    out = torch.zeros_like(table)
    out[win.own0:win.own1] = table[win.own0:win.own1]
    with torch.no_grad():
        table.copy_(all_reduce_sum_(out))
    return table


def global_loss_elements(n_views, n_rays_global):
    """Element count of the global-batch MSE mean (3 colour channels)."""
    return 3 * n_views * n_rays_global


class GradBucket:
    """Flat gradient bucket over a fixed list of parameter groups.

    gather() copies every .grad into one contiguous fp32 buffer (missing grads count as zero),
    all_reduce() sums it over ranks in place (no-op for world 1), segment(i) returns the slice of
    group i for the optimizer.  `first`: groups laid out at the FRONT of the buffer (default: group order) -- the engine puts the
    group whose gradients are final earliest there (the fine network's), so that `head()` and `tail()` are the two contiguous
    pieces of a split exchange: the head travels while the backward of the rest is still running."""

    def __init__(self, groups, device, first=()):
        self.groups = [list(g) for g in groups]
        self.sizes = [sum(p.numel() for p in g) for g in self.groups]
        self.order = [i for i in first] + [i for i in range(len(self.groups)) if i not in first]      # memory order of the groups
        self.starts = [0] * len(self.groups)
        off = 0
        for i in self.order:
            self.starts[i] = off
            off += self.sizes[i]
        self.n_head = sum(self.sizes[i] for i in first)
        self.flat = torch.zeros(off, device=device, dtype=torch.float32)
        self.sunk = set()      # groups whose backward kernels write their segment directly (ops grad_sink): gather() skips them

    def head(self):
        """the `first` groups' gradients (empty without any)"""
        return self.flat[:self.n_head]

    def tail(self):
        return self.flat[self.n_head:]

    def gather(self):
        """One concatenation kernel per group (not one copy per parameter: a step is ~100 tensors)."""
        for i, g in enumerate(self.groups):
            if i in self.sunk:
                continue
            seg = self.segment(i)
            if all(p.grad is not None for p in g):
                torch.cat([p.grad.reshape(-1) for p in g], out=seg)
                continue
            off = 0
            for p in g:
                n = p.numel()
                if p.grad is None:
                    seg[off:off + n].zero_()
                else:
                    seg[off:off + n].copy_(p.grad.reshape(-1))
                off += n
        return self.flat

    def all_reduce(self):
        if _collectives_live():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        return self.flat

    def segment(self, i):
        return self.flat[self.starts[i]:self.starts[i] + self.sizes[i]]

    def scatter(self):
        """Write the (reduced) bucket back into the .grad tensors (for optimizers that read .grad)."""
        for i, g in enumerate(self.groups):
            off = self.starts[i]
            for p in g:
                n = p.numel()
                if p.grad is None:
                    p.grad = self.flat[off:off + n].view_as(p).clone()
                else:
                    p.grad.copy_(self.flat[off:off + n].view_as(p))
                off += n
