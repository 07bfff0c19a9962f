"""Training step around the Graph, with the call sequence of the reference engine for one iteration
(model/nerf_inn_llff.py:80-100 + model/barf_inn_llff.py:106-120 / model/barf_inn_dtu.py:355-369 +
model/base.py:130-142):

    zero_grad -> graph.forward(mode="train", iter=it) -> compute_loss -> summarize_loss (sum of 10^w * loss)
    -> backward -> Adam(nerf[, nerf_fine]) + Adam(warp network, latent table) -> ExponentialLR -> it += 1
    -> progress = it / max_iter

The Adam updates run as fused launches over flat buffers (niw_adam_step); under ray sharding the gradients of all
groups are summed over ranks in ONE flat all-reduce first (..parallel.GradBucket).  Two model families share it:
`barf_inn_llff` (warp modules on the graph) and `barf_inn_dtu` (warp inside INNPoseParams, metric depth from the data).
"""
import math

import torch

import ctypes

from . import _lib as _lib_mod
from . import ops, parallel
from ._lib import NiwError
from .util import edict


class CaptureError(NiwError):
    """HIP-graph capture of the train iteration failed: not recoverable in this process (HIP's capture error is sticky); a launcher
    starts a fresh process with hip_graph=False (bench.py does)"""


class StepConstants:
    """Every step-dependent SCALAR of a train iteration in one small device buffer, refreshed by one host-to-device copy per step:
    c2f band weights (10 + 4), the warp's annealing windows (6 + 6), the two step-dependent Adam scalars of every optimizer group
    and the number of the random pixel draw.  The kernels read them at run time (band_dev / window_dev / hyper_dev / draw_dev
    arguments of the C ABI) instead of receiving them by value, which is what lets a captured HIP graph of the whole iteration
    be replayed unchanged from step to step."""
    BAND, WINDOW, HYPER, DRAW, BYTES = 0, 64, 128, 240, 256          # byte offsets (DRAW 8-byte aligned)

    def __init__(self, device, n_groups):
        assert self.HYPER + 8 * n_groups <= self.DRAW
        self.n_groups = n_groups
        self.pinned = torch.device(device).type == "cuda"
        self.dev = torch.zeros(self.BYTES, dtype=torch.uint8, device=device)
        f = lambda buf, off, n: buf[off:off + 4 * n].view(torch.float32)
        self.band, self.window = f(self.dev, self.BAND, 14), f(self.dev, self.WINDOW, 12)
        self.hyper = [f(self.dev, self.HYPER + 8 * g, 2) for g in range(n_groups)]
        self.hyper_all = f(self.dev, self.HYPER, 2 * n_groups)                 # [n_groups][2], what niw_adam_step_multi reads
        self.draw = self.dev[self.DRAW:self.DRAW + 8].view(torch.int64)

    def upload(self, band, window, hyper, draw):
        """A FRESH pinned staging buffer per step: the copy is asynchronous and the host runs many steps ahead of the device, so a
        single staging buffer would be overwritten with a later step's scalars before the device has read it (round 2: un-synchronised
        graph replays trained with the pixel draws, bands and learning rates of steps still to come).  The caching host allocator
        recycles a block only after the copy that read it has completed."""
        host = torch.empty(self.BYTES, dtype=torch.uint8, pin_memory=self.pinned)
        f = lambda off, n: host[off:off + 4 * n].view(torch.float32)
        f(self.BAND, 14).copy_(torch.tensor(band, dtype=torch.float32))
        f(self.WINDOW, 12).copy_(torch.tensor(window, dtype=torch.float32))
        f(self.HYPER, 2 * self.n_groups).copy_(torch.tensor(hyper, dtype=torch.float32).reshape(-1))
        host[self.DRAW:self.DRAW + 8].view(torch.int64).fill_(int(draw))
        self.dev.copy_(host, non_blocking=True)                 # stream-ordered in front of the kernels that read it


def _sched_gamma(lr0, lr_end, sched, max_iter, what):
    """decay rate of the reference's scheduler block (nerf_inn_llff.py:40-47, barf_inn_llff.py:96-104)"""
    if not sched:
        return 1.0
    kind = sched.get("type")
    if kind != "ExponentialLR":
        raise NiwError(f"{what}.type={kind!r}: the fused Adam step implements ExponentialLR (or no scheduler)")
    if lr_end:
        return (lr_end / lr0) ** (1.0 / max_iter)
    return float(sched.get("gamma") or 1.0)



def noise_stream_seed(opt, rank, tag):
    """Philox key of a network's density-noise stream (niw_normal_rng): folds in opt.seed, the rank (ranks render different rays) and the
    network (tag 0: coarse, 1: fine -- consecutive keys, which is what niw_train_step assumes: noise_seed, noise_seed + 1)"""
    return (int(getattr(opt, "seed", 0) or 0) * 0x9E3779B97F4A7C15 + 0x4E015E + rank * 0xD1B54A32D192ED03 + tag) & (2 ** 64 - 1)


def set_noise_keys(opt, nets, draw, draw_dev=None):
    """before an iteration: every field network learns the stream its density noise comes from in THIS iteration (mirror: model/nerf.py
    NeRF._run; the one-call form gets the same key through niw_train_desc.noise_seed)"""
    rank = (getattr(opt, "ray_shard", None) or (0, 1))[0]
    for tag, net in enumerate(nets):
        net.noise_key = (noise_stream_seed(opt, rank, tag), int(draw), draw_dev)


class FusedStep:
    """One train iteration of an INN trainer -- forward, losses, backward of every stage -- as ONE library call (niw_train_step,
    csrc/niw_step.hip) into ONE persistent workspace: no autograd tape, no torch arithmetic, no allocation inside the iteration.  It
    runs the same kernels with the same launch shapes as the autograd mirror (Graph.forward + compute_loss + backward over ..ops), and
    sums the gradient routes in autograd's order, so the two train bit for bit alike (tests/test_gpu_fused_step.py); the mirror stays
    the reference-shaped interface of a stand-alone Graph and the fallback for what the call does not cover (`unsupported`)."""

    def __init__(self, trainer):
        self.tr = trainer
        self.ws = None                 # workspace, allocated at the first iteration (the batch fixes its size)
        self.ring = None               # [slots, 4] loss rows: a returned loss stays valid for `slots` further iterations (eager mode)
        self.desc = None
        self._keep = []

    @staticmethod
    def unsupported(trainer):
        """-> None when niw_train_step covers this trainer's configuration, else the reason (the autograd mirror then runs)"""
        from .model import nerf_inn_llff
        opt = trainer.opt
        if torch.device(opt.device).type != "cuda":
            return "no GPU"
        if opt.nerf.ray_sampler != "feistel":
            return "ray_sampler is not 'feistel' (draws injected through torch.randperm)"
        if opt.nerf.sample_stratified and opt.nerf.stratified_rng != "philox":
            return "stratified draws injected through torch.rand"
        vanilla = getattr(trainer, "family", None) == "vanilla"
        if opt.nerf.density_noise_reg and opt.nerf.get("density_noise_rng") != "philox":
            return "density noise injected through torch.randn"
        if opt.data.dataset == "blender" and not vanilla:
            return "blender initial poses"
        if nerf_inn_llff.ALIGN_BACKEND is not None and not vanilla:
            return "a test backend of the alignment term is plugged in"
        if trainer.n_views > 64:
            return "more than 64 views"
        if len({n._state.precision for n in trainer.nets}) != 1:
            return "networks of different precision"
        if not opt.nerf.rand_rays:
            return "nerf.rand_rays unset"
        return None

    # ------------------------------------------------------------------ descriptor
    def _weights(self):
        lw = self.tr.opt.loss_weight
        w = lambda k: -1.0 if lw.get(k) is None else 10 ** float(lw[k])
        return w("render"), w("render_fine"), w("global_alignment")

    def _build(self, var):
        tr, opt = self.tr, self.tr.opt
        g = tr.graph
        B = len(var.idx)
        if B != tr.n_views:
            raise NiwError(f"train_iteration: the batch holds {B} views, the trainer was built for {tr.n_views}")
        R = opt.nerf.rand_rays // B
        vanilla = tr.family == "vanilla"
        win = None if vanilla else g.view_window(opt, B, R)
        g._last_window = win
        S = opt.nerf.sample_intvs
        Sf = (opt.nerf.sample_intvs_fine or 0) if opt.nerf.fine_sampling else 0
        dev = var.image.device
        for k in ("image", "intr"):
            # the descriptor holds the caller's storage (its address is the batch's identity, _batch_signature): a private contiguous copy
            # would go stale behind an in-place update of the caller's tensor without anything noticing
            t = var[k]
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise NiwError(f"train_iteration: var.{k} must be a contiguous float32 tensor (got {t.dtype}, contiguous={t.is_contiguous()}); "
                               "the one-call iteration reads the batch in place")
        image, intr = ops._f32(var.image, "var.image"), ops._f32(var.intr, "var.intr")
        if image.shape[0] != B or image.shape[-1] * image.shape[-2] != opt.H * opt.W:
            raise NiwError(f"train_iteration: var.image is {tuple(image.shape)}, expected [{B},3,{opt.H},{opt.W}]")
        d = _lib_mod.TrainDesc()
        d.image, d.intr = image.data_ptr(), intr.data_ptr()
        self._keep = [image, intr]
        if vanilla:
            # the cameras the rays come from (reference model/nerf.py:264 get_pose -> var.pose): read in place, like the images
            pose = var.pose
            if pose.dtype != torch.float32 or not pose.is_contiguous() or tuple(pose.shape) != (B, 3, 4):
                raise NiwError(f"train_iteration: var.pose must be a contiguous float32 [{B},3,4] tensor (got {tuple(pose.shape)} {pose.dtype})")
            self._keep.append(pose)
            d.pose_init = pose.data_ptr()
            rng = opt.nerf.depth.range
        elif tr.family == "dtu":
            pose_init = ops._f32(tr.pose_net.initial_poses_w2c, "initial_poses_w2c")
            self._keep.append(pose_init)
            d.pose_init = pose_init.data_ptr()
            if opt.nerf.depth.param == "inverse":
                rng = opt.nerf.depth.range
            else:
                rng = g._host_depth_range(var.depth_range)
        else:
            rng = opt.nerf.depth.range
        d.n_views, d.H, d.W, d.rays_per_view = B, opt.H, opt.W, R
        if win is None:
            d.view0, d.view1, d.own0, d.own1, d.ray_lo, d.ray_hi = 0, B, 0, B, 0, B * R
        else:
            d.view0, d.view1, d.own0, d.own1, d.ray_lo, d.ray_hi = win.v0, win.v1, win.own0, win.own1, win.lo, win.hi
        n = d.ray_hi - d.ray_lo
        if n <= 0:
            raise NiwError("train_iteration: this rank's share of the rays is empty (more ranks than rays)")
        d.stratified = 1 if opt.nerf.sample_stratified else 0
        seed = int(getattr(opt, "seed", 0) or 0)
        rank = (getattr(opt, "ray_shard", None) or (0, 1))[0]
        d.pixel_seed = seed & (2 ** 64 - 1)
        d.depth_seed = (seed * 0x9E3779B97F4A7C15 + 0x5D1F + rank * 0xD1B54A32D192ED03) & (2 ** 64 - 1)      # model/nerf.py Graph.sample_depth, call 0
        d.n_samples, d.n_fine = S, Sf
        d.inverse_depth = 1 if opt.nerf.depth.param == "inverse" else 0
        d.density_activ = ops.ACT[opt.arch.density_activ]
        d.depth_min, d.depth_max = float(rng[0]), float(rng[1])
        if Sf:
            unif, bins = ops._pdf_tables(S, Sf, opt.nerf.depth.range, dev)          # bins: the yaml range, also for DTU (nerf_inn_dtu.py:549)
            d.unif, d.bins = unif.data_ptr(), bins.data_ptr()
        d.precision = ops.PREC[tr.nets[0]._state.precision]
        if d.precision == 0:
            d.pack_index = ops.pack_index(dev).data_ptr()
        d.use_index_window = 0 if vanilla else (1 if tr.warp_mlp.reference_exact else 0)
        d.w_render, d.w_render_fine, d.w_align = self._weights()
        if vanilla:
            d.w_align = -1.0
        d.density_noise = float(opt.nerf.density_noise_reg or 0.0)
        d.noise_seed = noise_stream_seed(opt, rank, 0)
        d.ndc, d.ndc_near = (1 if opt.camera.ndc else 0), 1.0
        d.has_bg, d.bg = (1 if opt.nerf.setbg_opaque else 0), float(opt.data.get("bgcolor") or 0.0)
        if not Sf:
            d.w_render_fine = -1.0
        d.always_register = 1 if tr.family == "dtu" else 0
        d.overlap = 1 if tr.overlap else 0
        d.mse_norm = float(getattr(opt, "loss_norm_elements", None) or 3 * B * R)
        if vanilla:
            table = None
        elif tr.family == "dtu":
            table = tr.pose_net.pose_global.weight
        else:
            table = g.global_rigid.weight if hasattr(g, "global_rigid") else None
        self.pose_table = table
        self.rgb = torch.empty(n, 3, device=dev)
        self.rgb_fine = torch.empty(n, 3, device=dev) if Sf else None
        d.rgb, d.rgb_fine = self.rgb.data_ptr(), (self.rgb_fine.data_ptr() if Sf else None)
        self.shape = (B, R, S, Sf, n, win)
        self._built_for = self._batch_signature(var)
        self.desc = d
        self._refresh(var)
        if self.ring is None:
            self.ring = torch.zeros(1 if tr.hip_graph else 64, 4, device=dev)
        d.loss = self.ring.data_ptr()
        floats = _lib_mod.load().niw_train_step_workspace_floats(ctypes.byref(d))
        if floats <= 0:
            raise NiwError(_lib_mod.load().niw_last_error_string().decode())
        if self.ws is None or self.ws.numel() != floats:
            self.ws = None                                 # (release the old block before asking for the new one)
            self.ws = torch.empty(floats, device=dev, dtype=torch.float32)

    def _refresh(self, var):
        """the members that can change between two iterations without a new descriptor: parameter / gradient / table storage"""
        tr, d = self.tr, self.desc
        tr._install_grad_sinks()
        n_nets = len(tr.nets)
        d.nerf_params, d.d_nerf = tr.nets[0].flat_params.data_ptr(), tr.bucket.segment(0).data_ptr()
        if n_nets > 1:
            d.nerf_fine_params, d.d_nerf_fine = tr.nets[1].flat_params.data_ptr(), tr.bucket.segment(1).data_ptr()
        if tr.family != "vanilla":
            d.warp_params, d.d_warp = tr.warp_mlp.flat_params.data_ptr(), tr.bucket.segment(n_nets).data_ptr()
            latent = tr.warp_latent.weight
            if not latent.is_contiguous():
                raise NiwError("the latent table must be contiguous")
            d.latent, d.d_latent = latent.data_ptr(), tr.bucket.segment(n_nets + 1).data_ptr()
        table = self.pose_table
        d.poses = None if table is None else table.data.data_ptr()
        d.fine_grads_ready = tr.fine_ready_event()

    def _batch_signature(self, var):
        """what the descriptor was built from: the caller's tensors (storage AND content version: the DTU depth range is read to the host
        when the descriptor is built), the batch shape and the loss weights"""
        opt = self.tr.opt
        sig = [len(var.idx), opt.nerf.rand_rays, opt.nerf.sample_intvs, opt.nerf.sample_intvs_fine if opt.nerf.fine_sampling else 0, self._weights(),
               bool(opt.nerf.sample_stratified), tuple(opt.nerf.depth.range), opt.nerf.depth.param]
        sig.append((bool(opt.camera.ndc), float(opt.nerf.density_noise_reg or 0.0), bool(opt.nerf.setbg_opaque), float(opt.data.get("bgcolor") or 0.0)))
        for k in ("image", "intr", "depth_range") + (("pose",) if self.tr.family == "vanilla" else ()):
            t = var.get(k) if hasattr(var, "get") else getattr(var, k, None)
            sig.append(None if t is None else (t.data_ptr(), tuple(t.shape), t._version if k == "depth_range" else 0))
        return sig

    def _same_batch(self, var):
        return self._batch_signature(var) == self._built_for

    # ------------------------------------------------------------------ one iteration
    def run(self, var, it):
        """forward + backward of 0-based iteration `it` on the resident batch -> loss edict (views of a device row: render[, render_fine]
        [, global_alignment], all); the gradients of all optimizer groups are in the trainer's bucket afterwards"""
        tr, opt = self.tr, self.tr.opt
        if self.desc is None or not self._same_batch(var):
            self._build(var)
        else:
            self._refresh(var)
        d = self.desc
        from .model.nvp import nvp_ndr
        net = tr.nets[0]
        consts = tr.consts if tr.hip_graph else None
        b3, bv = ops._farr(net.band_weights(opt, ops.L3D), ops.L3D), ops._farr(net.band_weights(opt, ops.LVIEW), ops.LVIEW)
        fp = ctypes.POINTER(ctypes.c_float)
        d.band_w3d, d.band_wview = ctypes.cast(b3, fp), ctypes.cast(bv, fp)
        if tr.family != "vanilla":
            chan_w, index_window = tr.warp_mlp._anneal(float(nvp_ndr.embedding_anneal_ratio(opt, it)))
            cw = ops._farr(chan_w, 6)
            iw = None if index_window is None else ops._farr(index_window, 6)
            d.chan_w = ctypes.cast(cw, fp)
            d.index_window = ctypes.cast(iw, fp) if iw is not None else None
        d.band_dev = None if consts is None else consts.band.data_ptr()
        d.window_dev = None if consts is None else consts.window.data_ptr()
        d.draw_dev = None if consts is None else consts.draw.data_ptr()
        d.draw = int(it) + 1
        row = self.ring[0 if consts is not None else it % self.ring.shape[0]]
        d.loss = row.data_ptr()
        B, R, S, Sf, n, win = self.shape
        st = ops._stream()
        if ops.TIMING.enabled:
            # per-stage device events (bench.py's kernel table): the same call, stage by stage
            units = {"mlp_fwd": n * S, "mlp_bwd_dx": n * S, "mlp_bwd_dw": n * S, "composite_fwd": n * S, "composite_bwd": n * S,
                     "mlp_fwd_fine": n * (S + Sf), "mlp_bwd_dx_fine": n * (S + Sf), "mlp_bwd_dw_fine": n * (S + Sf), "composite_fwd_fine": n * (S + Sf),
                     "composite_bwd_fine": n * (S + Sf)}
            alias = {"mlp_fwd": "mlp_fwd_train", "mlp_fwd_fine": "mlp_fwd_train"}
            # passes whose compositing, photometric residual and their backward are ONE launch (niw_composite_mse_train, csrc/niw_step.hip:
            # the span kernels cover the sample count and the pass has a loss): the launch sits in the composite_fwd stage, the pass's
            # composite_bwd stage is empty
            import os
            one = os.environ.get("NIW_TRAIN_ONE_LAUNCH_LOSS", "1") != "0"
            one = one and not d.has_bg                      # (an opaque background: the three-launch form)
            fused_pass = {"": one and d.w_render >= 0 and S % 4 == 0 and S <= 256,
                          "_fine": one and Sf > 0 and d.w_render_fine >= 0 and (S + Sf) % 4 == 0 and S + Sf <= 256}
            for k, name in enumerate(_lib_mod.TRAIN_STAGES):
                if name.endswith("_fine") and not Sf or name == "resample" and not Sf:
                    continue
                suffix = "_fine" if name.endswith("_fine") else ""
                if name.startswith("composite_bwd") and fused_pass[suffix]:
                    _lib_mod.call("niw_train_step", ctypes.byref(d), ops._p(self.ws), k, k + 1, st)      # (launches nothing)
                    continue
                if name.startswith("composite_fwd") and fused_pass[suffix]:
                    alias[name] = "composite_train"
                base = name[:-5] if name.endswith("_fine") else name
                with ops.timed(alias.get(name, base), units.get(name, n)):
                    _lib_mod.call("niw_train_step", ctypes.byref(d), ops._p(self.ws), k, k + 1, st)
        else:
            _lib_mod.call("niw_train_step", ctypes.byref(d), ops._p(self.ws), 0, len(_lib_mod.TRAIN_STAGES), st)
        loss = edict()
        if d.w_render >= 0:
            loss.render = row[0]
        if d.w_render_fine >= 0:
            loss.render_fine = row[1]
        if d.w_align >= 0:
            loss.global_alignment = row[2]
        loss.all = row[3]
        var.rgb = self.rgb.view(B, R, 3) if win is None else self.rgb.view(1, n, 3)
        if Sf:
            var.rgb_fine = self.rgb_fine.view(B, R, 3) if win is None else self.rgb_fine.view(1, n, 3)
        var.view_window = win
        return loss


class INNTrainer:
    def __init__(self, opt, n_views, rank=0, world=1, warp_perturb=0.0, seed=0, initial_poses_w2c=None, ray_sampler=None, hip_graph=False,
                 fused_step="auto", overlap=True, split_exchange="auto", collectives=True):
        """ray_sampler: "feistel" (one sort-free launch, default on the GPU) or "randperm" (the reference's torch.randperm call).
        hip_graph: capture the whole iteration (forward, backward, gradient gather, Adam) into a HIP graph after two eager
        steps and replay it from then on -- one graph launch + one 256-byte constants upload per step instead of ~140 launches;
        needs inputs that stay at the same device addresses from step to step (the engine's resident `var` tensors do).
        fused_step: "auto" (default) runs the iteration as one niw_train_step call (FusedStep) wherever that call covers the
        configuration and through the autograd mirror elsewhere; True insists (NiwError otherwise); False = always the mirror.
        split_exchange (ranks > 1, a fine network present): the gradient exchange as TWO all-reduces -- the fine network's segment,
        final a third of the way into the backward, on a communication stream while the coarse network's and the warp's backward
        still run; the rest behind the backward.  "auto": where the iteration is one launched niw_train_step call (the library records
        an event when the segment is final); True: everywhere (without that event the two calls simply follow the backward); False:
        one flat all-reduce (rounds 1-4).  Same sums either way.
        collectives=False: this trainer is a REPLICA (one scene per GPU, bench.py --placement replicas): it never exchanges anything,
        whatever process group is live."""
        self.opt, self.rank, self.world, self.n_views = opt, rank, world, n_views
        o = opt.optim
        if o.get("algo", "Adam") != "Adam":
            raise NiwError(f"optim.algo={o.algo!r}: only Adam is fused (niw_adam_step)")
        torch.manual_seed(seed)                       # identical initial weights on every rank
        self.family = "dtu" if opt.model == "barf_inn_dtu" else "llff"
        if self.family == "dtu":
            from .model import barf_inn_dtu
            from .model.pose_models.inn import INNPoseParams
            if initial_poses_w2c is None:
                raise NiwError("barf_inn_dtu needs initial_poses_w2c (Model.set_initial_poses)")
            self.pose_net = INNPoseParams(opt, num_poses=n_views, initial_poses_w2c=initial_poses_w2c, device=opt.device)
            self.graph = barf_inn_dtu.Graph(opt, self.pose_net)
            self.warp_mlp, self.warp_latent = self.pose_net.pose_embedding, self.pose_net.pose_latent
            train_warp = train_latent = True          # barf_inn_dtu.py:341-345: both groups, unconditionally
        else:
            from .model import barf_inn_llff
            self.graph = barf_inn_llff.Graph(opt).attach_warp(opt, n_views)
            self.warp_mlp, self.warp_latent = self.graph.warp_mlp, self.graph.warp_latent
            train_warp = bool(opt.inn.get("optimize", {}).get("enabled", True))
            train_latent = bool(opt.warp_latent.get("optimize", {}).get("enabled", True))
        if warp_perturb:
            # the reference zero-initialises the last layer of every sub-network (identity warp);
            # a small perturbation makes the warp and its gradients non-trivial for benchmarking
            with torch.no_grad():
                for name, p in self.warp_mlp.named_parameters():
                    if "_1." in name or "_c." in name:
                        p.normal_(0.0, warp_perturb)
        self.nets = [self.graph.nerf] + ([self.graph.nerf_fine] if opt.nerf.fine_sampling else [])
        # optimizer groups = the flat buffers the kernels read (NeRF nets, warp network) + the latent table
        latent = self.warp_latent.weight
        dev = latent.device
        # (the fine network's segment leads the buffer: it is final first, and head / tail are the two pieces of a split exchange)
        self.bucket = parallel.GradBucket([n.field_parameters() for n in self.nets] +
                                          [list(self.warp_mlp.parameters()), [latent]], dev, first=(1,) if len(self.nets) > 1 else ())
        self.split_exchange, self.collectives = split_exchange, bool(collectives)
        # the backward kernels write every group's gradient straight into its bucket segment (ops grad_sink): no .grad tensors, no
        # per-parameter accumulation copies, no gather
        self._install_grad_sinks()
        flats = self._flats()
        self.m = [torch.zeros_like(f) for f in flats]
        self.v = [torch.zeros_like(f) for f in flats]
        g_main = _sched_gamma(o.lr, o.get("lr_end"), o.get("sched", {"type": "ExponentialLR"}), opt.max_iter, "optim.sched")
        g_pose = _sched_gamma(o.lr_pose, o.get("lr_pose_end"), o.get("sched_pose", {"type": "ExponentialLR"}), opt.max_iter, "optim.sched_pose")
        n = len(self.nets)
        self.lrs = [(o.lr, o.get("lr_end"))] * n + [(o.lr_pose, o.get("lr_pose_end"))] * 2       # (checkpoint.py reads these)
        self.gammas = [g_main] * n + [g_pose] * 2
        self.trainable = [True] * n + [train_warp, train_latent]
        if not (train_warp or train_latent):
            # the reference would build torch.optim.Adam([]) here and fail inside it (barf_inn_llff.py:84-95)
            raise NiwError("inn.optimize.enabled and warp_latent.optimize.enabled are both false: the pose optimizer has no parameters")
        self._warmup_group = n if train_warp else n + 1          # param_groups[0] of the reference's optim_pose
        self.warmup_pose = o.get("warmup_pose")
        self.it = 0
        on_gpu = torch.device(opt.device).type == "cuda"
        opt.nerf.ray_sampler = ray_sampler or opt.nerf.get("ray_sampler") or ("feistel" if on_gpu else "randperm")
        # stratified depth draws: inside the kernel (Philox) wherever the engine makes its own pixel draw; a harness that injects the
        # reference's torch.randperm / torch.rand draws (ray_sampler="randperm") keeps torch.rand
        opt.nerf.stratified_rng = opt.nerf.get("stratified_rng") or ("philox" if opt.nerf.ray_sampler == "feistel" else "torch")
        # density noise (nerf.density_noise_reg): drawn by niw_normal_rng wherever the engine makes its own draws, torch.randn for a harness
        opt.nerf.density_noise_rng = opt.nerf.get("density_noise_rng") or ("philox" if opt.nerf.ray_sampler == "feistel" else "torch")
        self.hip_graph = bool(hip_graph) and on_gpu
        self._captured = None
        if self.hip_graph:
            if opt.nerf.ray_sampler != "feistel":
                raise NiwError("hip_graph=True needs ray_sampler='feistel' (torch.randperm cannot be replayed with a fresh draw)")
            self.consts = StepConstants(dev, len(self.bucket.groups))
        self._bind_constants(False)                      # outside a train iteration the modules read host state (see _bind_constants)
        self.overlap = bool(overlap)                      # niw_train_desc.overlap: small independent stages on a second stream
        self.comm_events = None                          # a list: device events around every gradient all-reduce are appended (bench.py comm_ms)
        self.fused = None
        if fused_step:
            why = FusedStep.unsupported(self)
            if why is None:
                self.fused = FusedStep(self)
                if self.overlap:
                    _lib_mod.call("niw_train_step_prepare")      # the second stream exists before any later stream user creates theirs
            elif fused_step is True:
                raise NiwError(f"fused_step=True: niw_train_step does not cover this configuration ({why})")
        self.fused_fallback_reason = None if self.fused is not None else (FusedStep.unsupported(self) if fused_step else "fused_step=False")
        if self.collectives and (world > 1 or parallel.FORCE_COLLECTIVES):      # (a forced one-rank group shards 1-way: same arithmetic, collectives issued)
            opt.ray_shard = (rank, world)
            opt.loss_norm_elements = parallel.global_loss_elements(n_views, opt.nerf.rand_rays // n_views)

    def _install_grad_sinks(self):
        """(re-)attach the bucket segments as gradient sinks: cheap, and done every iteration so that a module that was re-created
        or re-flattened in between cannot train against a stale buffer"""
        n = len(self.nets)
        for i, net in enumerate(self.nets):
            net.grad_sink = self.bucket.segment(i)
        self.warp_mlp.grad_sink = (self.bucket.segment(n), self.bucket.segment(n + 1))
        self.bucket.sunk = set(range(n + 2))

    def _flats(self):
        """The flat parameter buffers the kernels read, looked up every step: re-flattening (after .to() or an external
        re-assignment of parameter storage) replaces them, and Adam must update the buffer that is actually in use."""
        return [n.flat_params for n in self.nets] + [self.warp_mlp.flat_params, self.warp_latent.weight.data.view(-1)]

    def learning_rate(self, group, it):
        """lr of optimizer group `group` in 0-based iteration `it` (scheduler stepped `it` times; the reference's linear
        pose warm-up touches param_groups[0] of optim_pose only (barf_inn_llff.py:108-111): the warp network when it is trained,
        otherwise the latent table, which is then the first -- and only -- group of that optimizer, :84-95)"""
        lr = self.lrs[group][0] * self.gammas[group] ** it
        if self.warmup_pose and group == self._warmup_group:
            lr *= min(1.0, it / self.warmup_pose)
        return lr

    def summarize_loss(self, loss):
        """reference model/base.py:130-142"""
        total = 0.
        for key in loss:
            w = self.opt.loss_weight[key]
            if w is not None:
                total = total + 10 ** float(w) * loss[key]
        loss.update(all=total)
        return loss

    # ------------------------------------------------------------------ one iteration
    def _bind_constants(self, on):
        """Point the modules at the device-resident step constants for the duration of ONE train iteration of this engine, and back
        to None after it: every other caller of the same modules (validation, evaluation, a render between two steps) then gets
        its c2f bands / annealing windows / pixel draw by value from the host state, as without hip_graph.  (Round 2 left the
        pointers in place: a validate() before the first step rendered with an all-zero band table, later ones with the bands of
        the previous train iteration.)"""
        consts = self.consts if (on and self.hip_graph) else None
        for n in self.nets:
            n.band_dev = None if consts is None else consts.band
        self.warp_mlp.window_dev = None if consts is None else consts.window
        self.graph.draw_dev = None if consts is None else consts.draw

    def _forward_backward(self, var, it):
        opt = self.opt
        if self.fused is not None:
            # what the one-call form does not cover is decided when the trainer is built; a switch flipped afterwards (density noise, an
            # opaque background, a plugged-in alignment backend, a network moved to another precision) must not be silently ignored
            why = FusedStep.unsupported(self)
            if why is not None:
                raise NiwError(f"train_iteration: the configuration changed after the trainer was built and niw_train_step does not cover it "
                               f"any more ({why}); build a new trainer (fused_step='auto' then takes the autograd mirror)")
            return self.fused.run(var, it)
        self._install_grad_sinks()
        self._bind_constants(True)
        set_noise_keys(opt, self.nets, it + 1, self.consts.draw if self.hip_graph else None)
        try:
            var = self.graph.forward(opt, var, mode="train", iter=it)
            loss = self.graph.compute_loss(opt, var, mode="train")
            self._backward_weighted(loss)
        finally:
            self._bind_constants(False)
        self.bucket.gather()
        return loss

    def _backward_weighted(self, loss):
        """loss.all = sum of 10^w * loss_k (reference base.py:130-142) and its backward pass.  Same arithmetic as summarize_loss +
        loss.all.backward(), in fewer launches: the weights enter the backward as the terms' incoming gradients (persistent device
        scalars) instead of through an autograd chain of multiplications and additions, and the reported total is formed without a
        tape, one fused multiply-add per term."""
        keys = [k for k in loss if self.opt.loss_weight[k] is not None]
        if not hasattr(self, "_loss_w"):
            self._loss_w = {}
        for k in keys:
            w = 10 ** float(self.opt.loss_weight[k])
            if k not in self._loss_w or self._loss_w[k][0] != w:
                self._loss_w[k] = (w, torch.tensor(w, dtype=torch.float32, device=loss[k].device))
        with torch.no_grad():
            total = None
            for k in keys:
                w = self._loss_w[k][0]
                total = loss[k] * w if total is None else torch.add(total, loss[k], alpha=w)
        loss.update(all=total)
        torch.autograd.backward([loss[k] for k in keys], [self._loss_w[k][1] for k in keys])

    def _collectives_live(self):
        return self.collectives and parallel._collectives_live()

    def _overlapped_exchange(self):
        """the fine network's segment travels while the backward is still running: only where the backward is ONE launched library
        call that records an event when that segment is final (a captured iteration keeps its eager flat all-reduce between graphs).
        "auto" also asks that the rank's field launches span at least two rounds of workgroups (>= 65,536 samples in the coarse pass): a
        collective's kernel that holds a few CUs while a ONE-round launch is dispatched makes that launch double up on the remaining CUs
        (measured with the library's own side kernels, HISTORY.md round 5: 431 us instead of 239) -- at a 1/8 share of the reference's
        batch the exchange therefore stays behind the backward, at the weak-scaled sizes it travels beside it.  True forces it."""
        if not (self.split_exchange in ("auto", True) and self.fused is not None and not self.hip_graph and self.bucket.n_head > 0 and
                self._collectives_live() and self.bucket.flat.is_cuda):
            return False
        if self.split_exchange is True:
            return True
        shape = getattr(self.fused, "shape", None)
        return shape is None or shape[4] * shape[2] >= 65536          # rays of the share x coarse samples per ray

    def fine_ready_event(self):
        """-> raw handle of the event niw_train_step records behind the fine network's weight gradient, or None when no split exchange
        will wait for it.  (A torch event owns a HIP event only once it has been recorded: it is recorded here, once.)"""
        if not self._overlapped_exchange():
            return None
        if getattr(self, "_fine_ready", None) is None:
            self._fine_ready = torch.cuda.Event()
            self._fine_ready.record()
            self._comm = torch.cuda.Stream()
        return self._fine_ready.cuda_event

    def _all_reduce(self):
        """The gradient exchange of an iteration: in-place sums of the flat bucket over the ranks (no-op without a live group) -- one
        all-reduce, or two with the first one overlapped with the backward (split_exchange).  With `comm_events` a list, device events
        are appended: ([(begin, end) of every collective], (end of the backward, start of the optimizer) = the exposed part)."""
        if not self._collectives_live():
            return
        import torch.distributed as dist
        bucket, timing = self.bucket, self.comm_events is not None and self.bucket.flat.is_cuda
        ev = lambda: torch.cuda.Event(enable_timing=True)
        split = bucket.n_head > 0 and (self.split_exchange is True or self._overlapped_exchange())
        if not split:
            a, b = (ev(), ev()) if timing else (None, None)
            if timing:
                a.record()
            bucket.all_reduce()
            if timing:
                b.record()
                self.comm_events.append(([(a, b)], (a, b)))
            return
        pairs = []
        if self._overlapped_exchange() and getattr(self, "_fine_ready", None) is not None:
            main = torch.cuda.current_stream()
            self._comm.wait_event(self._fine_ready)              # recorded by niw_train_step behind the fine network's dW reduction
            with torch.cuda.stream(self._comm):
                a, b = (ev(), ev()) if timing else (None, None)
                if timing:
                    a.record()
                dist.all_reduce(bucket.head(), op=dist.ReduceOp.SUM)      # (stream-ordered on the communication stream)
                if timing:
                    b.record()
                    pairs.append((a, b))
            e0 = ev() if timing else None
            if timing:
                e0.record()                                      # the main stream has finished the backward here
            a2, b2 = (ev(), ev()) if timing else (None, None)
            if timing:
                a2.record()
            dist.all_reduce(bucket.tail(), op=dist.ReduceOp.SUM)
            if timing:
                b2.record()
                pairs.append((a2, b2))
            main.wait_stream(self._comm)
            if timing:
                e1 = ev()
                e1.record()
                self.comm_events.append((pairs, (e0, e1)))
            return
        # split, not overlapped (the autograd mirror, CPU ranks): the same two sums behind the backward
        e0 = ev() if timing else None
        if timing:
            e0.record()
        for piece in (bucket.head(), bucket.tail()):
            a, b = (ev(), ev()) if timing else (None, None)
            if timing:
                a.record()
            dist.all_reduce(piece, op=dist.ReduceOp.SUM)
            if timing:
                b.record()
                pairs.append((a, b))
        if timing:
            self.comm_events.append((pairs, (e0, pairs[-1][1])))

    def _optimizer_step(self, it):
        """torch.optim.Adam's update of every trained group (reference nerf.py:34-38, barf_inn_llff.py:84-104: two optimizers stepped one
        after the other) as ONE launch over the flat buffers"""
        groups = [(flat, self.bucket.segment(i), self.m[i], self.v[i], self.learning_rate(i, it), it + 1) if self.trainable[i] else None
                  for i, flat in enumerate(self._flats())]
        ops.adam_step_multi(groups, hyper_dev=self.consts.hyper_all if self.hip_graph else None)

    def _upload_constants(self, it):
        """the scalars of 0-based iteration `it` (progress it / max_iter was set after the previous step, barf_inn_llff.py:117)"""
        from .model.nvp import nvp_ndr
        net = self.nets[0]
        band = net.band_weights(self.opt, ops.L3D) + net.band_weights(self.opt, ops.LVIEW)
        chan_w, index_window = self.warp_mlp._anneal(float(nvp_ndr.embedding_anneal_ratio(self.opt, it)))
        hyper = [ops.adam_hyper(self.learning_rate(i, it), it + 1) for i in range(len(self.bucket.groups))]
        self.consts.upload(band, list(chan_w) + list(index_window or [1.0] * 6), hyper, it + 1)

    def train_iteration(self, var, replay=True):
        """One iteration on the resident batch `var` (idx, image, intr[, pose, depth_range]) -> loss edict.  With hip_graph the
        returned tensors are the graph's static outputs: read them before the next call.  Launched, the one-call iteration returns VIEWS
        of a 64-row ring of loss rows: a loss stays valid for the next 63 iterations -- clone what is kept longer (windowed logging).
        replay=False runs the iteration launch by launch even when a graph exists (per-kernel timing)."""
        it = self.it                                   # the reference's self.it during the step (0-based)
        if self.hip_graph:
            loss = self._graph_iteration(var, it, replay)
        else:
            loss = self._forward_backward(var, it)
            self._all_reduce()
            self._optimizer_step(it)
        self.it = it + 1
        for n in self.nets:
            if hasattr(n, "set_progress"):
                # (host copy only: the kernels take the bands by value / from the step constants; the Parameter itself is written when
                # state is read, sync_state() -- a fill launch per network and step otherwise)
                n.set_progress(self.it / self.opt.max_iter, device_copy=False)
        return loss

    def _graph_iteration(self, var, it, replay=True):
        self._upload_constants(it)
        # the iteration -- warm-up, capture and every replay alike -- reads the trainer's PRIVATE copies of the batch tensors, in place
        if getattr(self, "_static_inputs", None) is None:
            mine = self._private_batch(var)
        else:
            self._check_static_inputs(var)
            mine = type(var)(var)
            for k, (t, _) in self._static_inputs.items():
                mine[k] = t
        if self._captured is None or not replay:
            self._eager_runs = getattr(self, "_eager_runs", 0)
            if not replay or self._eager_runs < 2 or not self._capture(mine, it):
                # (two launch-by-launch iterations of THIS trainer come first, whatever `it` is: a resumed run starts at a large one)
                self._eager_runs += 1
                # warm-up (allocator, lazily built tables, kernel attributes) and fall-back: the same body, launch by launch, on the
                # side stream the capture will use (autograd's gradient accumulators stay tied to the stream they first ran on)
                side = self._side_stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    loss = self._forward_backward(mine, it)
                    self._all_reduce()
                    self._optimizer_step(it)
                torch.cuda.current_stream().wait_stream(side)
                return loss
        if self._capture_addresses() != self._captured_addresses:
            # a tensor the captured launches address has moved (parameters re-flattened by .to(), a table re-assigned from outside): the
            # graph would read and write the old storage -- record the iteration again instead
            self._captured = None
            return self._graph_iteration(var, it, replay)
        fb, adam, loss = self._captured
        fb.replay()
        if adam is not None:                           # ranks exchange gradients between the two graphs
            self._all_reduce()
            adam.replay()
        return loss

    def _check_static_inputs(self, var):
        """A replay reads PRIVATE copies of the batch tensors (made at capture time), never the caller's storage.  The copy of a tensor
        is refreshed -- stream-ordered in front of the replay -- whenever the caller hands over another tensor, other storage or a
        tensor written since (its version counter); the same untouched tensor costs nothing.  A shape / dtype / device change is an
        error: a replay never trains silently on stale inputs, and never writes into memory the caller owns (round 3 copied a new
        batch INTO the first batch's tensor, corrupting a harness that alternates batches)."""
        for k, (mine, seen) in self._static_inputs.items():
            t = var.get(k) if hasattr(var, "get") else getattr(var, k, None)
            if t is None:
                raise NiwError(f"train_iteration: the captured iteration reads var.{k}, which this call does not provide")
            if not isinstance(t, torch.Tensor):
                raise NiwError(f"train_iteration: var.{k} was a tensor when the iteration was captured, now {type(t).__name__}")
            if t.shape != mine.shape or t.dtype != mine.dtype or t.device != mine.device:
                raise NiwError(f"train_iteration: var.{k} is {tuple(t.shape)} {t.dtype} on {t.device}, the captured iteration was recorded with "
                               f"{tuple(mine.shape)} {mine.dtype} on {mine.device}; build a new trainer (or hip_graph=False) for another batch shape")
            now = (t.data_ptr(), t._version)
            if now != seen:
                mine.copy_(t)
                self._static_inputs[k] = (mine, now)
        for k, v in var.items():
            if isinstance(v, torch.Tensor) and k not in self._static_inputs:
                raise NiwError(f"train_iteration: var.{k} was not part of the captured iteration")

    def _private_batch(self, var):
        """trainer-owned copies of the batch tensors the captured iteration will read in place -> a var of the same type over them"""
        self._static_inputs = {k: (v.clone(), (v.data_ptr(), v._version)) for k, v in var.items() if isinstance(v, torch.Tensor)}
        mine = type(var)(var)
        for k, (t, _) in self._static_inputs.items():
            mine[k] = t
        return mine

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        return self._side

    def _capture(self, var, it):
        """Capture iteration `it` (it also executes on replay, below).  One graph for a single rank; under ray sharding two graphs
        (forward + backward + gather | Adam) with the RCCL all-reduce of the flat gradient bucket issued eagerly in between, so that
        no collective is captured.  A capture that fails while recording is an error (NiwError)."""
        import sys
        import torch.distributed as dist
        live_group = parallel._collectives_live()
        # (round 3: no collective sits inside the forward any more -- the warp and the alignment term are replicated, ..parallel --
        # so a sharded iteration is captured like any other: forward + backward + gather | all-reduce, eager | Adam)
        torch.cuda.synchronize()                    # (`var` holds the trainer's own copies of the batch: the graph reads them in place)
        # No cyclic garbage collection while the stream is capturing (round 6): a collection that starts inside the capture window finalises
        # whatever unreachable objects exist at that moment -- graphs, events and streams of trainers long out of use among them -- and a
        # HIP call from such a finaliser on the capturing thread aborts the process ("Fatal Python error: Aborted ... Garbage-collecting",
        # seen once the iteration allocated enough Python objects to cross a collection threshold inside the window).  Collect first,
        # then keep the collector off until the capture has ended.
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            # ranks with a live RCCL communicator: its watchdog thread polls events while we capture, which "global" error mode
            # would treat as a capture violation
            mode = dict(capture_error_mode="thread_local") if live_group else {}
            fb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(fb, stream=self._side_stream(), **mode):
                loss = self._forward_backward(var, it)
                if self.world == 1:
                    self._optimizer_step(it)
            adam = None
            if self.world > 1:
                adam = torch.cuda.CUDAGraph()
                with torch.cuda.graph(adam, pool=fb.pool(), stream=self._side_stream(), **mode):
                    self._optimizer_step(it)
        except Exception as e:  # noqa: BLE001
            # a capture that was invalidated half-way leaves HIP's capture error sticky on this process (measured: neither a new
            # stream nor CUDAGraph.reset() nor reseeding the generator brings launches back), so there is no eager fall-back to
            # offer: fail loudly and name the switch
            raise CaptureError(f"HIP-graph capture of the train iteration failed ({type(e).__name__}: {e}); "
                               "run with hip_graph=False (bench.py --hip-graph off)") from e
        finally:
            if gc_was_on:
                gc.enable()
        self._captured = (fb, adam, loss)
        self._captured_addresses = self._capture_addresses()
        return True

    def _capture_addresses(self):
        """device addresses a captured iteration has baked into its launches: parameters, gradient bucket, Adam moments, pose table"""
        t = [f.data_ptr() for f in self._flats()] + [self.bucket.flat.data_ptr()] + [x.data_ptr() for x in self.m + self.v]
        table = self.pose_net.pose_global.weight if self.family == "dtu" else getattr(self.graph, "global_rigid", None)
        if table is not None:
            t.append((table.weight if hasattr(table, "weight") else table).data.data_ptr())
        return t

    def _give_up_capture(self, why):
        import sys
        print(f"[niw] HIP-graph capture not used ({why}); the iteration is launched kernel by kernel", file=sys.stderr, flush=True)
        self.hip_graph_failed = True
        self._captured = None
        self.hip_graph = False
        self._bind_constants(False)
        return False

    def sync_state(self, collective=True):
        """bring device-side mirrors of host state up to date (the c2f `progress` Parameter is only written on demand under
        graph replay): call before reading state_dict().  Under ray sharding the per-view pose tables are collected from the ranks
        that own the views -- an all-reduce, so EVERY rank must make this call at the same point of its program (validate() and
        save_checkpoint() of the Model do, before any rank gate); `collective=False` is the host-only part, for a caller that runs on
        one rank after all ranks have synchronised (checkpoint.save_checkpoint on rank 0)."""
        for n in self.nets:
            if hasattr(n, "set_progress"):
                n.set_progress(self.it / self.opt.max_iter if self.it else float(n.progress_host or 0.0))
        if self.world > 1 and collective and self.collectives:
            # per-view pose tables: under ray sharding every rank has refreshed the rows of its own views only
            win = getattr(self.graph, "_last_window", None)
            if self.family == "dtu":
                parallel.gather_owned_rows(self.pose_net.pose_global.weight.data, win)       # in place: a captured iteration holds the address
            else:
                self.graph.gather_global_rigid()


class NeRFTrainer:
    """One train iteration of the VANILLA model (reference model/nerf.py:77-100 train_iteration + :251-288 Graph.forward / compute_loss:
    ground-truth poses, no warp -- BASELINE configs[0], options/nerf_llff_repr.yaml) on the engine's plumbing: the Feistel pixel draw, the
    in-kernel stratified draw and the in-kernel density noise instead of torch.randperm / torch.rand / torch.randn, gradients written
    straight into a flat bucket, one Adam launch over both networks with the reference's ExponentialLR.
    fused_step (round 6): "auto" runs the iteration as ONE niw_train_step call (warp_params = NULL: rays of the cameras var.pose, NDC,
    density noise, both passes, losses, backward without the ray-gradient tail of the dX chain) wherever that call covers the options;
    False (or an uncovered option) = the reference-shaped mirror (Graph.render over ..ops under autograd).  Both train bit for bit alike
    (tests/test_gpu_fused_step.py)."""
    family, hip_graph, overlap, consts, rank, world = "vanilla", False, False, None, 0, 1

    def __init__(self, opt, n_views, seed=0, fused_step="auto"):
        from .model import nerf as nerf_model
        torch.manual_seed(seed)
        self.opt, self.n_views, self.it = opt, n_views, 0
        self.graph = nerf_model.Graph(opt).to(opt.device)
        self.nets = [self.graph.nerf] + ([self.graph.nerf_fine] if opt.nerf.fine_sampling else [])
        dev = torch.device(opt.device)
        self.bucket = parallel.GradBucket([n.field_parameters() for n in self.nets], dev)
        self._install_grad_sinks()
        self.m = [torch.zeros_like(n.flat_params) for n in self.nets]
        self.v = [torch.zeros_like(n.flat_params) for n in self.nets]
        o = opt.optim
        self.lr0, self.gamma = o.lr, _sched_gamma(o.lr, o.get("lr_end"), o.get("sched", {"type": "ExponentialLR"}), opt.max_iter, "optim.sched")
        n = len(self.nets)
        self.lrs, self.gammas, self.trainable = [(o.lr, o.get("lr_end"))] * n, [self.gamma] * n, [True] * n          # (checkpoint.py reads these)
        on_gpu = dev.type == "cuda"
        opt.nerf.ray_sampler = opt.nerf.get("ray_sampler") or ("feistel" if on_gpu else "randperm")
        opt.nerf.stratified_rng = opt.nerf.get("stratified_rng") or ("philox" if on_gpu else "torch")
        opt.nerf.density_noise_rng = opt.nerf.get("density_noise_rng") or ("philox" if on_gpu else "torch")
        self._captured = None
        self.fused = None
        if fused_step:
            why = FusedStep.unsupported(self)
            if why is None:
                self.fused = FusedStep(self)
            elif fused_step is True:
                raise NiwError(f"fused_step=True: niw_train_step does not cover this configuration ({why})")
        self.fused_fallback_reason = None if self.fused is not None else (FusedStep.unsupported(self) if fused_step else "fused_step=False")

    def _install_grad_sinks(self):
        for i, net in enumerate(self.nets):
            net.grad_sink = self.bucket.segment(i)
        self.bucket.sunk = set(range(len(self.nets)))

    def fine_ready_event(self):
        return None

    def train_iteration(self, var, replay=True):
        opt, g, it = self.opt, self.graph, self.it
        if self.fused is not None:
            why = FusedStep.unsupported(self)
            if why is not None:
                raise NiwError(f"train_iteration: the configuration changed after the trainer was built and niw_train_step does not cover it any more ({why})")
            loss = self.fused.run(var, it)
        else:
            B = len(var.idx)
            n = opt.nerf.rand_rays // B
            self._install_grad_sinks()
            g._depth_draw, g._depth_call_in_iter = it + 1, 0
            set_noise_keys(opt, self.nets, it + 1)
            var.ray_idx = ops.draw_ray_idx(opt.H * opt.W, n, int(getattr(opt, "seed", 0) or 0), it + 1, opt.device)
            var.update(g.render(opt, g.get_pose(opt, var, mode="train"), intr=var.intr, ray_idx=var.ray_idx, mode="train"))
            loss = g.compute_loss(opt, var, mode="train")
            keys = [k for k in loss if opt.loss_weight[k] is not None]
            total = None
            with torch.no_grad():
                for k in keys:
                    w = 10 ** float(opt.loss_weight[k])
                    total = loss[k] * w if total is None else torch.add(total, loss[k], alpha=w)
            torch.autograd.backward([loss[k] for k in keys], [torch.full_like(loss[k], 10 ** float(opt.loss_weight[k])) for k in keys])
            loss.update(all=total)
        lr = self.lr0 * self.gamma ** it
        ops.adam_step_multi([(n_.flat_params, self.bucket.segment(i), self.m[i], self.v[i], lr, it + 1) for i, n_ in enumerate(self.nets)])
        self.it = it + 1
        return loss

    def _flats(self):
        return [n.flat_params for n in self.nets]


def synthetic_scene(opt, n_views, seed=0):
    """SURVEY section 8(d) synthetic inputs: uniform random images, pin-hole intrinsics
    fx = fy = 0.8 W, principal point at the image centre, identity initial poses."""
    gen = torch.Generator().manual_seed(seed)
    image = torch.rand(n_views, 3, opt.H, opt.W, generator=gen)
    intr = torch.tensor([[0.8 * opt.W, 0, opt.W / 2], [0, 0.8 * opt.W, opt.H / 2], [0, 0, 1]], dtype=torch.float32).repeat(n_views, 1, 1)
    pose = torch.eye(3, 4).repeat(n_views, 1, 1)
    return edict(idx=torch.arange(n_views), image=image.to(opt.device), intr=intr.to(opt.device), pose=pose.to(opt.device))


def synthetic_dtu_scene(opt, n_views, seed=0):
    """DTU-shaped synthetic inputs (cfg 5): cameras on a ring of radius 3 looking at the origin, ground-truth world-to-camera
    poses, metric depth range [1.2, 5.2] per view (data/dtu.py:110-111), and initial poses = se(3) noise (sigma = pose.noise)
    composed with the ground truth (barf_inn_dtu.py:38-45).  -> (var, initial_poses_w2c)"""
    from . import camera
    var = synthetic_scene(opt, n_views, seed)
    gen = torch.Generator().manual_seed(seed + 1)
    poses = []
    for i in range(n_views):
        a = 2 * math.pi * i / max(n_views, 1) * 0.25            # a quarter arc, as neighbouring DTU views
        eye = torch.tensor([3 * math.sin(a), 0.0, -3 * math.cos(a)])
        z = -eye / eye.norm()
        x = torch.linalg.cross(torch.tensor([0.0, 1.0, 0.0]), z)
        x = x / x.norm()
        y = torch.linalg.cross(z, x)
        R = torch.stack([x, y, z])                              # rows: camera axes in world coordinates (world -> camera)
        poses.append(torch.cat([R, (-R @ eye)[:, None]], dim=1))
    gt = torch.stack(poses).to(opt.device)
    noise = camera.lie.se3_to_SE3(torch.randn(n_views, 6, generator=gen).to(opt.device) * float(opt.pose.noise))
    init = camera.pose.compose([noise, gt])[:, :3]
    var.pose = gt
    var.depth_range = torch.tensor([[1.2, 5.2]] * n_views, device=opt.device)
    return var, init
