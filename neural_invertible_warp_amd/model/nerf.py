"""Mirror of the reference's model/nerf.py classes that sit on the render path: `NeRF`
(nerf.py:367-483) and `Graph` (nerf.py:243-365), same method names, argument meaning and
state-dict keys; the arithmetic is delegated to libniw_hip.so through ..ops.
"""
import math

import numpy as np
import torch

from .. import camera, ops
from .._lib import NiwError
from ..util import edict

_SUPPORTED_ARCH = dict(layers_feat=[None, 256, 256, 256, 256, 256, 256, 256, 256], layers_rgb=[None, 128, 3], skip=[4])


def _slice_rays(opt):
    """Rays per slice of a full-image render.  The reference slices by `nerf.rand_rays` (nerf.py:325) to bound its memory;
    `nerf.eval_slice_rays` overrides it (the HIP path takes up to 2^24 samples per launch: a 300x400 image in one go)."""
    n = opt.nerf.get("eval_slice_rays") or opt.nerf.rand_rays
    s = opt.nerf.sample_intvs + (opt.nerf.sample_intvs_fine or 0 if opt.nerf.fine_sampling else 0)
    return max(1, min(int(n), ((1 << 24) - 128) // max(s, 1)))


def _layer_table(arch):
    """-> [(stack, fan_in, fan_out, head)] in state-dict order.  Feature stack: the encoded point (3 + 6 L_3D) enters layer 0 and
    again every layer listed in `arch.skip`; the last feature layer emits one extra row, the raw density.  Colour stack: features
    and the encoded view direction (3 + 6 L_view) in, rgb out.  `head` names the initialisation of the two output layers."""
    d_point, d_view = 3 + 6 * arch.posenc.L_3D, 3 + 6 * arch.posenc.L_view
    rows = []
    widths = list(arch.layers_feat[1:])
    for depth, width in enumerate(widths):
        fan_in = (d_point if depth == 0 else widths[depth - 1]) + (d_point if depth in arch.skip else 0)
        closing = depth == len(widths) - 1
        rows.append(("mlp_feat", fan_in, width + (1 if closing else 0), "first" if closing else None))
    widths_rgb = list(arch.layers_rgb[1:])
    for depth, width in enumerate(widths_rgb):
        fan_in = widths[-1] + d_view if depth == 0 else widths_rgb[depth - 1]
        rows.append(("mlp_rgb", fan_in, width, "all" if depth == len(widths_rgb) - 1 else None))
    return rows


def _glorot_uniform_(weight, gain):
    """U(-a, a) with a = gain * sqrt(3) * sqrt(2 / (fan_in + fan_out)), evaluated like torch.nn.init.xavier_uniform_ so that the same
    generator state yields the same numbers"""
    fan_out, fan_in = weight.shape
    bound = 3.0 ** 0.5 * (gain * (2.0 / float(fan_in + fan_out)) ** 0.5)
    return weight.uniform_(-bound, bound)


class _BaseGraph(torch.nn.Module):
    """reference model/base.py:191-211"""

    def __init__(self, opt):
        super().__init__()

    def L1_loss(self, pred, label=0):
        return (pred.contiguous() - label).abs().mean()

    def MSE_loss(self, pred, label=0):
        return ((pred.contiguous() - label) ** 2).mean()


class NeRF(torch.nn.Module):

    def __init__(self, opt):
        super().__init__()
        self.define_network(opt)

    # ------------------------------------------------------------------ parameters
    def define_network(self, opt):
        """The two layer stacks of reference nerf.py:373-402 (`mlp_feat.N`, `mlp_rgb.N`: the state-dict keys), as views of ONE
        flat parameter buffer laid out in state-dict order -- the buffer the kernels, the fused Adam and the gradient bucket
        address.  The kernels are built for the one architecture every reference config uses; anything else is refused loudly."""
        arch = opt.arch
        if (list(arch.layers_feat) != _SUPPORTED_ARCH["layers_feat"] or list(arch.layers_rgb) != _SUPPORTED_ARCH["layers_rgb"]
                or list(arch.skip) != _SUPPORTED_ARCH["skip"] or not arch.posenc or arch.posenc.L_3D != ops.L3D
                or arch.posenc.L_view != ops.LVIEW or not opt.nerf.view_dep):
            raise NiwError("NeRF: only the 8x256 / skip[4] / rgb[128,3] / L_3D=10 / L_view=4 view-dependent architecture "
                           "of the reference configs is built into libniw_hip.so")
        if arch.density_activ not in ops.ACT:
            raise NiwError(f"NeRF: density activation {arch.density_activ!r} not supported (relu, softplus)")
        flat = torch.zeros(ops.NERF_PARAM_FLOATS, device=torch.device(opt.device), dtype=torch.float32)
        self.mlp_feat, self.mlp_rgb = torch.nn.ModuleList(), torch.nn.ModuleList()
        stacks = dict(mlp_feat=self.mlp_feat, mlp_rgb=self.mlp_rgb)
        cursor = 0
        for stack, fan_in, fan_out, head in _layer_table(arch):
            layer = torch.nn.Module()
            layer.in_features, layer.out_features = fan_in, fan_out
            layer.weight = torch.nn.Parameter(flat[cursor:cursor + fan_in * fan_out].view(fan_out, fan_in))
            layer.bias = torch.nn.Parameter(flat[cursor + fan_in * fan_out:cursor + (fan_in + 1) * fan_out])
            cursor += (fan_in + 1) * fan_out
            if arch.tf_init:
                self.tensorflow_init_weights(opt, layer, out=head)
            else:                                                   # nn.Linear's own default
                torch.nn.init.kaiming_uniform_(layer.weight, a=5 ** 0.5)
                torch.nn.init.uniform_(layer.bias, -fan_in ** -0.5, fan_in ** -0.5)
            stacks[stack].append(layer)
        assert cursor == ops.NERF_PARAM_FLOATS
        self.total_param = sum(l.weight.numel() for l in self.field_layers())
        # arithmetic of the field MLP: "fp32" (exact, default) or one of the opt-in fast modes of include/niw.h, chosen by
        # `opt.arch.precision` (not a reference key) or set_precision()
        self._state = ops.FieldState(flat, precision=opt.arch.get("precision") or "fp32")
        self.progress_host = None
        self.band_dev = None            # device tensor [14] of c2f band weights read by the kernel at run time (engine.StepConstants)
        self.grad_sink = None           # flat buffer that receives this network's parameter gradients instead of .grad (ops.field_mlp)

    def tensorflow_init_weights(self, opt, linear, out=None):
        """Glorot-uniform weights and zero biases as the TensorFlow NeRF code initialises them (reference nerf.py:404-414): hidden
        layers with the ReLU gain; the colour output (`out="all"`) with gain 1; the layer that emits density and features
        (`out="first"`) as two independent blocks -- the density row with gain 1, the feature rows with the ReLU gain."""
        blocks = {"all": [(slice(None), 1.0)], "first": [(slice(0, 1), 1.0), (slice(1, None), 2.0 ** 0.5)]}.get(out, [(slice(None), 2.0 ** 0.5)])
        with torch.no_grad():
            for rows, gain in blocks:
                _glorot_uniform_(linear.weight[rows], gain)
            linear.bias.zero_()

    def field_layers(self):
        return list(self.mlp_feat) + list(self.mlp_rgb)

    def field_parameters(self):
        """The 20 weight/bias Parameters in state-dict order (views of one flat buffer)."""
        return [p for layer in self.field_layers() for p in (layer.weight, layer.bias)]

    @property
    def flat_params(self):
        self._ensure_flat()
        return self._state.flat

    def _ensure_flat(self):
        """Parameters must alias the flat buffer the kernels read; re-flatten after .to()/.cuda()
        or any external re-assignment of parameter storage."""
        ps = self.field_parameters()
        flat = self._state.flat
        off, ok = 0, True
        for p in ps:
            if p.data_ptr() != flat.data_ptr() + 4 * off or p.device != flat.device:
                ok = False
                break
            off += p.numel()
        if ok:
            return
        new = torch.cat([p.detach().reshape(-1).float() for p in ps]).contiguous()
        off = 0
        for p in ps:
            p.data = new[off:off + p.numel()].view(p.shape)
            off += p.numel()
        self._state = ops.FieldState(new, precision=self._state.precision)

    def set_precision(self, precision):
        """"fp32" | "bf16x3" | "bf16" (ops.PREC): the arithmetic of this network's forward (and, where built, backward) kernels"""
        self._state.set_precision(precision)
        return self

    # ------------------------------------------------------------------ encoding weights
    def band_weights(self, opt, L):
        """BARF coarse-to-fine band mask (model/barf_inn_llff.py:427-442); all ones here."""
        return [1.0] * L

    def positional_encoding(self, opt, input, L):
        """reference nerf.py:476-483.  Interface parity only: the render path computes the
        encoding inside niw_mlp_fwd in MFMA slot order and never materialises it."""
        shape = input.shape
        freq = 2 ** torch.arange(L, dtype=torch.float32, device=input.device) * np.pi
        spectrum = input[..., None] * freq
        enc = torch.stack([spectrum.sin(), spectrum.cos()], dim=-2)
        w = torch.tensor(self.band_weights(opt, L), dtype=torch.float32, device=input.device)
        return (enc * w).view(*shape[:-1], -1)

    # ------------------------------------------------------------------ evaluation
    def _run(self, opt, center, ray, depth, mode):
        self._ensure_flat()
        noise = None
        if opt.nerf.density_noise_reg and mode == "train":
            key = getattr(self, "noise_key", None)
            if opt.nerf.get("density_noise_rng") == "philox" and key is not None:
                # the engine's draw (niw_normal_rng): a pure function of (stream key of this network, number of the iteration's pixel
                # draw, sample index) -- what niw_train_step draws for the same iteration; `noise_key` = (seed, draw, draw_dev) is set by
                # the trainer before every iteration
                noise = ops.normal_rng(key[0], key[1], depth.numel(), opt.nerf.density_noise_reg, depth.device, draw_dev=key[2]).view(depth.shape)
            else:
                noise = torch.randn(depth.shape, device=depth.device) * opt.nerf.density_noise_reg    # nerf.py:428-429
        return ops.field_mlp(self._state, self.field_parameters(), center, ray, depth,
                             self.band_weights(opt, ops.L3D), self.band_weights(opt, ops.LVIEW), opt.arch.density_activ, noise,
                             band_dev=self.band_dev, grad_sink=self.grad_sink if mode == "train" else None)

    def forward(self, opt, points_3D, ray_unit=None, mode=None):
        """reference nerf.py:416-447: points [...,3], unit view directions [...,3] -> rgb [...,3],
        density [...].  Evaluated by the same fused kernel as forward_samples with one sample per
        ray at depth 0 (x = c + 0 * v is exact)."""
        assert ray_unit is not None
        shape = points_3D.shape[:-1]
        pts = points_3D.reshape(-1, 3)
        dirs = ray_unit.expand_as(points_3D).reshape(-1, 3)
        rgb, density = self._run(opt, pts, dirs, torch.zeros(pts.shape[0], 1, device=pts.device), mode)
        return rgb.view(*shape, 3), density.view(*shape)

    def forward_samples(self, opt, center, ray, depth_samples, mode=None):
        """reference nerf.py:449-456: center, ray [B,R,3], depth_samples [B,R,S,1] ->
        rgb [B,R,S,3], density [B,R,S]."""
        B, R, S = depth_samples.shape[:3]
        rgb, density = self._run(opt, center.reshape(-1, 3), ray.reshape(-1, 3), depth_samples.reshape(B * R, S), mode)
        return rgb.view(B, R, S, 3), density.view(B, R, S)

    def composite(self, opt, ray, rgb_samples, density_samples, depth_samples):
        """reference nerf.py:458-474 -> rgb [B,R,3], depth [B,R,1], opacity [B,R,1], prob [B,R,S,1]."""
        B, R, S = density_samples.shape
        bg = opt.data.bgcolor if opt.nerf.setbg_opaque else None
        rgb, depth, opacity, prob = ops.composite(ray.reshape(-1, 3), rgb_samples.reshape(B * R, S, 3),
                                                  density_samples.reshape(B * R, S), depth_samples.reshape(B * R, S), bg)
        return rgb.view(B, R, 3), depth.view(B, R, 1), opacity.view(B, R, 1), prob.view(B, R, prob.shape[1], 1)


class Graph(_BaseGraph):

    def __init__(self, opt):
        super().__init__(opt)
        self.nerf = NeRF(opt)
        if opt.nerf.fine_sampling:
            self.nerf_fine = NeRF(opt)

    SAMPLED_MODES = ("train", "test-optim")        # modes that optimise on a random pixel subset (reference nerf.py:254)

    def forward(self, opt, var, mode=None):
        """reference nerf.py:251-274: a random pixel subset when optimising, whole images otherwise (in slices of `nerf.rand_rays`
        pixels whenever that bound is set).  The dead "render_train" branch of the reference is not carried over."""
        pose = self.get_pose(opt, var, mode=mode)
        n_rays = opt.nerf.rand_rays
        if n_rays and mode in self.SAMPLED_MODES:
            var.ray_idx = torch.randperm(opt.H * opt.W, device=opt.device)[:n_rays // len(var.idx)]
            out = self.render(opt, pose, intr=var.intr, ray_idx=var.ray_idx, mode=mode)
        else:
            whole_image = self.render_by_slices if n_rays else self.render
            out = whole_image(opt, pose, intr=var.intr, mode=mode)
        var.update(out)
        return var

    def compute_loss(self, opt, var, mode=None):
        """reference nerf.py:276-288"""
        loss = edict()
        ray_idx = var.ray_idx if (opt.nerf.rand_rays and mode in ["train", "test-optim"]) else None
        n_norm = getattr(opt, "loss_norm_elements", None)
        share = var.get("ray_share") if ray_idx is not None else None          # a rank's contiguous share of the B x R rays (..parallel)
        if opt.loss_weight.render is not None:
            loss.render = ops.mse_gather(var.rgb, var.image, ray_idx, n_norm, share=share)
        if opt.loss_weight.render_fine is not None:
            assert opt.nerf.fine_sampling
            loss.render_fine = ops.mse_gather(var.rgb_fine, var.image, ray_idx, n_norm, share=share)
        return loss

    def get_pose(self, opt, var, mode=None):
        return var.pose

    def _render_rays(self, opt, center, ray, mode=None, depth_range=None):
        """Shared tail of render / render_local (reference nerf.py:304-319)."""
        batch_size = ray.shape[0]
        depth_samples = self.sample_depth(opt, batch_size, num_rays=ray.shape[1]) if depth_range is None else \
            self.sample_depth(opt, batch_size, num_rays=ray.shape[1], depth_range=depth_range)
        rgb_samples, density_samples = self.nerf.forward_samples(opt, center, ray, depth_samples, mode=mode)
        rgb, depth, opacity, prob = self.nerf.composite(opt, ray, rgb_samples, density_samples, depth_samples)
        ret = edict(rgb=rgb, depth=depth, opacity=opacity)
        if opt.nerf.fine_sampling:
            with torch.no_grad():
                B, R, S = prob.shape[:3]
                _, merged = ops.sample_pdf_merge(prob.reshape(B * R, S), depth_samples.reshape(B * R, S),
                                                 opt.nerf.sample_intvs_fine, opt.nerf.depth.range)
                depth_samples = merged.view(B, R, -1, 1)
            rgb_samples, density_samples = self.nerf_fine.forward_samples(opt, center, ray, depth_samples, mode=mode)
            rgb_fine, depth_fine, opacity_fine, _ = self.nerf_fine.composite(opt, ray, rgb_samples, density_samples, depth_samples)
            ret.update(rgb_fine=rgb_fine, depth_fine=depth_fine, opacity_fine=opacity_fine)
        return ret

    def render(self, opt, pose, intr=None, ray_idx=None, mode=None):
        """reference nerf.py:293-319: rays of the poses (all pixels, or the pixels `ray_idx`) -> `_render_rays`.  The reference's
        retry loop around NaN rays (:296) guarded a CUDA bug at the price of a host sync per call; the HIP ray generator is
        deterministic, so it has no counterpart here."""
        return self._render_pixels(opt, pose, intr, mode, ray_idx=ray_idx)

    def _render_pixels(self, opt, pose, intr, mode, ray_idx=None, pixel_range=None, depth_range=None):
        center, ray = camera.get_center_and_ray(opt, pose, intr=intr, ray_idx=ray_idx, pixel_range=pixel_range)
        if opt.camera.ndc:
            center, ray = camera.convert_NDC(opt, center, ray, intr=intr)
        return self._render_rays(opt, center, ray, mode=mode, depth_range=depth_range)

    def _hold_weights(self):
        """context: pack the networks' weights once for a loop of gradient-free renders (ops.FieldState.hold)"""
        import contextlib
        stack = contextlib.ExitStack()
        if not torch.is_grad_enabled():
            for net in (self.nerf, getattr(self, "nerf_fine", None)):
                if net is not None:
                    net._ensure_flat()
                    stack.enter_context(net._state.hold())
        return stack

    def _sweep_image(self, opt, piece):
        """A full H x W render as a sequence of pixel RANGES (the reference bounds its memory the same way, by slices of
        `nerf.rand_rays` pixels: nerf.py:321-332): `piece(first, count)` renders pixels first .. first+count-1 of every view and its
        outputs land in pre-allocated [B, H*W, k] images.  No index tensors, no list-and-concatenate; weights are packed once."""
        total, step = opt.H * opt.W, _slice_rays(opt)
        image = None
        with self._hold_weights():
            for first in range(0, total, step):
                count = min(step, total - first)
                part = piece(first, count)
                if image is None:
                    image = edict({k: v.new_empty(v.shape[0], total, *v.shape[2:]) for k, v in part.items()})
                for k, v in part.items():
                    image[k][:, first:first + count] = v
        return image

    def render_by_slices(self, opt, pose, intr=None, mode=None):
        """reference nerf.py:321-332 -> edict of [B, H*W, k] maps"""
        if not torch.is_grad_enabled():
            return self._render_image(opt, pose, intr)
        return self._sweep_image(opt, lambda first, count: self._render_pixels(opt, pose, intr, mode, pixel_range=(first, count)))

    FUSED_SAMPLES = 1 << 26         # samples per niw_render_fwd call (its workspace: 24 B per sample, 1.6 GB at this size)

    def _render_image(self, opt, pose, intr, depth_range=None):
        """The gradient-free full-image render (val / eval / novel views) as library calls over pixel ranges (niw_render_fwd:
        rays -> NDC -> depths -> field -> compositing [-> fine pass] in one call), normally ONE for the whole image.  Same
        numbers as the stage-by-stage path, which stays in use wherever a gradient may be asked for."""
        B, total, S = pose.shape[0], opt.H * opt.W, opt.nerf.sample_intvs
        Sf = opt.nerf.sample_intvs_fine if opt.nerf.fine_sampling else 0
        step = max(1, min(total, self.FUSED_SAMPLES // (B * (S + (S + Sf if Sf else 0)))))
        fine = getattr(self, "nerf_fine", None) if Sf else None
        if fine is not None and fine._state.precision != self.nerf._state.precision:
            # niw_render_fwd takes ONE precision for both packed images of the call
            raise NiwError(f"render: nerf runs in {self.nerf._state.precision!r} and nerf_fine in {fine._state.precision!r}; "
                           "set_precision() both networks alike")
        image = None
        with self._hold_weights():
            for first in range(0, total, step):
                count = min(step, total - first)
                u = torch.rand(B * count, S, device=opt.device) if opt.nerf.sample_stratified else None
                part = ops.render_fwd(
                    intr, pose, opt.H, opt.W, (first, count), S, opt.nerf.depth.range if depth_range is None else depth_range,
                    opt.nerf.depth.param == "inverse", self.nerf._state.packed(), self.nerf.band_weights(opt, ops.L3D),
                    self.nerf.band_weights(opt, ops.LVIEW), opt.arch.density_activ, u=u, ndc_near=1.0 if opt.camera.ndc else None, n_fine=Sf,
                    packed_fine=None if fine is None else fine._state.packed(), pdf_range=opt.nerf.depth.range,
                    bg=opt.data.bgcolor if opt.nerf.setbg_opaque else None, band_dev=self.nerf.band_dev,
                    bands_fine=None if fine is None else (fine.band_weights(opt, ops.L3D), fine.band_weights(opt, ops.LVIEW), fine.band_dev),
                    precision=self.nerf._state.precision)
                if count == total:
                    return edict(part)
                if image is None:
                    image = edict({k: v.new_empty(B, total, v.shape[2]) for k, v in part.items()})
                for k, v in part.items():
                    image[k][:, first:first + count] = v
        return image

    def sample_depth(self, opt, batch_size, num_rays=None, depth_range=None):
        """reference nerf.py:334-344 -> [B,R,S,1]; the stratified draw is torch.rand as in the
        reference (nerf.py:337), the arithmetic is niw_sample_stratified."""
        rng = opt.nerf.depth.range if depth_range is None else depth_range
        num_rays = num_rays or opt.H * opt.W
        S = opt.nerf.sample_intvs
        if opt.nerf.sample_stratified and opt.nerf.get("stratified_rng") == "philox":
            # the draw happens inside the kernel (what the engine selects).  A stream is a pure function of (key, draw): the draw is the
            # number of the pixel draw of the current training iteration (`_depth_draw`, set by draw_ray_idx; replayable from a
            # captured graph through `draw_dev`), the key folds in opt.seed, the rank (ranks render different rays) and the number of
            # this sample_depth call SINCE that pixel draw -- so the slices of a full-image sweep and the validation renders between
            # two train steps each get a stream of their own (round 3: they all reused the last train step's).  Before the first
            # pixel draw the running count of calls numbers the draw.
            self._depth_calls = getattr(self, "_depth_calls", 0) + 1
            call = getattr(self, "_depth_call_in_iter", 0)
            self._depth_call_in_iter = call + 1
            rank = (getattr(opt, "ray_shard", None) or (0, 1))[0]
            stream_seed = (int(getattr(opt, "seed", 0) or 0) * 0x9E3779B97F4A7C15 + 0x5D1F + rank * 0xD1B54A32D192ED03
                           + call * 0x632BE59BD9B4E019) & (2 ** 64 - 1)
            d = ops.sample_stratified_rng(stream_seed, getattr(self, "_depth_draw", self._depth_calls), batch_size * num_rays, S, rng,
                                          opt.nerf.depth.param, opt.device, draw_dev=getattr(self, "draw_dev", None))
            return d.view(batch_size, num_rays, S, 1)
        u = torch.rand(batch_size, num_rays, S, 1, device=opt.device) if opt.nerf.sample_stratified else None
        d = ops.sample_stratified(None if u is None else u.view(-1, S), batch_size * num_rays, S, rng, opt.nerf.depth.param, opt.device)
        return d.view(batch_size, num_rays, S, 1)

    def sample_depth_from_pdf(self, opt, pdf):
        """reference nerf.py:346-365: pdf [B,R,S] -> [B,R,Sf,1]"""
        B, R, S = pdf.shape
        dummy = torch.zeros(B * R, S, device=pdf.device)
        fine, _ = ops.sample_pdf_merge(pdf.reshape(B * R, S), dummy, opt.nerf.sample_intvs_fine, opt.nerf.depth.range)
        return fine.view(B, R, -1, 1)


class Model:
    """The reference's engine interface for the vanilla model (model/nerf.py:20-160, driven by train.py:21-32), ground-truth
    poses.  With Adam on a GPU the iterations run through engine.NeRFTrainer (one niw_train_step call + one fused Adam launch, the
    reference's ExponentialLR; round 6); otherwise torch.optim with one param group per network and ExponentialLR exactly as the
    reference builds them (the parameters are ordinary nn.Parameters, views of the flat buffers the kernels read).  Checkpoints are
    the reference's model.ckpt dict either way (graph, optim, sched, epoch, iter).  Logging back-ends are outside the path."""

    trainer = None        # engine.NeRFTrainer when build_networks chose the engine (subclasses that build their own graph keep the torch path)

    def __init__(self, opt):
        import os
        self.opt = opt
        os.makedirs(opt.output_path, exist_ok=True)
        self.it = self.iter_start = 0

    def load_dataset(self, opt, eval_split="val"):
        """reference model/base.py:24-33 + nerf_inn_llff.py:22-32: both splits pre-loaded, tensors resident on the device"""
        from .. import data
        self.train_data, self.test_data = data.open_splits(opt, eval_split=eval_split)

    def build_networks(self, opt):
        """Adam on a GPU (the shipped options/nerf_llff_repr.yaml): the engine's trainer -- one niw_train_step call per iteration
        (warp_params = NULL) + one fused Adam launch over both networks (engine.NeRFTrainer, round 6; rounds 1-5 ran the mirror under
        torch.optim here).  Any other optimizer: the reference's own structure below, on the mirror."""
        self.trainer = None
        if opt.optim.get("algo", "Adam") == "Adam" and torch.device(opt.device).type == "cuda" and hasattr(self, "train_data"):
            from .. import engine
            self.trainer = engine.NeRFTrainer(opt, len(self.train_data), seed=opt.seed or 0)
            self.graph = self.trainer.graph
            return
        torch.manual_seed(opt.seed or 0)
        self.graph = Graph(opt).to(opt.device)

    def setup_optimizer(self, opt):
        if self.trainer is not None:
            return                                    # the trainer owns the flat Adam state and the ExponentialLR schedule
        optimizer = getattr(torch.optim, opt.optim.algo)
        self.optim = optimizer([dict(params=self.graph.nerf.parameters(), lr=opt.optim.lr)])
        if opt.nerf.fine_sampling:
            self.optim.add_param_group(dict(params=self.graph.nerf_fine.parameters(), lr=opt.optim.lr))
        gamma = (opt.optim.lr_end / opt.optim.lr) ** (1. / opt.max_iter) if opt.optim.get("lr_end") else 1.0
        self.sched = torch.optim.lr_scheduler.ExponentialLR(self.optim, gamma=gamma)

    def restore_checkpoint(self, opt):
        """reference util.py:124-145 as called from base.py:64-73: `--resume` (True = model.ckpt, a number = model/<n>.ckpt) restores
        networks, optimizer, scheduler and the counters; `--load=<file>` restores the networks only.  Children of the graph that the
        checkpoint does not mention keep their initialisation (partial checkpoints)."""
        from .. import checkpoint
        self.epoch_start = self.iter_start = 0
        if self.trainer is not None:
            ep = it = None
            if opt.resume:
                ep, it = checkpoint.restore_checkpoint(opt, self.trainer, resume=opt.resume)
            elif opt.load is not None:
                ep, it = checkpoint.restore_checkpoint(opt, self.trainer, load_name=opt.load)
            self.epoch_start, self.iter_start = ep or 0, it or 0
            return
        path = checkpoint.checkpoint_path(opt, resume=opt.resume) if opt.resume else opt.load
        if path is None:
            return
        state = torch.load(path, map_location=opt.device, weights_only=False)
        per_child = checkpoint.group_by_child(state["graph"])
        for name, module in self.graph.named_children():
            if name in per_child:
                module.load_state_dict(per_child[name])
        if opt.resume:
            for attr in ("optim", "sched"):
                getattr(self, attr).load_state_dict(state[attr])
            self.epoch_start, self.iter_start = state["epoch"] or 0, state["iter"] or 0

    def setup_visualizer(self, opt):
        pass

    def train(self, opt):
        self.graph.train()
        var = self.train_data.all
        self.it = self.iter_start
        if self.iter_start == 0:
            self.validate(opt, 0)
        if self.trainer is not None:
            self.trainer.it = self.it
        while self.it < opt.max_iter:
            loss = self.train_iteration(opt, var, None)
            if self.trainer is None:
                self.sched.step()
            if self.it % opt.freq.scalar == 0:
                print("[train it {}] {}".format(self.it, " ".join("{}={:.5f}".format(k, float(v.detach())) for k, v in loss.items())), flush=True)
            if self.it % opt.freq.val == 0:
                self.validate(opt, self.it)
            if self.it % opt.freq.ckpt == 0:
                self.save_checkpoint(opt, ep=None, it=self.it)
        return self

    def summarize_loss(self, opt, var, loss):
        total = 0.
        for key in loss:
            if opt.loss_weight[key] is not None:
                total = total + 10 ** float(opt.loss_weight[key]) * loss[key]
        loss.update(all=total)
        return loss

    def train_iteration(self, opt, var, loader=None):
        if self.trainer is not None:
            loss = self.trainer.train_iteration(edict(var))
            self.it = self.trainer.it
            return loss
        self.optim.zero_grad(set_to_none=True)
        var = self.graph.forward(opt, edict(var), mode="train")
        loss = self.summarize_loss(opt, var, self.graph.compute_loss(opt, var, mode="train"))
        loss.all.backward()
        self.optim.step()
        self.it += 1
        return loss

    @torch.no_grad()
    def validate(self, opt, ep=None):
        self.graph.eval()
        psnr = []
        allv = self.test_data.all
        for i in range(len(self.test_data)):
            var = self.graph.forward(opt, edict({k: v[i:i + 1] for k, v in allv.items()}), mode="val")
            rgb = var.rgb_fine if opt.nerf.fine_sampling else var.rgb
            rgb_map = rgb.view(-1, opt.H, opt.W, 3).permute(0, 3, 1, 2)
            psnr.append(-10 * self.graph.MSE_loss(rgb_map, var.image).log10().item())
        self.graph.train()
        out = edict(psnr=sum(psnr) / max(len(psnr), 1))
        print("[val it {}] PSNR {:.2f}".format(ep, out.psnr), flush=True)
        return out

    def save_checkpoint(self, opt, ep=0, it=0, latest=False):
        import os
        import shutil
        if self.trainer is not None:
            from .. import checkpoint
            return checkpoint.save_checkpoint(opt, self.trainer, ep=ep, it=it, latest=latest)
        os.makedirs("{0}/model".format(opt.output_path), exist_ok=True)
        ck = dict(epoch=ep, iter=it, graph=self.graph.state_dict(), optim=self.optim.state_dict(), sched=self.sched.state_dict())
        torch.save(ck, "{0}/model.ckpt".format(opt.output_path))
        if not latest:
            shutil.copy("{0}/model.ckpt".format(opt.output_path), "{0}/model/{1}.ckpt".format(opt.output_path, ep or it))
