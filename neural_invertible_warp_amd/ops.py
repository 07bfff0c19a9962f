"""Torch-facing wrappers of the C ABI: plain functions for the gradient-free stages and
`torch.autograd.Function`s for the differentiable ones.  PyTorch only provides device memory,
streams and the autograd tape here; every number is produced by libniw_hip.so.
"""
import ctypes
import contextlib
import math

import torch

from . import _lib

NERF_PARAM_FLOATS = 530052
SAVE_ROWS = 2346
GRAD_ROWS = 2336
L3D, LVIEW = 10, 4
TRAIN_LAUNCH_SAMPLES = (1 << 31) // (288 * 4)          # samples per differentiable field_mlp launch (see field_mlp)
ACT = {"relu": 0, "softplus": 1}
PREC = {"fp32": 0, "bf16x3": 1, "bf16": 2}         # enum niw_precision (include/niw.h); "fp32" = exact, the default everywhere
DX_PRECISIONS = {"fp32", "bf16x3", "bf16"}           # precisions whose dX chain / dW GEMMs exist.  A missing "bf16x3" pass would fall back to
DW_PRECISIONS = {"fp32", "bf16x3", "bf16"}           # the exact-fp32 kernel (same fp32 workspaces); "bf16" cannot (bf16 workspaces, see _backward_precision)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise _lib.NiwError(f"{name}: expected a float32 CUDA/HIP tensor, got {type(t).__name__} "
                            f"{getattr(t, 'dtype', None)} on {getattr(t, 'device', None)}")
    return t.contiguous()


class _Timing:
    """Optional per-launch timing with device events on the launch stream (bench.py turns it on).
    records[name] = list of (start_event, end_event, units)."""

    def __init__(self):
        self.enabled = False
        self.records = {}

    def reset(self):
        self.records = {}

    def summary(self):
        """-> {name: (launches, mean_ms, units_per_launch)}; call after a device synchronize."""
        out = {}
        for name, recs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _ in recs]
            out[name] = (len(ms), sum(ms) / len(ms), sum(u for _, _, u in recs) / len(recs))
        return out


TIMING = _Timing()


class timed:
    def __init__(self, name, units):
        self.name, self.units = name, units

    def __enter__(self):
        if TIMING.enabled:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()          # torch's current stream == the stream handed to the C ABI
        return self

    def __exit__(self, *exc):
        if TIMING.enabled:
            self.b.record()
            TIMING.records.setdefault(self.name, []).append((self.a, self.b, self.units))
        return False


def _farr(vals, n):
    vals = [float(v) for v in vals]
    assert len(vals) == n
    return (ctypes.c_float * n)(*vals)


# ------------------------------------------------------------------------------------------
# gradient-free stages
# ------------------------------------------------------------------------------------------

def raygen(intr, pose, ray_idx, H, W, mode, pixel_range=None):
    """mode 0: (center, grid) of camera.get_unwarped_center_and_ray; mode 1: (center, ray) of
    camera.get_center_and_ray.  Returns two [B,R,3] tensors.  Rays: `ray_idx` (int64 pixel ids), or the consecutive pixels
    `pixel_range` = (first, count) (a slice of a full-image render: no index tensor), or the whole image."""
    intr = _f32(intr, "intr")
    B = intr.shape[0]
    pose = None if pose is None else _f32(pose, "pose")
    first = 0
    if ray_idx is not None:
        ray_idx = ray_idx.to(device=intr.device, dtype=torch.int64).contiguous()
        R = ray_idx.numel()
    elif pixel_range is not None:
        first, R = int(pixel_range[0]), int(pixel_range[1])
    else:
        R = H * W
    a = torch.empty(B, R, 3, device=intr.device, dtype=torch.float32)
    b = torch.empty_like(a)
    _lib.call("niw_raygen", _p(intr), _p(pose), _p(ray_idx), first, B, R, H, W, mode, _p(a), _p(b), _stream())
    return a, b


def draw_ray_idx(n_pixels, n, seed, draw, device, first=0, stride=1, draw_dev=None):
    """`torch.randperm(n_pixels)[:n]` of the reference (nerf_inn_llff.py:510) as one sort-free launch: element i is
    P(first + i * stride) of a keyed pseudo-random permutation P of [0, n_pixels) (include/niw.h niw_draw_ray_idx).
    `draw` numbers the draw (the training iteration); `draw_dev` (uint64 device word) overrides it at run time."""
    out = torch.empty(n, device=device, dtype=torch.int64)
    _lib.call("niw_draw_ray_idx", int(n_pixels), int(n), int(seed) & (2 ** 64 - 1), int(draw) & (2 ** 64 - 1), _p(draw_dev), int(first), int(stride),
              _p(out), _stream())
    return out


def _convert_ndc_raw(center, ray, intr, near):
    oc, orr = torch.empty_like(center), torch.empty_like(ray)
    _lib.call("niw_convert_ndc", _p(center), _p(ray), _p(intr), center.shape[0], center.shape[1], float(near),
              _p(oc), _p(orr), _stream())
    return oc, orr


class _ConvertNDC(torch.autograd.Function):
    """niw_convert_ndc with niw_convert_ndc_bwd as its reverse pass (round 6): the gradient of warped rays through the NDC
    re-parametrisation (reference camera.py:523-540 under autograd) as one launch, the same launch niw_train_step makes"""

    @staticmethod
    def forward(ctx, center, ray, intr, near):
        ctx.save_for_backward(center, ray, intr)
        ctx.near = float(near)
        return _convert_ndc_raw(center, ray, intr, near)

    @staticmethod
    def backward(ctx, g_center, g_ray):
        center, ray, intr = ctx.saved_tensors
        gc = None if g_center is None else _f32(g_center, "d_center_ndc")
        gr = None if g_ray is None else _f32(g_ray, "d_ray_ndc")
        d_center, d_ray = torch.empty_like(center), torch.empty_like(ray)
        _lib.call("niw_convert_ndc_bwd", _p(center), _p(ray), _p(intr), center.shape[0], center.shape[1], ctx.near, _p(gc), _p(gr), _p(d_center), _p(d_ray),
                  _stream())
        return d_center, d_ray, None, None


def convert_ndc(center, ray, intr, near=1.0):
    center, ray, intr = _f32(center, "center"), _f32(ray, "ray"), _f32(intr, "intr")
    if torch.is_grad_enabled() and (center.requires_grad or ray.requires_grad):
        return _ConvertNDC.apply(center, ray, intr, float(near))
    return _convert_ndc_raw(center, ray, intr, near)


def sample_stratified(u, n_rays, S, depth_range, param, device):
    """Graph.sample_depth: u [n_rays,S] (or None for the 0.5 mid-points) -> depth [n_rays,S]."""
    if param not in ("metric", "inverse"):
        raise KeyError(param)
    if u is not None:
        u = _f32(u, "u")
    out = torch.empty(n_rays, S, device=device, dtype=torch.float32)
    _lib.call("niw_sample_stratified", _p(u), n_rays, S, float(depth_range[0]), float(depth_range[1]),
              1 if param == "inverse" else 0, _p(out), _stream())
    return out


def sample_stratified_rng(seed, draw, n_rays, S, depth_range, param, device, draw_dev=None, return_u=False):
    """Graph.sample_depth with the stratified draw made inside the kernel (niw_sample_stratified_rng: Philox keyed by `seed`,
    counter = (sample, `draw`); `draw_dev` = uint64 device word overriding `draw` at run time) -> depth [n_rays,S] (, u)."""
    if param not in ("metric", "inverse"):
        raise KeyError(param)
    out = torch.empty(n_rays, S, device=device, dtype=torch.float32)
    u = torch.empty(n_rays, S, device=device, dtype=torch.float32) if return_u else None
    _lib.call("niw_sample_stratified_rng", int(seed) & (2 ** 64 - 1), int(draw) & (2 ** 64 - 1), _p(draw_dev), n_rays, S,
              float(depth_range[0]), float(depth_range[1]), 1 if param == "inverse" else 0, _p(out), _p(u), _stream())
    return (out, u) if return_u else out


def normal_rng(seed, draw, n, scale, device, draw_dev=None):
    """niw_normal_rng: n values scale * N(0, 1) -- Box-Muller over the Philox stream keyed by `seed`, counter (i / 4, `draw`) -> [n].
    The density noise of the train-mode field forward (reference model/nerf.py:428-429) when the engine draws it on the device."""
    out = torch.empty(n, device=device, dtype=torch.float32)
    _lib.call("niw_normal_rng", int(seed) & (2 ** 64 - 1), int(draw) & (2 ** 64 - 1), _p(draw_dev), n, float(scale), _p(out), _stream())
    return out


_table_cache = {}


def _pdf_tables(S, Sf, depth_range, device):
    key = (S, Sf, float(depth_range[0]), float(depth_range[1]), str(device))
    if key not in _table_cache:
        g = torch.linspace(0, 1, Sf + 1)                       # nerf.py:352
        unif = 0.5 * (g[:-1] + g[1:])                          # nerf.py:353
        bins = torch.linspace(depth_range[0], depth_range[1], S + 1)   # nerf.py:356
        _table_cache[key] = (unif.to(device).contiguous(), bins.to(device).contiguous())
    return _table_cache[key]


def sample_pdf_merge(pdf, depth_coarse, Sf, depth_range):
    """Returns (depth_fine [N,Sf], depth_merged [N,S+Sf] ascending)."""
    pdf, depth_coarse = _f32(pdf, "pdf"), _f32(depth_coarse, "depth_coarse")
    N, S = pdf.shape
    unif, bins = _pdf_tables(S, Sf, depth_range, pdf.device)
    fine = torch.empty(N, Sf, device=pdf.device, dtype=torch.float32)
    merged = torch.empty(N, S + Sf, device=pdf.device, dtype=torch.float32)
    _lib.call("niw_sample_pdf_merge", _p(pdf), _p(depth_coarse), _p(unif), _p(bins), N, S, Sf, _p(fine), _p(merged), _stream())
    return fine, merged


def render_fwd(intr, pose, H, W, pixel_range, n_samples, depth_range, inverse_depth, packed, band3d, bandview, activ, u=None, ndc_near=None,
               n_fine=0, packed_fine=None, pdf_range=None, bg=None, band_dev=None, bands_fine=None, precision="fp32"):
    """Gradient-free render of the pixels `pixel_range` = (first, count) of every view as ONE library call (niw_render_fwd): rays,
    NDC (when `ndc_near` is given), stratified depths from `u` [B*count, S] (None: mid-points), field MLP, compositing and, when
    n_fine > 0, the inverse-CDF / merge / fine-network pass.  `packed` / `packed_fine`: FieldState.packed() images; `bands_fine` = (band3d, bandview, band_dev) of the fine
    network (None: all ones).
    -> dict(rgb [B,count,3], depth [B,count,1], opacity [B,count,1] [, rgb_fine, depth_fine, opacity_fine])"""
    intr, pose = _f32(intr, "intr"), _f32(pose, "pose")
    B, dev = intr.shape[0], intr.device
    first, count = int(pixel_range[0]), int(pixel_range[1])
    if u is not None:
        u = _f32(u, "u")
        if u.numel() != B * count * n_samples:
            raise _lib.NiwError(f"render_fwd: u has {u.numel()} elements for {B} x {count} rays x {n_samples} samples")
    d = _lib.RenderDesc(intr=intr.data_ptr(), pose=pose.data_ptr(), n_views=B, H=H, W=W, ndc=0 if ndc_near is None else 1, first_pixel=first,
                        n_pixels=count, ndc_near=0.0 if ndc_near is None else float(ndc_near), depth_min=float(depth_range[0]),
                        depth_max=float(depth_range[1]), inverse_depth=1 if inverse_depth else 0, n_samples=n_samples, n_fine=n_fine,
                        density_activ=ACT[activ], precision=PREC[precision], has_bg=0 if bg is None else 1, bg=0.0 if bg is None else float(bg),
                        u=None if u is None else u.data_ptr(), packed=packed.data_ptr(), band_dev=None if band_dev is None else band_dev.data_ptr())
    b3, bv = _farr(band3d, L3D), _farr(bandview, LVIEW)
    d.band_w3d, d.band_wview = ctypes.cast(b3, ctypes.POINTER(ctypes.c_float)), ctypes.cast(bv, ctypes.POINTER(ctypes.c_float))
    out = {k: torch.empty(B, count, c, device=dev, dtype=torch.float32) for k, c in (("rgb", 3), ("depth", 1), ("opacity", 1))}
    tables = None
    if n_fine:
        tables = _pdf_tables(n_samples, n_fine, depth_range if pdf_range is None else pdf_range, dev)
        d.unif, d.bins, d.packed_fine = tables[0].data_ptr(), tables[1].data_ptr(), packed_fine.data_ptr()
        if bands_fine is not None:
            f3, fv = _farr(bands_fine[0], L3D), _farr(bands_fine[1], LVIEW)
            d.band_w3d_fine, d.band_wview_fine = ctypes.cast(f3, ctypes.POINTER(ctypes.c_float)), ctypes.cast(fv, ctypes.POINTER(ctypes.c_float))
            d.band_dev_fine = None if bands_fine[2] is None else bands_fine[2].data_ptr()
        out.update({k + "_fine": torch.empty_like(v) for k, v in list(out.items())})
    ws = torch.empty(_lib.load().niw_render_fwd_workspace_floats(B, count, n_samples, n_fine), device=dev, dtype=torch.float32)
    with timed("render_fwd", B * count * (n_samples + (n_samples + n_fine if n_fine else 0))):
        _lib.call("niw_render_fwd", ctypes.byref(d), _p(ws), _p(out["rgb"]), _p(out["depth"]), _p(out["opacity"]), _p(out.get("rgb_fine")),
                  _p(out.get("depth_fine")), _p(out.get("opacity_fine")), _stream())
    return out


def adam_hyper(lr, step, beta1=0.9, beta2=0.999):
    """(lr / bias_correction1, sqrt(bias_correction2)) as torch.optim.Adam forms them: the two step-dependent scalars of the update"""
    return lr / (1.0 - beta1 ** step), math.sqrt(1.0 - beta2 ** step)


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, hyper_dev=None):
    """hyper_dev: device tensor [2] = adam_hyper(lr, step), read by the kernel at run time instead of lr / step (graph replays)"""
    _lib.call("niw_adam_step", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), float(lr), float(beta1),
              float(beta2), float(eps), int(step), _p(hyper_dev), _stream())


def adam_step_multi(groups, beta1=0.9, beta2=0.999, eps=1e-8, hyper_dev=None):
    """Adam over several flat buffers in ONE launch (niw_adam_step_multi).  groups: list of (param, grad, exp_avg, exp_avg_sq, lr, step)
    or None for a group that is not trained (it keeps its slot: hyper_dev [len(groups), 2] is indexed by position)."""
    arr = (_lib.AdamGroup * len(groups))()
    for k, g in enumerate(groups):
        if g is None:
            continue
        param, grad, m, v, lr, step = g
        arr[k] = _lib.AdamGroup(param=param.data_ptr(), grad=grad.data_ptr(), exp_avg=m.data_ptr(), exp_avg_sq=v.data_ptr(), n=param.numel(),
                                lr=float(lr), step=int(step))
    with timed("adam", sum(g[0].numel() for g in groups if g is not None)):
        _lib.call("niw_adam_step_multi", arr, len(groups), float(beta1), float(beta2), float(eps), _p(hyper_dev), _stream())


def pack_index(device):
    """gather table of the packed fp32 weight layout (architecture constant, built once per device)"""
    key = str(device)
    if key not in FieldState._index:
        idx = torch.empty(_lib.load().niw_mlp_packed_floats(), device=device, dtype=torch.int32)
        _lib.call("niw_mlp_pack_index", _p(idx), _stream())
        FieldState._index[key] = idx
    return FieldState._index[key]


# ------------------------------------------------------------------------------------------
# field MLP
# ------------------------------------------------------------------------------------------

class FieldState:
    """Flat parameter storage of one NeRF MLP + the packed-weight cache.  `precision` selects the arithmetic of the field MLP
    (PREC: "fp32" exact -- the default --, "bf16x3" / "bf16" the opt-in fast modes of include/niw.h)."""

    def __init__(self, flat, precision="fp32"):
        assert flat.numel() == NERF_PARAM_FLOATS
        self.flat = flat
        self._held = None
        self.set_precision(precision)

    def set_precision(self, precision):
        if precision not in PREC:
            raise _lib.NiwError(f"field MLP precision {precision!r}: choose from {sorted(PREC)}")
        self.precision = precision
        self._held = None

    @contextlib.contextmanager
    def hold(self):
        """Inside this context the weights are known not to change (the slice loop of a full-image render): pack once."""
        self._held = self._pack()
        try:
            yield self
        finally:
            self._held = None

    def packed(self):
        """The weight image of the state's precision class.  Re-packed on every call (outside hold()): the parameters are views of
        `flat` updated in place by any optimizer, which no version counter of `flat` observes; packing 2.4 M floats costs ~50 us
        next to a multi-millisecond MLP launch.  A fresh buffer is returned so that a pending backward keeps the weights its
        forward used."""
        return self._held if self._held is not None else self._pack()

    _index = {}          # device -> gather table of the packed layout (architecture constant, built once)

    def _pack(self):
        return self.packed_fp32() if self.precision == "fp32" else self._pack_fast()

    def packed_fp32(self):
        n = _lib.load().niw_mlp_packed_floats()
        packed = torch.empty(n, device=self.flat.device, dtype=torch.float32)
        _lib.call("niw_mlp_pack_weights_indexed", _p(self.flat), _p(pack_index(self.flat.device)), _p(packed), _stream())
        return packed

    def _pack_fast(self):
        """split-bf16 image (hi / mid planes in MFMA fragment order, niw_mlp_pack_weights_prec); one image serves bf16x3 and bf16"""
        n = _lib.load().niw_mlp_packed_bytes(PREC[self.precision])
        image = torch.empty(n // 4, device=self.flat.device, dtype=torch.float32)
        _lib.call("niw_mlp_pack_weights_prec", _p(self.flat), PREC[self.precision], _p(image), _stream())
        return image

    def _pack_decode(self):
        n = _lib.load().niw_mlp_packed_floats()
        packed = torch.empty(n, device=self.flat.device, dtype=torch.float32)
        _lib.call("niw_mlp_pack_weights", _p(self.flat), _p(packed), _stream())
        return packed


def _backward_precision(precision, built, what):
    if precision in built:
        return precision
    if precision == "bf16":
        raise _lib.NiwError(f"field MLP: the {what} of precision 'bf16' are not built, and its bf16 workspaces cannot be handed to the fp32 kernels")
    return "fp32"


class _FieldMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, band3d, bandview, band_dev, activ, noise, grad_mode, grad_sink, center, ray, depth, *params):
        center, ray, depth = _f32(center, "center"), _f32(ray, "ray"), _f32(depth, "depth_samples")
        n_rays, S = depth.shape
        if center.shape != (n_rays, 3) or ray.shape != (n_rays, 3):
            raise _lib.NiwError(f"forward_samples: center {tuple(center.shape)} / ray {tuple(ray.shape)} do not match depth {tuple(depth.shape)}")
        lib = _lib.load()
        dev = center.device
        packed = state.packed()
        rgb = torch.empty(n_rays, S, 3, device=dev, dtype=torch.float32)
        sigma = torch.empty(n_rays, S, device=dev, dtype=torch.float32)
        # training mode (activations saved) only when a gradient can actually be asked for: needs_input_grad reflects
        # requires_grad of the inputs even under torch.no_grad(), and grad mode itself is always off inside Function.forward,
        # so the caller's grad mode is passed in (without it every eval render ran the saving kernel: 15 % slower, 7 GB)
        need = grad_mode and any(ctx.needs_input_grad)
        mpad = lib.niw_mlp_padded_rows(n_rays, S)
        save = torch.empty(SAVE_ROWS * mpad, device=dev, dtype=torch.float32) if need else None
        b3, bv = _farr(band3d, L3D), _farr(bandview, LVIEW)
        if noise is not None:
            noise = _f32(noise, "noise")
        with timed("mlp_fwd_train" if need else "mlp_fwd", n_rays * S):
            _lib.call("niw_mlp_fwd", _p(packed), _p(center), _p(ray), _p(depth), _p(noise), n_rays, S,
                      b3, bv, _p(band_dev), ACT[activ], PREC[state.precision], _p(rgb), _p(sigma), _p(save), _stream())
        ctx.state, ctx.b3, ctx.bv, ctx.activ, ctx.mpad = state, b3, bv, activ, mpad
        # a backward pass of "bf16x3" that had no kernel of its own would run on the exact-fp32 kernels (that mode's saves are the
        # same fp32 workspace) with the fp32 image of the same weights; "bf16" keeps bf16 half-pitch quad rows in `save` /
        # `gradws` (niw_mlp_fast.hip kHalfWorkspace), which the fp32 kernels would read as garbage -- no fallback there
        ctx.dx_precision = _backward_precision(state.precision, DX_PRECISIONS, "dX chain")
        ctx.dw_precision = _backward_precision(state.precision, DW_PRECISIONS, "dW GEMMs")
        ctx.save_ws, ctx.grad_sink = save, grad_sink
        ctx.packed = packed if (not need or ctx.dx_precision == state.precision) else state.packed_fp32()
        ctx.set_materialize_grads(False)
        ctx.param_shapes = [p.shape for p in params]
        ctx.save_for_backward(center, ray, depth, rgb)
        ctx.mark_non_differentiable()
        return rgb, sigma

    @staticmethod
    def backward(ctx, d_rgb, d_sigma):
        center, ray, depth, rgb = ctx.saved_tensors
        n_rays, S = depth.shape
        dev = center.device
        lib = _lib.load()
        d_rgb = torch.zeros_like(rgb) if d_rgb is None else _f32(d_rgb, "d_rgb")
        d_sigma = torch.zeros(n_rays, S, device=dev) if d_sigma is None else _f32(d_sigma, "d_sigma")
        gradws = torch.empty(GRAD_ROWS * ctx.mpad, device=dev, dtype=torch.float32)
        partial = torch.empty(lib.niw_mlp_bwd_workspace_floats(n_rays, S), device=dev, dtype=torch.float32)
        # grad_sink: the caller's flat gradient buffer in state-dict order (the engine's all-reduce / Adam bucket).  The kernels
        # overwrite it in place and autograd receives no parameter gradients: no per-parameter accumulation copies, no gather.
        sink = ctx.grad_sink
        d_params = sink if sink is not None else torch.empty(NERF_PARAM_FLOATS, device=dev, dtype=torch.float32)
        ray_grad = ctx.needs_input_grad[8] or ctx.needs_input_grad[9]
        d_both = torch.empty(2, n_rays, 3, device=dev, dtype=torch.float32) if ray_grad else None      # overwritten (fixed-order per-ray sums)
        d_center, d_ray = (d_both[0], d_both[1]) if ray_grad else (None, None)
        with timed("mlp_bwd_dx", n_rays * S):
            _lib.call("niw_mlp_bwd_dx", _p(ctx.packed), _p(center), _p(ray), _p(depth), n_rays, S, ACT[ctx.activ], PREC[ctx.dx_precision], _p(rgb),
                      _p(d_rgb), _p(d_sigma), _p(ctx.save_ws), _p(gradws), _p(d_center), _p(d_ray), _stream())
        # (running this group on a second stream beside the rest of the backward was tried: once the seven 256 x 256
        # pieces became one 511-workgroup launch it fills the chip by itself and the overlap cost 40 %)
        with timed("mlp_bwd_dw", n_rays * S):
            _lib.call("niw_mlp_bwd_dw", _p(ctx.save_ws), _p(gradws), n_rays, S, PREC[ctx.dw_precision], _p(partial), _p(d_params), _stream())
        ctx.save_ws = None
        grads, off = [], 0
        for shp in ctx.param_shapes:
            n = math.prod(shp)
            grads.append(None if sink is not None else d_params[off:off + n].view(shp))
            off += n
        return (None, None, None, None, None, None, None, None, d_center, d_ray, None, *grads)


def field_mlp(state, params, center, ray, depth, band3d, bandview, activ, noise=None, band_dev=None, grad_sink=None):
    """NeRF.forward_samples on flattened rays: center, ray [N,3], depth [N,S] -> rgb [N,S,3], sigma [N,S].
    band_dev: device tensor [14] = {band3d, bandview}; when given the kernel reads the c2f weights from it at run time.
    One launch takes fewer than 2^24 padded samples (32-bit byte offsets into the workspaces, niw_mlp_device.h), 1.86 M when
    gradients are wanted (the dW GEMM reaches a 288-row operand through one 2 GiB descriptor); larger
    batches are split over the rays (autograd sums the parameter gradients of the pieces).
    grad_sink: flat float32 buffer [NERF_PARAM_FLOATS] that receives the parameter gradients of this call IN PLACE of the
    Parameters' .grad (engine.INNTrainer: a segment of its gradient bucket); one call per backward pass may write it."""
    n_rays, S = depth.shape
    # one launch: < 2^24 padded samples (32-bit byte offsets inside a 32-row window); with gradients < 2^31 / (288 * 4) = 1.86 M,
    # because the dW GEMM addresses a whole operand (up to 288 rows x samples x 4 bytes) through one 2 GiB buffer descriptor
    training = torch.is_grad_enabled() and bool(params)
    max_rays = ((TRAIN_LAUNCH_SAMPLES if training else (1 << 24)) - 256) // S
    if grad_sink is not None and (not torch.is_grad_enabled() or not params):
        grad_sink = None
    if grad_sink is not None and (grad_sink.numel() != NERF_PARAM_FLOATS or grad_sink.dtype != torch.float32 or not grad_sink.is_contiguous()):
        raise _lib.NiwError("field_mlp: grad_sink must be a contiguous float32 buffer of NERF_PARAM_FLOATS elements")
    if n_rays <= max_rays:
        return _FieldMLP.apply(state, band3d, bandview, band_dev, activ, noise, torch.is_grad_enabled(), grad_sink, center, ray, depth, *params)
    if grad_sink is not None:
        raise _lib.NiwError("field_mlp: a batch beyond one launch's sample limit (1.86 M with gradients, 2^24 without) is split into pieces "
                            "whose gradients autograd sums; that cannot be combined with grad_sink")
    rgb, sigma = [], []
    with (state.hold() if not params else contextlib.nullcontext()):
        for a in range(0, n_rays, max_rays):
            b = min(a + max_rays, n_rays)
            r, s_ = _FieldMLP.apply(state, band3d, bandview, band_dev, activ, None if noise is None else noise[a:b], torch.is_grad_enabled(), None, center[a:b], ray[a:b],
                                    depth[a:b], *params)
            rgb.append(r)
            sigma.append(s_)
    return torch.cat(rgb), torch.cat(sigma)


# ------------------------------------------------------------------------------------------
# compositing
# ------------------------------------------------------------------------------------------

class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ray, rgb_s, sigma_s, depth_s, bg):
        ray, rgb_s, sigma_s, depth_s = _f32(ray, "ray"), _f32(rgb_s, "rgb_samples"), _f32(sigma_s, "density_samples"), _f32(depth_s, "depth_samples")
        N, S = sigma_s.shape
        dev = ray.device
        rgb = torch.empty(N, 3, device=dev)
        depth = torch.empty(N, device=dev)
        opacity = torch.empty(N, device=dev)
        prob = torch.empty(N, S if S > 1 else 0, device=dev)      # S = 1: the reference's weights are EMPTY (nerf.py:461-462, see niw.h)
        with timed("composite_fwd", N * S):
            _lib.call("niw_composite_fwd", _p(ray), _p(rgb_s), _p(sigma_s), _p(depth_s), N, S, 0 if bg is None else 1,
                      0.0 if bg is None else float(bg), _p(rgb), _p(depth), _p(opacity), _p(prob), _stream())
        ctx.save_for_backward(ray, rgb_s, sigma_s, depth_s)
        ctx.bg = bg
        ctx.set_materialize_grads(False)      # unused outputs (depth, opacity, prob) arrive as None instead of zero-filled tensors
        return rgb, depth, opacity, prob

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_opacity, g_prob):
        ray, rgb_s, sigma_s, depth_s = ctx.saved_tensors
        N, S = sigma_s.shape
        g = [None if t is None else _f32(t, "grad") for t in (g_rgb, g_depth, g_opacity, g_prob)]
        d_rgb_s, d_sigma_s, d_ray = torch.empty_like(rgb_s), torch.empty_like(sigma_s), torch.empty_like(ray)
        with timed("composite_bwd", N * S):
            _lib.call("niw_composite_bwd", _p(ray), _p(rgb_s), _p(sigma_s), _p(depth_s), N, S, 0 if ctx.bg is None else 1,
                      0.0 if ctx.bg is None else float(ctx.bg), _p(g[0]), _p(g[1]), _p(g[2]), _p(g[3]),
                      _p(d_rgb_s), _p(d_sigma_s), _p(d_ray), _stream())
        return d_ray, d_rgb_s, d_sigma_s, None, None


def composite(ray, rgb_s, sigma_s, depth_s, bg=None):
    """ray [N,3], rgb_s [N,S,3], sigma_s, depth_s [N,S] -> rgb [N,3], depth [N], opacity [N], prob [N,S]."""
    return _Composite.apply(ray, rgb_s, sigma_s, depth_s, bg)


# ------------------------------------------------------------------------------------------
# NVP warp (per-point part)
# ------------------------------------------------------------------------------------------

class _Warp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w_emb, view_b, w_head, pts, chan_w, index_window, ps_a, ps_b, inverse, window_dev=None, use_index_window=False):
        w_emb, view_b, w_head, pts = _f32(w_emb, "w_emb"), _f32(view_b, "view_b"), _f32(w_head, "w_head"), _f32(pts, "pts")
        B, P = pts.shape[0], pts.shape[1]
        out = torch.empty_like(pts)
        cw = _farr(chan_w, 6)
        iw = None if index_window is None else _farr(index_window, 6)       # by value with the launch: no H2D copy, no sync
        # training: the kernel leaves every coupling block's input point behind for the backward (which otherwise recomputes them)
        xin = torch.empty(B, P, 3, 3, device=pts.device, dtype=torch.float32) if (not inverse and any(ctx.needs_input_grad)) else None
        _lib.call("niw_warp_fwd", _p(w_emb), _p(view_b), _p(w_head), _p(pts), B, P, cw, iw, _p(window_dev), 1 if use_index_window else 0,
                  _p(ps_a), _p(ps_b), 1 if inverse else 0, _p(out), _p(xin), _stream())
        ctx.xin = xin
        ctx.save_for_backward(w_emb, view_b, w_head, pts, ps_a, ps_b)
        ctx.cw, ctx.iw, ctx.inverse, ctx.window_dev, ctx.use_iw = cw, iw, inverse, window_dev, use_index_window
        return out

    @staticmethod
    def backward(ctx, d_out):
        if ctx.inverse:
            raise _lib.NiwError("DeformNetwork.inverse is gradient-free here (the reference only uses it in debug helpers)")
        w_emb, view_b, w_head, pts, ps_a, ps_b = ctx.saved_tensors
        B, P = pts.shape[0], pts.shape[1]
        lib = _lib.load()
        ws = torch.empty(lib.niw_warp_bwd_workspace_floats(B, P), device=pts.device, dtype=torch.float32)
        d_w_emb, d_view_b, d_w_head = torch.empty_like(w_emb), torch.empty_like(view_b), torch.empty_like(w_head)   # fully overwritten
        d_pts = torch.empty_like(pts) if ctx.needs_input_grad[3] else None
        _lib.call("niw_warp_bwd", _p(w_emb), _p(view_b), _p(w_head), _p(pts), B, P, ctx.cw, ctx.iw, _p(ctx.window_dev), 1 if ctx.use_iw else 0,
                  _p(ps_a), _p(ps_b), _p(ctx.xin), _p(_f32(d_out, "d_out")), _p(ws), _p(d_w_emb), _p(d_view_b), _p(d_w_head), _p(d_pts), _stream())
        ctx.xin = None
        return d_w_emb, d_view_b, d_w_head, d_pts, None, None, None, None, None, None, None


WARP_PARAM_FLOATS = 165900
WARP_WEMB_FLOATS = 3 * (128 * 28 + 128 * 16)      # rows padded to 16-byte multiples (include/niw.h)
WARP_WHEAD_FLOATS = 3 * (128 + 1 + 3 * 128 + 3)


class _WarpPrep(torch.autograd.Function):
    """Weight norm + code projection + latent folding of DeformNetwork in one launch (and one for the
    backward).  `flat` is the parameter storage the module's Parameters are views of; the Parameters
    themselves are passed so that autograd routes their gradients."""

    @staticmethod
    def forward(ctx, flat, code, grad_sink, *params):
        code = _f32(code, "deformation_code")
        B = code.shape[0]
        dev = code.device
        w_emb = torch.empty(WARP_WEMB_FLOATS, device=dev)
        view_b = torch.empty(B, 3, 2, 128, device=dev)
        w_head = torch.empty(WARP_WHEAD_FLOATS, device=dev)
        ws = torch.empty(_lib.load().niw_warp_prep_fwd_workspace_floats(B), device=dev)
        _lib.call("niw_warp_prep_fwd", _p(flat), _p(code), B, _p(ws), _p(w_emb), _p(view_b), _p(w_head), _stream())
        ctx.flat, ctx.grad_sink = flat, grad_sink
        ctx.param_shapes = [p.shape for p in params]
        ctx.save_for_backward(code)
        ctx.set_materialize_grads(False)
        return w_emb, view_b, w_head

    @staticmethod
    def backward(ctx, d_w_emb, d_view_b, d_w_head):
        (code,) = ctx.saved_tensors
        B = code.shape[0]
        dev = code.device
        z = lambda g, n: torch.zeros(n, device=dev) if g is None else _f32(g, "grad")
        d_w_emb, d_view_b, d_w_head = z(d_w_emb, WARP_WEMB_FLOATS), z(d_view_b, B * 3 * 2 * 128), z(d_w_head, WARP_WHEAD_FLOATS)
        scratch = torch.empty(_lib.load().niw_warp_prep_bwd_workspace_floats(B), device=dev)
        sink = ctx.grad_sink                         # (flat parameter gradient, code gradient) buffers of the caller, or None
        d_params = sink[0] if sink is not None else torch.empty(WARP_PARAM_FLOATS, device=dev)
        d_code = sink[1] if sink is not None else torch.empty(B, 128, device=dev)
        _lib.call("niw_warp_prep_bwd", _p(ctx.flat), _p(code), B, _p(d_w_emb), _p(d_view_b), _p(d_w_head), _p(scratch),
                  _p(d_params), _p(d_code), _stream())
        grads, off = [], 0
        for shp in ctx.param_shapes:
            n = math.prod(shp)
            grads.append(None if sink is not None else d_params[off:off + n].view(shp))
            off += n
        return (None, None if sink is not None else d_code, None, *grads)


def warp_prepare(flat, params, code, grad_sink=None):
    """-> (w_emb, view_b, w_head) in the operand layout of niw_warp_fwd.  grad_sink: (d_params [WARP_PARAM_FLOATS], d_code [B*128])
    contiguous float32 buffers that receive the gradients in place of the Parameters' / the code's .grad."""
    if grad_sink is not None:
        if not torch.is_grad_enabled():
            grad_sink = None
        elif grad_sink[0].numel() != WARP_PARAM_FLOATS or grad_sink[1].numel() != code.numel():
            raise _lib.NiwError("warp_prepare: grad_sink = (buffer of WARP_PARAM_FLOATS, buffer of code.numel()) float32 elements")
    return _WarpPrep.apply(flat, code, grad_sink, *params)


def warp_points(w_emb, view_b, w_head, pts, chan_w, index_window=None, ps_a=None, ps_b=None, inverse=False, window_dev=None,
                use_index_window=False):
    """pts [B,P,3] -> warped [B,P,3]; see include/niw.h niw_warp_fwd for the operand layout.  window_dev: device tensor [12] =
    {chan_w, index_window} read by the kernels at run time (then use_index_window says whether the index window applies)."""
    return _Warp.apply(w_emb, view_b, w_head, pts, chan_w, index_window, ps_a, ps_b, inverse, window_dev, use_index_window)


# ------------------------------------------------------------------------------------------
# global-alignment loss: rotation of the rigid registration
# ------------------------------------------------------------------------------------------

class _KabschRotation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, M):
        M = _f32(M, "M")
        n = M.shape[0]
        R, Us, V = torch.empty_like(M), torch.empty_like(M), torch.empty_like(M)
        S = torch.empty(n, 3, device=M.device)
        _lib.call("niw_kabsch_rotation_fwd", _p(M), n, _p(R), _p(Us), _p(V), _p(S), _stream())
        ctx.save_for_backward(Us, V, S)
        return R

    @staticmethod
    def backward(ctx, dR):
        Us, V, S = ctx.saved_tensors
        dM = torch.empty_like(Us)
        _lib.call("niw_kabsch_rotation_bwd", _p(Us), _p(V), _p(S), _p(_f32(dR, "dR")), Us.shape[0], _p(dM), _stream())
        return dM


def kabsch_rotation(M):
    """M [n,3,3] -> R [n,3,3] = U diag(1,1,det(UV^T)) V^T of M = U S V^T, differentiable, no host sync."""
    return _KabschRotation.apply(M)


def rigid_registration(target, source, reduce_moments=None):
    """[R|t] [B,3,4] minimising sum_i |R x_i + t - y_i|^2 per view, x = target, y = source [B,N,3] (Kabsch with reflection fix:
    what `roma.rigid_points_registration(target, source)` returns at reference nerf_inn_llff.py:569, pose_models/inn.py:100).
    Two launches (fp64 moments, per-view solve); gradient-free: wherever the reference differentiates through it the derivative
    vanishes (see alignment_residual).  reduce_moments: callable summing the [B,16] float64 moment buffer over ranks in place
    (ray sharding: every rank registers the GLOBAL point set)."""
    target, source = _f32(target.detach(), "target"), _f32(source.detach(), "source")
    B, N = target.shape[0], target.shape[1]
    mom = torch.empty(B, 16, device=target.device, dtype=torch.float64)
    _lib.call("niw_align_moments", _p(target), _p(source), B, N, _p(mom), _stream())
    if reduce_moments is not None:
        reduce_moments(mom)
    poses = torch.empty(B, 3, 4, device=target.device)
    _lib.call("niw_align_solve", _p(mom), B, _p(poses), _stream())
    return poses


class _AlignResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, target, source, poses, n_norm):
        target, source, poses = _f32(target, "target"), _f32(source, "source"), _f32(poses, "poses")
        B, N = target.shape[0], target.shape[1]
        loss = torch.empty(1, device=target.device)
        d_target = torch.empty_like(target) if ctx.needs_input_grad[0] else None
        _lib.call("niw_align_loss", _p(target), _p(source), _p(poses), B, N, float(n_norm if n_norm is not None else 3 * B * N),
                  _p(loss), _p(d_target), _stream())
        ctx.save_for_backward(d_target)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (d_target,) = ctx.saved_tensors
        return (None if d_target is None else d_target * g), None, None, None


def alignment_residual(target, source, poses, n_norm=None):
    """mean |target - cam2world(source, poses)|^2 (nerf_inn_llff.py:571-572, nerf_inn_dtu.py:413-414) and its gradient w.r.t.
    `target` in one launch.  `poses` is treated as a constant: when it is the rigid registration of target onto source the loss
    is stationary in it, so the direct term is the total derivative (csrc/niw_align.hip; the DTU model detaches it anyway).
    n_norm: element count of the mean (3 * B * N of the GLOBAL batch under ray sharding)."""
    return _AlignResidual.apply(target, source, poses, n_norm)


# ------------------------------------------------------------------------------------------
# photometric loss
# ------------------------------------------------------------------------------------------

class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, image, ray_idx, n_norm, share):
        rgb, image = _f32(rgb, "rgb"), _f32(image, "image")
        B = image.shape[0]
        hw = image.shape[-1] * image.shape[-2] if image.dim() == 4 else image.shape[-1]
        if ray_idx is not None:
            ray_idx = ray_idx.to(device=rgb.device, dtype=torch.int64).contiguous()
        R = ray_idx.numel() if ray_idx is not None else hw
        first, count = (0, 0) if share is None else (int(share[0]), int(share[1]) - int(share[0]))
        if rgb.numel() != 3 * (count if share is not None else B * R):
            raise _lib.NiwError(f"mse_gather: rgb {tuple(rgb.shape)} does not hold " + (f"the {count} rays of the share {tuple(share)}" if share is not None else f"{B} x {R} rays"))
        loss = torch.empty(1, device=rgb.device)
        d_rgb = torch.empty_like(rgb)
        n = float(n_norm if n_norm is not None else rgb.numel())
        _lib.call("niw_mse_fwd_bwd", _p(rgb), _p(image), _p(ray_idx), B, R, hw, first, count, n, 1.0, _p(loss), _p(d_rgb), _stream())
        ctx.save_for_backward(d_rgb)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (d_rgb,) = ctx.saved_tensors
        return d_rgb * g, None, None, None, None


def mse_gather(rgb, image, ray_idx=None, n_norm=None, share=None):
    """mean((rgb - image[:, :, ray_idx])^2) with image [B,3,H,W]; n_norm overrides the element count of the mean (global batch
    under ray sharding).  share = (lo, hi): rgb holds the rays lo .. hi-1 of the flattened view-major [B][R] ray list (one rank's
    contiguous share, ..parallel.flat_share), any leading shape; None: the whole batch [B,R,3]."""
    return _MSE.apply(rgb, image, ray_idx, n_norm, share)
