"""CPU oracle for the volumetric-rendering hot path (TEST INFRASTRUCTURE ONLY).

This file is a CPU restatement (PyTorch, fp32, autograd for gradients) of the
algorithm the reference implements for the path named in BASELINE.json:
ray generation -> NVP invertible warp -> depth sampling -> positional encoding
-> NeRF MLP -> alpha compositing -> hierarchical resampling -> photometric loss.

It is NOT part of the product.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  The product path
(`neural_invertible_warp_amd`) never falls back to it: it raises when the HIP
library is missing.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the real reference
(in the build container only) and writes the fixtures under `tests/golden/`;
`tests/test_oracle_golden.py` checks every function below against them.
The only unpinned piece is the Kabsch registration (`rigid_registration`), whose
reference implementation lives in the absent third-party package roma==1.4.1
(reference requirements.txt:1); it restates the published Kabsch/Umeyama
algorithm and is marked "parity unpinned".

Every function cites the reference file:line it follows (paths relative to the
reference repository root).  All tensors are fp32; the functions follow the device of their inputs, so the
same restatement can be run on the GPU by torch's own ROCm kernels (tests at BASELINE shapes, bench.py's
`torch_rocm_baseline`) -- still the checker / a reported baseline, never the product path.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

# --------------------------------------------------------------------------------------
# deterministic parameter construction shared by oracle / golden generator / tests / bench
# (fixtures carry seeds, not weights: numpy's PCG64 stream is platform-stable)
# --------------------------------------------------------------------------------------

FEAT_DIMS = 256
RGB_HIDDEN = 128


def nerf_layer_shapes(L_3D: int = 10, L_view: int = 4, n_feat: int = 8, width: int = FEAT_DIMS,
                      rgb_hidden: int = RGB_HIDDEN, skip: Sequence[int] = (4,), view_dep: bool = True):
    """Layer shapes of the field MLP; follows model/nerf.py:373-400 (define_network)."""
    in3d = 3 + 6 * L_3D
    inview = 3 + 6 * L_view
    shapes = []
    for li in range(n_feat):
        k_in = in3d if li == 0 else width
        if li in skip:
            k_in += in3d
        k_out = width + (1 if li == n_feat - 1 else 0)
        shapes.append((f"mlp_feat.{li}", k_out, k_in))
    shapes.append(("mlp_rgb.0", rgb_hidden, width + (inview if view_dep else 0)))
    shapes.append(("mlp_rgb.1", 3, rgb_hidden))
    return shapes


def make_nerf_params(seed: int, L_3D: int = 10, L_view: int = 4, bias_scale: float = 0.05) -> Params:
    """Xavier-uniform-like weights (model/nerf.py:404-414) from a numpy PCG64 stream.

    Biases are drawn non-zero (the reference zero-initialises them) so that parity tests
    exercise the bias path; the architecture is the one of all five BASELINE configs.
    """
    rng = np.random.default_rng(seed)
    p: Params = {}
    for name, k_out, k_in in nerf_layer_shapes(L_3D, L_view):
        gain = math.sqrt(2.0)
        if name == "mlp_rgb.1":
            gain = 1.0
        bound = gain * math.sqrt(6.0 / (k_in + k_out))
        w = rng.uniform(-bound, bound, size=(k_out, k_in)).astype(np.float32)
        if name == "mlp_feat.7":  # first output row (density) uses gain 1 (nerf.py:409-411)
            b1 = math.sqrt(6.0 / (k_in + 1))
            w[0] = rng.uniform(-b1, b1, size=(k_in,)).astype(np.float32)
        b = (rng.standard_normal(k_out) * bias_scale).astype(np.float32)
        p[name + ".weight"] = torch.from_numpy(w)
        p[name + ".bias"] = torch.from_numpy(b)
    return p


WARP_HID = 128
WARP_LAT = 128
WARP_MULTIRES = 6
WARP_BLOCKS = 3


def make_warp_params(seed: int, perturb: float = 0.02, d_feature: int = WARP_LAT) -> Params:
    """DeformNetwork parameters as constructed at model/nvp/nvp_ndr.py:230-345 with the
    arguments of model/barf_inn_llff.py:54-55.  The reference zero-initialises the last layer
    of every sub-network (identity warp); `perturb` > 0 draws them N(0, perturb) instead so
    that the warp is non-trivial in parity tests."""
    rng = np.random.default_rng(seed)
    ea, eb = 2 * (1 + 2 * WARP_MULTIRES), 1 * (1 + 2 * WARP_MULTIRES)  # 26, 13
    p: Params = {}
    std = math.sqrt(2.0) / math.sqrt(WARP_HID)

    def t(a):
        return torch.from_numpy(np.asarray(a, dtype=np.float32))

    for b in range(WARP_BLOCKS):
        for part, emb, ori, out in (("a", ea, 2, 1), ("b", eb, 1, 3)):
            v = np.zeros((WARP_HID, emb + d_feature), np.float32)
            v[:, :ori] = rng.standard_normal((WARP_HID, ori)) * std
            # the reference leaves every other column at zero; fill them lightly so that the
            # embedding / latent columns matter in tests
            v[:, ori:] = rng.standard_normal((WARP_HID, emb + d_feature - ori)) * (perturb if perturb else 0.0)
            g = np.linalg.norm(v, axis=1, keepdims=True).astype(np.float32)
            g = g * (1.0 + 0.1 * rng.standard_normal(g.shape)).astype(np.float32) if perturb else g
            p[f"lin{b}_{part}_0.weight_v"] = t(v)
            p[f"lin{b}_{part}_0.weight_g"] = t(g)
            p[f"lin{b}_{part}_0.bias"] = t(rng.standard_normal(WARP_HID) * perturb)
            p[f"lin{b}_{part}_1.weight"] = t(rng.standard_normal((out, WARP_HID)) * perturb)
            p[f"lin{b}_{part}_1.bias"] = t(rng.standard_normal(out) * perturb)
        p[f"lin{b}_c.weight"] = t(rng.standard_normal((d_feature, d_feature)) * perturb)
        p[f"lin{b}_c.bias"] = t(rng.standard_normal(d_feature) * perturb)
    return p


def make_latent(seed: int, n_views: int, dim: int = WARP_LAT) -> Tensor:
    """nn.Embedding default init N(0,1) (model/barf_inn_llff.py:42)."""
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.standard_normal((n_views, dim)).astype(np.float32))


# --------------------------------------------------------------------------------------
# R1 / R2 / R3: ray generation
# --------------------------------------------------------------------------------------

def pixel_grid(H: int, W: int, device=None) -> Tensor:
    """Pixel-centre grid, index = y*W + x; camera.py:369-374."""
    y = torch.arange(H, dtype=torch.float32, device=device) + 0.5
    x = torch.arange(W, dtype=torch.float32, device=device) + 0.5
    Y, X = torch.meshgrid(y, x, indexing="ij")
    return torch.stack([X, Y], dim=-1).reshape(-1, 2)


def _hom(x: Tensor) -> Tensor:
    """camera.py:330-333."""
    return torch.cat([x, torch.ones_like(x[..., :1])], dim=-1)


def invert_pose(pose: Tensor) -> Tensor:
    """[R|t] -> [R^T | -R^T t]; camera.py:89-95."""
    R, t = pose[..., :3], pose[..., 3:]
    Rt = R.transpose(-1, -2)
    return torch.cat([Rt, -Rt @ t], dim=-1)


def cam2world(X: Tensor, pose: Tensor) -> Tensor:
    """camera.py:343-346."""
    return _hom(X) @ invert_pose(pose).transpose(-1, -2)


def img2cam(X: Tensor, intr: Tensor) -> Tensor:
    """camera.py:341-342."""
    return X @ intr.inverse().transpose(-1, -2)


def unwarped_center_and_grid(H: int, W: int, intr: Tensor, ray_idx: Optional[Tensor] = None,
                             pose_init: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """R1: camera.py:359-390 (get_unwarped_center_and_ray)."""
    B = intr.shape[0]
    xy = pixel_grid(H, W, intr.device).repeat(B, 1, 1)
    grid = img2cam(_hom(xy), intr)
    center = torch.zeros_like(grid)
    if pose_init is not None:
        grid = cam2world(grid, pose_init)
        center = cam2world(center, pose_init)
    if ray_idx is not None:
        center, grid = center[:, ray_idx], grid[:, ray_idx]
    return center, grid


def center_and_ray(H: int, W: int, pose: Tensor, intr: Tensor) -> Tuple[Tensor, Tensor]:
    """R2: camera.py:419-443 (get_center_and_ray); pose is world->camera."""
    B = pose.shape[0]
    xy = pixel_grid(H, W, intr.device).repeat(B, 1, 1)
    grid = img2cam(_hom(xy), intr)
    center = torch.zeros_like(grid)
    grid = cam2world(grid, pose)
    center = cam2world(center, pose)
    return center, grid - center


def convert_ndc(center: Tensor, ray: Tensor, intr: Tensor, near: float = 1.0) -> Tuple[Tensor, Tensor]:
    """R3: camera.py:523-540 (convert_NDC)."""
    center = center + (near - center[..., 2:]) / ray[..., 2:] * ray
    cx, cy, cz = center.unbind(-1)
    rx, ry, rz = ray.unbind(-1)
    sx = (intr[:, 0, 0] / intr[:, 0, 2])[:, None]
    sy = (intr[:, 1, 1] / intr[:, 1, 2])[:, None]
    c = torch.stack([sx * (cx / cz), sy * (cy / cz), 1 - 2 * near / cz], dim=-1)
    r = torch.stack([sx * (rx / rz - cx / cz), sy * (ry / rz - cy / cz), 2 * near / cz], dim=-1)
    return c, r


# --------------------------------------------------------------------------------------
# W2 / W3 / W4: annealed embedder and the NVP coupling warp
# --------------------------------------------------------------------------------------

def anneal_window(alpha_ratio: float, n_freq: int) -> list:
    """Nerfies window w_i = (1 - cos(pi * clamp(alpha*n - i, 0, 1))) / 2; model/nvp/embedder.py:47-49."""
    return [(1.0 - math.cos(math.pi * max(min(alpha_ratio * n_freq - i, 1.0), 0.0))) * 0.5 for i in range(n_freq)]


def warp_embed(x: Tensor, alpha_ratio: float, n_freq: int = WARP_MULTIRES, reference_exact: bool = True) -> Tensor:
    """W2: model/nvp/embedder.py:41-50.  Channels: identity(d), then per band sin(d), cos(d);
    band frequency 2^i * pi.  `reference_exact` reproduces the dim-1 slicing of line 47: on
    the reference's 4-D inputs [B, P, 1, d] the annealing window multiplies ALL channels of
    points (2i+1)d .. (2i+3)d-1 along dim 1 instead of the channels of band i."""
    d = x.shape[-1]
    chans = [x]
    for i in range(n_freq):
        f = float(np.float32(math.pi)) * (2.0 ** i)   # the reference's band table is an fp32 tensor (embedder.py:26)
        chans += [torch.sin(x * f), torch.cos(x * f)]
    out = torch.cat(chans, dim=-1)
    w = anneal_window(alpha_ratio, n_freq)
    if reference_exact:
        scale = torch.ones(out.shape[1], dtype=out.dtype)
        for i in range(n_freq):
            scale[(2 * i + 1) * d:(2 * i + 3) * d] *= w[i]
        shape = [1] * out.dim()
        shape[1] = out.shape[1]
        return out * scale.view(shape).to(out.device)
    cw = torch.ones(out.shape[-1], dtype=out.dtype)
    for i in range(n_freq):
        cw[(2 * i + 1) * d:(2 * i + 3) * d] = w[i]
    return out * cw.to(out.device)


def _wn_weight(p: Params, name: str) -> Tensor:
    """old-style nn.utils.weight_norm, dim=0: w = g * v / ||v||_row (nvp_ndr.py:291-292)."""
    v, g = p[name + ".weight_v"], p[name + ".weight_g"]
    return v * (g / v.norm(dim=1, keepdim=True))


def _softplus100(x: Tensor) -> Tensor:
    """nn.Softplus(beta=100), threshold 20 (nvp_ndr.py:362)."""
    return F.softplus(x, beta=100.0, threshold=20.0)


_FOCUS = (2, 1, 0)            # mode 0,1,2 focus on z,y,x (nvp_ndr.py:389-399)
_OTHER = ((0, 1), (0, 2), (1, 2))


def warp_forward(p: Params, code: Tensor, pts: Tensor, alpha_ratio: float = 0.0,
                 reference_exact: bool = True) -> Tensor:
    """W3: DeformNetwork.forward, model/nvp/nvp_ndr.py:365-468 (form 0 only: n_blocks=3).
    code [B,128]; pts [B,P,1,3] -> [B,P,1,3]."""
    x = pts
    B, P = pts.shape[0], pts.shape[1]
    for b in range(WARP_BLOCKS):
        cb = F.linear(code, p[f"lin{b}_c.weight"], p[f"lin{b}_c.bias"]) + code       # :381
        cb = cb[:, None, None, :].expand(B, P, 1, -1)                                # :570-572
        fi, oi = _FOCUS[b], list(_OTHER[b])
        focus, other = x[..., [fi]], x[..., oi]
        # part a (:412-426)
        h = torch.cat([warp_embed(other, alpha_ratio, reference_exact=reference_exact), cb], dim=-1)
        h = _softplus100(F.linear(h, _wn_weight(p, f"lin{b}_a_0"), p[f"lin{b}_a_0.bias"]))
        focus = focus - F.linear(h, p[f"lin{b}_a_1.weight"], p[f"lin{b}_a_1.bias"])
        # part b (:429-452)
        h = torch.cat([warp_embed(focus, alpha_ratio, reference_exact=reference_exact), cb], dim=-1)
        h = _softplus100(F.linear(h, _wn_weight(p, f"lin{b}_b_0"), p[f"lin{b}_b_0.bias"]))
        o = F.linear(h, p[f"lin{b}_b_1.weight"], p[f"lin{b}_b_1.bias"])
        th, tr = o[..., 0], o[..., 1:]
        c, s = torch.cos(th), torch.sin(th)
        d0, d1 = other[..., 0] - tr[..., 0], other[..., 1] - tr[..., 1]
        # euler2rot_2dinv (:166-174) assembles [[cos, sin], [-sin, cos]]
        n0 = c * d0 + s * d1
        n1 = -s * d0 + c * d1
        cols = [None, None, None]
        cols[fi] = focus[..., 0]
        cols[oi[0]], cols[oi[1]] = n0, n1
        x = torch.stack(cols, dim=-1)                                                 # :453-459
    return x


def warp_inverse(p: Params, code: Tensor, pts: Tensor, alpha_ratio: float = 0.0,
                 reference_exact: bool = True) -> Tensor:
    """W4: DeformNetwork.inverse, model/nvp/nvp_ndr.py:471-567."""
    x = pts
    B, P = pts.shape[0], pts.shape[1]
    for b in reversed(range(WARP_BLOCKS)):
        cb = F.linear(code, p[f"lin{b}_c.weight"], p[f"lin{b}_c.bias"]) + code
        cb = cb[:, None, None, :].expand(B, P, 1, -1)
        fi, oi = _FOCUS[b], list(_OTHER[b])
        pair, single = x[..., oi], x[..., [fi]]
        h = torch.cat([warp_embed(single, alpha_ratio, reference_exact=reference_exact), cb], dim=-1)
        h = _softplus100(F.linear(h, _wn_weight(p, f"lin{b}_b_0"), p[f"lin{b}_b_0.bias"]))
        o = F.linear(h, p[f"lin{b}_b_1.weight"], p[f"lin{b}_b_1.bias"])
        th, tr = o[..., 0], o[..., 1:]
        c, s = torch.cos(th), torch.sin(th)
        # euler2rot_2d (:155-163) assembles [[cos, -sin], [sin, cos]]
        n0 = c * pair[..., 0] - s * pair[..., 1] + tr[..., 0]
        n1 = s * pair[..., 0] + c * pair[..., 1] + tr[..., 1]
        pair = torch.stack([n0, n1], dim=-1)
        h = torch.cat([warp_embed(pair, alpha_ratio, reference_exact=reference_exact), cb], dim=-1)
        h = _softplus100(F.linear(h, _wn_weight(p, f"lin{b}_a_0"), p[f"lin{b}_a_0.bias"]))
        single = single + F.linear(h, p[f"lin{b}_a_1.weight"], p[f"lin{b}_a_1.bias"])
        cols = [None, None, None]
        cols[fi] = single[..., 0]
        cols[oi[0]], cols[oi[1]] = pair[..., 0], pair[..., 1]
        x = torch.stack(cols, dim=-1)
    return x


def warped_rays(p: Params, latent: Tensor, center_cam: Tensor, grid_cam: Tensor, alpha_ratio: float,
                reference_exact: bool = True) -> Tuple[Tensor, Tensor, Tensor]:
    """W5: train branch of Graph.get_pose, model/barf_inn_llff.py:325-364 (inputs detached,
    cat([grid, center], 1), whole latent table as code).  Returns ray, center_3D, grid_3D."""
    R = grid_cam.shape[1]
    pts = torch.cat([grid_cam.detach(), center_cam.detach()], dim=1).unsqueeze(2)
    out = warp_forward(p, latent, pts, alpha_ratio, reference_exact).squeeze(2)
    grid3, cen3 = out[:, :R], out[:, R:]
    return grid3 - cen3, cen3, grid3


# --------------------------------------------------------------------------------------
# S1 / S2 / P1 / M2 / M3: sampling, encoding, field MLP
# --------------------------------------------------------------------------------------

def sample_depth(u, S: int, depth_range: Sequence[float], param: str) -> Tensor:
    """S1: Graph.sample_depth, model/nerf.py:334-344.  `u` is the stratified draw [B,R,S,1]
    (torch.rand in the reference) or the float 0.5."""
    lo, hi = depth_range
    r = u + torch.arange(S, dtype=torch.float32, device=getattr(u, "device", None))[None, None, :, None]
    d = r / S * (hi - lo) + lo
    if param == "metric":
        return d
    if param == "inverse":
        return 1 / (d + 1e-8)
    raise KeyError(param)


def points_from_depth(center: Tensor, ray: Tensor, depth: Tensor) -> Tensor:
    """S2: camera.py:517-521 (multi_samples=True)."""
    return center[:, :, None] + ray[:, :, None] * depth


def c2f_weights(progress: float, barf_c2f: Optional[Sequence[float]], L: int) -> Tensor:
    """BARF coarse-to-fine band weights, model/barf_inn_llff.py:430-436."""
    if barf_c2f is None:
        return torch.ones(L)
    start, end = barf_c2f
    alpha = (torch.tensor(float(progress)) - start) / (end - start) * L
    k = torch.arange(L, dtype=torch.float32)
    return (1 - (alpha - k).clamp(min=0, max=1).mul(math.pi).cos()) / 2


def positional_encoding(x: Tensor, L: int, band_w: Optional[Tensor] = None) -> Tensor:
    """P1: NeRF.positional_encoding, model/nerf.py:476-483 (+ c2f mask barf_inn_llff.py:437-439).
    Layout per coordinate: [sin f0..f(L-1), cos f0..f(L-1)]."""
    freq = (2 ** torch.arange(L, dtype=torch.float32) * np.pi).to(x.device)       # (formed on the host: the same fp32 table everywhere)
    spec = x[..., None] * freq
    enc = torch.stack([spec.sin(), spec.cos()], dim=-2)
    if band_w is not None:
        enc = enc * band_w.to(x.device)
    return enc.reshape(*x.shape[:-1], -1)


def _density_act(x: Tensor, kind: str) -> Tensor:
    if kind == "relu":
        return F.relu(x)
    if kind == "softplus":
        return F.softplus(x)
    raise KeyError(kind)


def nerf_forward(p: Params, points: Tensor, ray_unit: Tensor, L_3D: int = 10, L_view: int = 4,
                 density_activ: str = "softplus", w3d: Optional[Tensor] = None, wview: Optional[Tensor] = None,
                 density_noise: Optional[Tensor] = None, skip: Sequence[int] = (4,)) -> Tuple[Tensor, Tensor]:
    """M2: NeRF.forward, model/nerf.py:416-447."""
    enc = torch.cat([points, positional_encoding(points, L_3D, w3d)], dim=-1)
    feat = enc
    n_feat = 8
    for li in range(n_feat):
        if li in skip:
            feat = torch.cat([feat, enc], dim=-1)
        feat = F.linear(feat, p[f"mlp_feat.{li}.weight"], p[f"mlp_feat.{li}.bias"])
        if li == n_feat - 1:
            raw = feat[..., 0]
            if density_noise is not None:
                raw = raw + density_noise
            density = _density_act(raw, density_activ)
            feat = feat[..., 1:]
        feat = F.relu(feat)
    renc = torch.cat([ray_unit, positional_encoding(ray_unit, L_view, wview)], dim=-1)
    feat = torch.cat([feat, renc], dim=-1)
    feat = F.relu(F.linear(feat, p["mlp_rgb.0.weight"], p["mlp_rgb.0.bias"]))
    rgb = torch.sigmoid(F.linear(feat, p["mlp_rgb.1.weight"], p["mlp_rgb.1.bias"]))
    return rgb, density


def forward_samples(p: Params, center: Tensor, ray: Tensor, depth: Tensor, **kw) -> Tuple[Tensor, Tensor]:
    """M3: NeRF.forward_samples, model/nerf.py:449-456."""
    pts = points_from_depth(center, ray, depth)
    unit = F.normalize(ray, dim=-1)[..., None, :].expand_as(pts)
    return nerf_forward(p, pts, unit, **kw)


# --------------------------------------------------------------------------------------
# C1 / H1 / H2: compositing and hierarchical resampling
# --------------------------------------------------------------------------------------

def composite(ray: Tensor, rgb_s: Tensor, sigma_s: Tensor, depth_s: Tensor,
              bgcolor: Optional[float] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """C1: NeRF.composite, model/nerf.py:458-474."""
    length = ray.norm(dim=-1, keepdim=True)
    intv = depth_s[..., 1:, 0] - depth_s[..., :-1, 0]
    intv = torch.cat([intv, torch.full_like(intv[..., :1], 1e10)], dim=2)
    sd = sigma_s * (intv * length)
    alpha = 1 - torch.exp(-sd)
    T = torch.exp(-torch.cat([torch.zeros_like(sd[..., :1]), sd[..., :-1]], dim=2).cumsum(dim=2))
    prob = (T * alpha)[..., None]
    depth = (depth_s * prob).sum(dim=2)
    rgb = (rgb_s * prob).sum(dim=2)
    opacity = prob.sum(dim=2)
    if bgcolor is not None:
        rgb = rgb + bgcolor * (1 - opacity)
    return rgb, depth, opacity, prob


def sample_depth_from_pdf(pdf: Tensor, S: int, Sf: int, depth_range: Sequence[float]) -> Tensor:
    """H1: Graph.sample_depth_from_pdf, model/nerf.py:346-365 (un-normalised pdf, mid-point
    quantiles, bins in the parametrised space)."""
    lo, hi = depth_range
    # `pdf.cumsum(dim=-1)` of the reference, as torch's CPU kernel evaluates it for float32: ONE sequential pass that accumulates in double
    # (at::acc_type<float, false>) and rounds every prefix to float.  Written out so that the oracle is the same function on every device
    # (torch's GPU cumsum is a parallel fp32 scan: other roundings, other fine sample positions, which the 2^9 pi band then amplifies);
    # bit-identical to pdf.cumsum on the CPU (tests/test_oracle_golden.py); the HIP kernel follows the same sequential double sum.
    cdf = pdf.double().cumsum(dim=-1).to(pdf.dtype)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    g = torch.linspace(0, 1, Sf + 1)                                            # (host linspace, then moved: one table on every device)
    unif = (0.5 * (g[:-1] + g[1:])).to(pdf.device).repeat(*cdf.shape[:-1], 1)
    idx = torch.searchsorted(cdf, unif, right=True)
    bins = torch.linspace(lo, hi, S + 1).to(pdf.device).repeat(*cdf.shape[:-1], 1)
    il, ih = (idx - 1).clamp(min=0), idx.clamp(max=S)
    dl, dh = bins.gather(2, il), bins.gather(2, ih)
    cl, ch = cdf.gather(2, il), cdf.gather(2, ih)
    t = (unif - cl) / (ch - cl + 1e-8)
    return (dl + t * (dh - dl))[..., None]


def merge_depth(coarse: Tensor, fine: Tensor) -> Tensor:
    """H2: model/nerf.py:314-315 (cat + ascending sort)."""
    return torch.cat([coarse, fine], dim=2).sort(dim=2).values


# --------------------------------------------------------------------------------------
# G1 / L1: render orchestration and photometric loss
# --------------------------------------------------------------------------------------

def render_rays(p: Params, center: Tensor, ray: Tensor, u, S: int, depth_range, param: str,
                p_fine: Optional[Params] = None, Sf: int = 0, **mlp_kw) -> Dict[str, Tensor]:
    """G1: Graph.render_local, model/nerf_inn_llff.py:581-612 (== Graph.render nerf.py:304-319
    once center/ray are given)."""
    depth = sample_depth(u, S, depth_range, param)
    rgb_s, sig_s = forward_samples(p, center, ray, depth, **{k: v for k, v in mlp_kw.items() if k != "density_noise_fine"})
    rgb, d, op, prob = composite(ray, rgb_s, sig_s, depth)
    out = dict(rgb=rgb, depth=d, opacity=op, prob=prob, depth_samples=depth)
    if p_fine is not None:
        with torch.no_grad():
            fine = sample_depth_from_pdf(prob[..., 0], S, Sf, depth_range)
            depth_all = merge_depth(depth, fine)
        kw = dict(mlp_kw)
        kw.pop("density_noise", None)
        # (the reference draws a fresh randn_like in EVERY train-mode forward, model/nerf.py:428-429: the fine pass takes its own tensor)
        noise_fine = kw.pop("density_noise_fine", None)
        if noise_fine is not None:
            kw["density_noise"] = noise_fine
        rgb_s, sig_s = forward_samples(p_fine, center, ray, depth_all, **kw)
        rgb_f, d_f, op_f, _ = composite(ray, rgb_s, sig_s, depth_all)
        out.update(rgb_fine=rgb_f, depth_fine=d_f, opacity_fine=op_f, depth_samples_fine=depth_all)
    return out


def gather_pixels(image: Tensor, ray_idx: Optional[Tensor]) -> Tensor:
    """image [B,3,H,W] -> [B,R,3]; model/nerf.py:279-281."""
    B = image.shape[0]
    img = image.reshape(B, 3, -1).permute(0, 2, 1)
    return img if ray_idx is None else img[:, ray_idx]


def mse_loss(pred: Tensor, label: Tensor) -> Tensor:
    """model/base.py:209-211."""
    return ((pred.contiguous() - label) ** 2).mean()


def rigid_registration(x: Tensor, y: Tensor) -> Tuple[Tensor, Tensor]:
    """Kabsch: R, t minimising sum ||R x + t - y||^2 (maps x -> y).  Restates the published
    algorithm of roma==1.4.1 `rigid_points_registration` (absent third-party dependency,
    reference requirements.txt:1; call site model/nerf_inn_llff.py:569).  PARITY UNPINNED."""
    xm, ym = x.mean(dim=1, keepdim=True), y.mean(dim=1, keepdim=True)
    M = (y - ym).transpose(1, 2) @ (x - xm)
    U, _, Vt = torch.linalg.svd(M)
    det = torch.det(U @ Vt)
    D = torch.diag_embed(torch.stack([torch.ones_like(det), torch.ones_like(det), det], dim=-1))
    Rm = U @ D @ Vt
    t = ym[:, 0] - (Rm @ xm.transpose(1, 2))[..., 0]
    return Rm, t


def global_alignment_loss(grid_cam, center_cam, grid_3D, center_3D) -> Tuple[Tensor, Tensor]:
    """model/nerf_inn_llff.py:563-572.  Returns (loss, svd_poses[B,3,4])."""
    source = torch.cat([grid_cam, center_cam], dim=1)
    target = torch.cat([grid_3D, center_3D], dim=1)
    Rg, tg = rigid_registration(target, source)
    poses = torch.cat([Rg, tg[..., None]], dim=-1)
    return mse_loss(target, cam2world(source, poses)), poses


def inn_train_step(nerf_p: Params, warp_p: Params, latent: Tensor, image: Tensor, intr: Tensor,
                   ray_idx: Tensor, u, H: int, W: int, S: int, depth_range, param: str, alpha_ratio: float,
                   nerf_fine_p: Optional[Params] = None, Sf: int = 0, reference_exact: bool = True,
                   pose_init: Optional[Tensor] = None, ga_weight: Optional[float] = None, reference_cost: bool = False,
                   **mlp_kw) -> Dict[str, Tensor]:
    """G0 (INN variant): Graph.forward(mode="train") + compute_loss,
    model/nerf_inn_llff.py:493-573 with get_pose of model/barf_inn_llff.py:305-364.
    reference_cost: also make the un-warped ray grid call the reference makes a FIRST time and discards (nerf_inn_llff.py:519, then again
    inside get_pose, barf_inn_llff.py:325) -- no effect on any result; the timed CPU baseline (bench.py cpu_baseline) sets it so that the
    restatement costs what the reference costs (tools/calibrate_oracle.py)."""
    if reference_cost:
        unwarped_center_and_grid(H, W, intr, ray_idx, pose_init)
    center_cam, grid_cam = unwarped_center_and_grid(H, W, intr, ray_idx, pose_init)
    ray, center, grid3 = warped_rays(warp_p, latent, center_cam, grid_cam, alpha_ratio, reference_exact)
    out = render_rays(nerf_p, center, ray, u, S, depth_range, param, p_fine=nerf_fine_p, Sf=Sf, **mlp_kw)
    target = gather_pixels(image, ray_idx)
    out["loss_render"] = mse_loss(out["rgb"], target)
    total = out["loss_render"]
    if nerf_fine_p is not None:
        out["loss_render_fine"] = mse_loss(out["rgb_fine"], target)
        total = total + out["loss_render_fine"]
    if ga_weight is not None:
        out["loss_ga"], out["svd_poses"] = global_alignment_loss(grid_cam, center_cam, grid3, center)
        total = total + (10.0 ** ga_weight) * out["loss_ga"]
    out.update(loss=total, ray=ray, center=center, grid_3D=grid3)
    return out
