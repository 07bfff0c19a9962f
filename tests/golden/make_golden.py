#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING THE REAL REFERENCE (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference lives read-only at /root/reference and never travels to the GPU box; only
the small .npz fixtures written here do.  Fixtures carry seeds, not weights: every
parameter tensor is regenerated from `oracle.niw_oracle.make_*_params(seed)` (numpy PCG64,
platform-stable) on both sides.  Stored: inputs that cannot be regenerated, reference
outputs, and for gradients the per-tensor L2 norm plus a strided sample.

Peripheral imports the reference pulls in at module top level but that are absent from this
image (easydict, ipdb, termcolor, lpips, torchvision, visdom, tensorboard, roma, imageio,
cv2) are replaced by inert stand-ins in sys.modules; none of them is on the measured path.
"""
import importlib.machinery
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import niw_oracle as O


# ----------------------------------------------------------------------------- stand-ins
class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(EasyDict(x) if isinstance(x, dict) else x for x in v)
        super().__setitem__(k, v)

    __setitem__ = __setattr__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, d=None, **kw):
        for k, v in dict(d or {}, **kw).items():
            setattr(self, k, v)

    def pop(self, k, *a):
        return super().pop(k, *a)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    _stub("easydict", EasyDict=EasyDict)
    _stub("ipdb", set_trace=lambda *a, **k: None)
    _stub("termcolor", colored=lambda s, *a, **k: s)

    class _LP:
        def __init__(self, *a, **k):
            pass

        def to(self, *a, **k):
            return self

    _stub("lpips", LPIPS=_LP)
    _stub("visdom", Visdom=object)
    tv = _stub("torchvision")
    tvt = _stub("torchvision.transforms")
    tvf = _stub("torchvision.transforms.functional")
    tv.transforms, tvt.functional = tvt, tvf
    _stub("imageio")
    _stub("cv2")
    _stub("tensorboard")
    tb = _stub("torch.utils.tensorboard", SummaryWriter=object)
    torch.utils.tensorboard = tb

    def rigid_points_registration(x, y, *a, **k):  # inert on the fixtures (GA loss disabled)
        raise RuntimeError("roma is not available; global-alignment parity is unpinned")

    _stub("roma", rigid_points_registration=rigid_points_registration)
    for n in ("matplotlib", "matplotlib.pyplot", "PIL", "PIL.Image", "PIL.ImageFile"):
        try:
            __import__(n)
        except Exception:
            _stub(n)


def load_opt(yaml_name, model, **over):
    import options
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        opt = options.load_options(f"options/{yaml_name}.yaml")
    finally:
        os.chdir(cwd)
    opt.model, opt.yaml, opt.cpu = model, yaml_name, True
    opt.output_root = "/tmp/niw_golden_out"
    for k, v in over.items():
        node = opt
        ks = k.split(".")
        for kk in ks[:-1]:
            node = node[kk]
        node[ks[-1]] = v
    opt.device = "cpu"
    opt.H, opt.W = opt.data.image_size
    torch.manual_seed(0)
    return opt


def set_params(module, params, prefix=""):
    sd = module.state_dict()
    for k, v in params.items():
        assert prefix + k in sd, (prefix + k, list(sd)[:5])
        assert sd[prefix + k].shape == v.shape, (k, sd[prefix + k].shape, v.shape)
    with torch.no_grad():
        for k, v in params.items():
            sd[prefix + k].copy_(v)


def gsum(t, stride=37):
    f = t.detach().reshape(-1).double()
    return dict(norm=np.array(float(f.norm())), sample=f[::stride].float().numpy(), stride=np.array(stride))


def save(name, **arrs):
    flat = {}
    for k, v in arrs.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}.{kk}"] = np.asarray(vv)
        elif isinstance(v, torch.Tensor):
            flat[k] = v.detach().cpu().numpy()
        else:
            flat[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(flat)} arrays)")


def rand_intr(rng, B, H, W):
    K = np.zeros((B, 3, 3), np.float32)
    K[:, 0, 0] = 0.8 * W * (1 + 0.05 * rng.standard_normal(B))
    K[:, 1, 1] = 0.8 * W * (1 + 0.05 * rng.standard_normal(B))
    K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = W / 2, H / 2, 1
    return torch.from_numpy(K)


def rand_pose(rng, B, rot=0.2, trans=0.3):
    import camera
    se3 = torch.from_numpy((rng.standard_normal((B, 6)) * np.array([rot] * 3 + [trans] * 3)).astype(np.float32))
    return camera.lie.se3_to_SE3(se3)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import camera
    import model.nvp.nvp_ndr as nvp
    import model.nvp.embedder as emb
    import model.nerf as ref_nerf
    import model.barf_inn_llff as ref_barf
    from easydict import EasyDict as edict

    rng = np.random.default_rng(1234)
    H, W, B = 12, 16, 3

    # ---------------------------------------------------------------- R1 / R2 / R3
    opt = edict(H=H, W=W, device="cpu", camera=edict(model="perspective", ndc=False))
    intr = rand_intr(rng, B, H, W)
    pose = rand_pose(rng, B)
    ray_idx = torch.from_numpy(rng.permutation(H * W)[:10].astype(np.int64))
    c1, g1 = camera.get_unwarped_center_and_ray(opt, intr=intr, ray_idx=ray_idx)
    c1p, g1p = camera.get_unwarped_center_and_ray(opt, intr=intr, ray_idx=ray_idx, pose_init=pose)
    c2, r2 = camera.get_center_and_ray(opt, pose, intr=intr)
    cn, rn = camera.convert_NDC(opt, c2[:, ray_idx] + torch.tensor([0., 0., 3.]), r2[:, ray_idx] + torch.tensor([0., 0., 2.]), intr=intr)
    save("raygen", H=H, W=W, intr=intr, pose=pose, ray_idx=ray_idx, center_unwarped=c1, grid_unwarped=g1,
         center_unwarped_pose=c1p, grid_unwarped_pose=g1p, center=c2, ray=r2, ndc_center=cn, ndc_ray=rn)

    # ---------------------------------------------------------------- W2 embedder (incl. dim-1 quirk)
    e2, _ = emb.get_embedder(6, input_dims=2)
    e1, _ = emb.get_embedder(6, input_dims=1)
    x2 = torch.from_numpy(rng.uniform(-1.5, 1.5, (2, 30, 1, 2)).astype(np.float32))
    x1 = torch.from_numpy(rng.uniform(-1.5, 1.5, (2, 30, 1, 1)).astype(np.float32))
    out = dict(x2=x2, x1=x1)
    for a in (0.0, 0.3, 0.55, 1.0):
        out[f"e2_a{a}"] = e2(x2.clone(), a)
        out[f"e1_a{a}"] = e1(x1.clone(), a)
    out["e2_flat_a0.3"] = e2(x2.reshape(-1, 2).clone(), 0.3)   # 2-D input: per-channel behaviour
    save("embedder", **out)

    # ---------------------------------------------------------------- W3 / W4 warp
    wp = O.make_warp_params(seed=11, perturb=0.05)
    net = nvp.DeformNetwork(d_feature=128, d_in=3, d_out_1=1, d_out_2=3, n_blocks=3, d_hidden=128, n_layers=1,
                            skip_in=[], multires=6, weight_norm=True, actfn="softplus")
    set_params(net, wp)
    code = O.make_latent(seed=12, n_views=B)
    code_p = torch.nn.Parameter(code.clone())
    pts = torch.from_numpy(rng.uniform(-1.0, 1.0, (B, 40, 1, 3)).astype(np.float32))
    out = dict(pts=pts, warp_seed=11, warp_perturb=0.05, latent_seed=12)
    for a in (0.3, 1.0):
        net.zero_grad()
        code_p.grad = None
        y = net.forward(code_p, pts, alpha_ratio=a)
        out[f"fwd_a{a}"] = y
        out[f"inv_a{a}"] = net.inverse(code_p, y.detach(), a)
        gw = torch.from_numpy(np.random.default_rng(5).standard_normal(y.shape).astype(np.float32))
        (y * gw).sum().backward()
        out[f"gw_a{a}"] = gw
        for k, prm in net.named_parameters():
            out[f"grad_a{a}.{k}"] = gsum(prm.grad)
        out[f"grad_a{a}.latent"] = code_p.grad.clone()
    # fp64 run of the reference: pins the oracle's semantics free of fp32 roundoff amplification
    net64 = nvp.DeformNetwork(d_feature=128, d_in=3, d_out_1=1, d_out_2=3, n_blocks=3, d_hidden=128, n_layers=1,
                              skip_in=[], multires=6, weight_norm=True, actfn="softplus")
    set_params(net64, wp)
    net64 = net64.double()
    for a in (0.3, 1.0):
        c64 = torch.nn.Parameter(code.double())
        y = net64.forward(c64, pts.double(), alpha_ratio=a)
        (y * out[f"gw_a{a}"].double()).sum().backward()
        out[f"fwd64_a{a}"] = y.detach().numpy()
        out[f"grad64_a{a}.latent"] = c64.grad.numpy()
        out[f"grad64_a{a}.lin1_b_0.weight_v"] = net64.lin1_b_0.weight_v.grad.numpy()[::8, ::5]
        out[f"grad64_a{a}.lin2_a_0.weight_g"] = net64.lin2_a_0.weight_g.grad.numpy()
        net64.zero_grad()
    # identity-at-init known answer (reference zero-init of the last layers, nvp_ndr.py:275-277)
    net0 = nvp.DeformNetwork(d_feature=128, d_in=3, d_out_1=1, d_out_2=3, n_blocks=3, d_hidden=128, n_layers=1,
                             skip_in=[], multires=6, weight_norm=True, actfn="softplus")
    out["identity_max_abs"] = (net0.forward(code, pts, 0.3) - pts).abs().max()
    save("warp", **out)

    # ---------------------------------------------------------------- M2 NeRF MLP, P1 PE (+c2f)
    np_ = O.make_nerf_params(seed=21)
    points = torch.from_numpy(rng.uniform(-2, 2, (2, 5, 8, 3)).astype(np.float32))
    points[0, 0, 0] *= 3e4          # large-argument trig (inverse-depth far samples)
    dirs = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((2, 5, 1, 3)).astype(np.float32)), dim=-1).expand(2, 5, 8, 3).contiguous()
    out = dict(points=points, dirs=dirs, nerf_seed=21)
    o1 = load_opt("nerf_llff_repr", "nerf", **{"data.image_size": [H, W]})
    n1 = ref_nerf.NeRF(o1)
    set_params(n1, np_)
    rgb, den = n1.forward(o1, points, ray_unit=dirs, mode=None)
    out["relu_rgb"], out["relu_density"] = rgb, den
    out["pe_L10"] = n1.positional_encoding(o1, points, L=10)
    o3 = load_opt("barf_inn_llff", "barf_inn_llff", **{"data.image_size": [H, W], "barf_c2f": [0.1, 0.5]})
    n3 = ref_barf.NeRF(o3)
    set_params(n3, np_)
    for prog in (0.0, 0.22, 0.8):
        n3.progress.data.fill_(prog)
        rgb, den = n3.forward(o3, points, ray_unit=dirs, mode="train")
        out[f"c2f{prog}_rgb"], out[f"c2f{prog}_density"] = rgb, den
        out[f"c2f{prog}_pe10"] = n3.positional_encoding(o3, points, L=10)
    # gradients of the MLP (softplus density, c2f 0.22) w.r.t. parameters and inputs
    n3.progress.data.fill_(0.22)
    pts_g = points.clone().requires_grad_(True)
    dir_g = dirs.clone().requires_grad_(True)
    rgb, den = n3.forward(o3, pts_g, ray_unit=dir_g, mode="train")
    g_rgb = torch.from_numpy(np.random.default_rng(6).standard_normal(tuple(rgb.shape)).astype(np.float32))
    g_den = torch.from_numpy(np.random.default_rng(7).standard_normal(tuple(den.shape)).astype(np.float32))
    ((rgb * g_rgb).sum() + (den * g_den).sum()).backward()
    out["g_rgb"], out["g_den"] = g_rgb, g_den
    out["grad_points"], out["grad_dirs"] = pts_g.grad, dir_g.grad
    for k, prm in n3.named_parameters():
        if prm.grad is not None:
            out[f"grad.{k}"] = gsum(prm.grad)
    save("nerf_mlp", **out)

    # ---------------------------------------------------------------- C1 composite (+ grads)
    Bc, Rc, Sc = 2, 6, 16
    ray = torch.from_numpy(rng.standard_normal((Bc, Rc, 3)).astype(np.float32)).requires_grad_(True)
    rgb_s = torch.from_numpy(rng.uniform(0, 1, (Bc, Rc, Sc, 3)).astype(np.float32)).requires_grad_(True)
    sig_s = torch.from_numpy((rng.uniform(0, 4, (Bc, Rc, Sc)) * (rng.uniform(0, 1, (Bc, Rc, Sc)) > 0.3)).astype(np.float32)).requires_grad_(True)
    d_s = torch.from_numpy(np.sort(rng.uniform(0.5, 6, (Bc, Rc, Sc, 1)), axis=2).astype(np.float32))
    o1.nerf.setbg_opaque = None
    rgb, dep, opa, prob = n1.composite(o1, ray, rgb_s, sig_s, d_s)
    gs = [torch.from_numpy(np.random.default_rng(30 + i).standard_normal(tuple(t.shape)).astype(np.float32)) for i, t in enumerate((rgb, dep, opa, prob))]
    sum((t * g).sum() for t, g in zip((rgb, dep, opa, prob), gs)).backward()
    save("composite", ray=ray, rgb_s=rgb_s, sig_s=sig_s, depth_s=d_s, rgb=rgb, depth=dep, opacity=opa, prob=prob,
         g_rgb=gs[0], g_depth=gs[1], g_opacity=gs[2], g_prob=gs[3],
         grad_ray=ray.grad, grad_rgb_s=rgb_s.grad, grad_sig_s=sig_s.grad)

    # ---------------------------------------------------------------- S1 / H1 / H2 sampling
    g1g = ref_nerf.Graph(o1)        # metric [0,1], S=64, Sf=128
    o1.nerf.sample_intvs, o1.nerf.sample_intvs_fine = 16, 32
    u = torch.from_numpy(rng.uniform(0, 1, (2, 5, 16, 1)).astype(np.float32))
    _rand = torch.rand
    torch.rand = lambda *a, **k: u.clone()
    d_metric = g1g.sample_depth(o1, 2, num_rays=5)
    o2 = load_opt("nerf_inn_llff", "nerf", **{"data.image_size": [H, W]})
    o2.nerf.sample_intvs, o2.nerf.sample_intvs_fine, o2.nerf.fine_sampling = 16, 32, False
    g2g = ref_nerf.Graph(o2)
    d_inverse = g2g.sample_depth(o2, 2, num_rays=5)
    torch.rand = _rand
    pdf = torch.from_numpy((rng.uniform(0, 0.08, (2, 5, 16)) * (rng.uniform(0, 1, (2, 5, 16)) > 0.4)).astype(np.float32))
    pdf[0, 0] = 0.0                       # all-zero pdf -> every quantile clamps to the far bound
    pdf[0, 1] = 1.0 / 16                  # uniform pdf known answer
    pdf[1, 0] *= 0.3 / float(pdf[1, 0].sum() + 1e-9)
    f_metric = g1g.sample_depth_from_pdf(o1, pdf)
    f_inverse = g2g.sample_depth_from_pdf(o2, pdf)
    m_metric = torch.cat([d_metric, f_metric], dim=2).sort(dim=2).values
    m_inverse = torch.cat([d_inverse, f_inverse], dim=2).sort(dim=2).values
    save("sampling", u=u, pdf=pdf, depth_metric=d_metric, depth_inverse=d_inverse, fine_metric=f_metric,
         fine_inverse=f_inverse, merged_metric=m_metric, merged_inverse=m_inverse)

    # ---------------------------------------------------------------- G1 render (cfg-1 like: relu, fine net, GT poses)
    Rr, S, Sf = 8, 16, 32
    o1 = load_opt("nerf_llff_repr", "nerf", **{"data.image_size": [H, W]})
    o1.nerf.sample_intvs, o1.nerf.sample_intvs_fine, o1.nerf.density_noise_reg = S, Sf, None
    g = ref_nerf.Graph(o1)
    pc, pf = O.make_nerf_params(seed=41), O.make_nerf_params(seed=42)
    set_params(g.nerf, pc)
    set_params(g.nerf_fine, pf)
    ray_idx = torch.from_numpy(rng.permutation(H * W)[:Rr].astype(np.int64))
    u = torch.from_numpy(rng.uniform(0, 1, (B, Rr, S, 1)).astype(np.float32))
    pose_r = rand_pose(rng, B, rot=0.1, trans=0.1)
    torch.rand = lambda *a, **k: u.clone()
    ret = g.render(o1, pose_r, intr=intr, ray_idx=ray_idx, mode="train")
    torch.rand = _rand
    image = torch.from_numpy(rng.uniform(0, 1, (B, 3, H, W)).astype(np.float32))
    var = edict(idx=torch.arange(B), image=image, ray_idx=ray_idx)
    var.update(ret)
    loss = g.compute_loss(o1, var, mode="train")
    (loss.render + loss.render_fine).backward()
    out = dict(H=H, W=W, S=S, Sf=Sf, intr=intr, pose=pose_r, ray_idx=ray_idx, u=u, image=image, seed_coarse=41, seed_fine=42,
               rgb=ret.rgb, depth=ret.depth, opacity=ret.opacity, rgb_fine=ret.rgb_fine, depth_fine=ret.depth_fine,
               opacity_fine=ret.opacity_fine, loss_render=loss.render, loss_render_fine=loss.render_fine)
    for k, prm in g.named_parameters():
        out[f"grad.{k}"] = gsum(prm.grad)
    save("render_cfg1", **out)

    # ---------------------------------------------------------------- G0 INN train step (cfg-3 like) + hierarchical (cfg-2 like)
    for tag, fine in (("cfg3", False), ("cfg2", True)):
        o3 = load_opt("barf_inn_llff", "barf_inn_llff", **{"data.image_size": [H, W], "barf_c2f": [0.1, 0.5]})
        o3.nerf.sample_intvs, o3.nerf.rand_rays = S, Rr * B
        if fine:
            o3.nerf.fine_sampling, o3.nerf.sample_intvs_fine = True, Sf
            o3.loss_weight.render_fine = 0
        o3.max_iter = 200000
        g = ref_barf.Graph(o3)
        pc, pf = O.make_nerf_params(seed=51), O.make_nerf_params(seed=52)
        set_params(g.nerf, pc)
        if fine:
            set_params(g.nerf_fine, pf)
        g.warp_latent = torch.nn.Embedding(B, 128)
        with torch.no_grad():
            g.warp_latent.weight.copy_(O.make_latent(53, B))
        g.warp_mlp = nvp.DeformNetwork(d_feature=128, d_in=3, d_out_1=1, d_out_2=3, n_blocks=3, d_hidden=128,
                                       n_layers=1, skip_in=[], multires=6, weight_norm=True, actfn="softplus")
        set_params(g.warp_mlp, O.make_warp_params(seed=54, perturb=0.02))
        g.global_rigid = torch.nn.Embedding(B, 12, _weight=torch.eye(3, 4).repeat(B, 1, 1).view(-1, 12))
        it = 30000
        prog = 0.22
        g.nerf.progress.data.fill_(prog)
        if fine:
            g.nerf_fine.progress.data.fill_(prog)
        u = torch.from_numpy(rng.uniform(0, 1, (B, Rr, S, 1)).astype(np.float32))
        ray_idx = torch.from_numpy(rng.permutation(H * W).astype(np.int64))
        image = torch.from_numpy(rng.uniform(0, 1, (B, 3, H, W)).astype(np.float32))
        var = edict(idx=torch.arange(B), image=image, intr=intr, pose=pose_r)
        _perm = torch.randperm
        torch.rand = lambda *a, **k: u.clone()
        torch.randperm = lambda *a, **k: ray_idx.clone()
        var = g.forward(o3, var, mode="train", iter=it)
        torch.rand, torch.randperm = _rand, _perm
        loss = g.compute_loss(o3, var, mode="train")
        total = loss.render + (loss.render_fine if fine else 0)
        total.backward()
        out = dict(H=H, W=W, S=S, Sf=Sf if fine else 0, R=Rr, it=it, progress=prog, max_pe_iter=o3.inn.real_nvp.max_pe_iter,
                   intr=intr, ray_idx=var.ray_idx, u=u, image=image, seed_coarse=51, seed_fine=52, seed_latent=53,
                   seed_warp=54, warp_perturb=0.02, alpha_ratio=var.inn_posenc_alpha,
                   ray=var.grid_3D - var.center, center=var.center, grid_3D=var.grid_3D, grid_cam=var.grid_cam,
                   rgb=var.rgb, depth=var.depth, opacity=var.opacity, loss_render=loss.render)
        if fine:
            out.update(rgb_fine=var.rgb_fine, depth_fine=var.depth_fine, opacity_fine=var.opacity_fine,
                       loss_render_fine=loss.render_fine)
        for k, prm in g.named_parameters():
            if prm.grad is not None:
                out[f"grad.{k}"] = gsum(prm.grad)
        save(f"inn_step_{tag}", **out)

    # ---------------------------------------------------------------- G0 DTU INN step (cfg-5 like): noisy initial poses,
    # data depth range, warp through INNPoseParams.  The reference's model/barf_inn_dtu.py module pulls the COLMAP /
    # PDC-Net tool chain in at import time, so its 3-line Graph.get_pose dispatch (barf_inn_dtu.py:535-545) is restated
    # on top of the reference's own nerf_inn_dtu.Graph; every number below is produced by reference functions
    # (INNPoseParams.get_warped_rays_in_world, nerf_inn_dtu.Graph.forward / render_local / compute_loss).
    import roma
    from oracle import niw_oracle as _O
    roma.rigid_points_registration = lambda x, y, *a, **k: _O.rigid_registration(x, y)   # only feeds the detached pose_global
    import model.nerf_inn_dtu as ref_dtu
    from model.pose_models.inn import INNPoseParams

    class _DtuGraph(ref_dtu.Graph):
        def __init__(self, opt, pose_net):
            super().__init__(opt)
            self.pose_net = pose_net

        def get_pose(self, opt, var, mode=None, iter=None):
            return self.pose_net.get_warped_rays_in_world(var, mode, iter)

    o5 = load_opt("barf_inn_dtu", "barf_inn_dtu", **{"data.image_size": [H, W]})
    o5.nerf.sample_intvs, o5.nerf.rand_rays = S, Rr * B
    o5.loss_weight.global_alignment = None
    pose_init = rand_pose(rng, B, rot=0.15, trans=0.15)
    pn = INNPoseParams(o5, num_poses=B, initial_poses_w2c=pose_init, device="cpu")
    set_params(pn.pose_embedding, O.make_warp_params(seed=64, perturb=0.02))
    with torch.no_grad():
        pn.pose_latent.weight.copy_(O.make_latent(63, B))
    g = _DtuGraph(o5, pn)
    set_params(g.nerf, O.make_nerf_params(seed=61))
    u = torch.from_numpy(rng.uniform(0, 1, (B, Rr, S, 1)).astype(np.float32))
    ray_idx = torch.from_numpy(rng.permutation(H * W)[:Rr].astype(np.int64))
    image = torch.from_numpy(rng.uniform(0, 1, (B, 3, H, W)).astype(np.float32))
    depth_range = torch.tensor([[1.2, 5.2]] * B)
    var = edict(idx=torch.arange(B), image=image, intr=intr, pose=pose_init, depth_range=depth_range)
    _perm = torch.randperm
    torch.rand = lambda *a, **k: u.clone()
    torch.randperm = lambda *a, **k: ray_idx.clone()
    it = 45000
    var = g.forward(o5, var, mode="train", iter=it)
    torch.rand, torch.randperm = _rand, _perm
    loss = g.compute_loss(o5, var, mode="train")
    loss.render.backward()
    out = dict(H=H, W=W, S=S, R=Rr, it=it, max_pe_iter=o5.inn.real_nvp.max_pe_iter, intr=intr, pose_init=pose_init, ray_idx=ray_idx, u=u,
               image=image, depth_range=depth_range, seed_coarse=61, seed_latent=63, seed_warp=64, warp_perturb=0.02,
               center=var.center_local, grid_3D=var.grid_local, grid_init=var.grid_init, center_init=var.center_init,
               rgb=var.rgb, depth=var.depth, opacity=var.opacity, loss_render=loss.render)
    for k, prm in list(g.nerf.named_parameters()) + [("pose_embedding." + n, p) for n, p in pn.pose_embedding.named_parameters()] + \
            [("pose_latent.weight", pn.pose_latent.weight)]:
        if prm.grad is not None:
            out[f"grad.{k}"] = gsum(prm.grad)
    save("inn_step_cfg5", **out)


if __name__ == "__main__":
    main()
