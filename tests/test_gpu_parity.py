"""Parity of the HIP path (through the C ABI and the reference-style classes) against the CPU
oracle on identical seeded inputs and against the golden vectors captured from the reference.
Stated tolerances (exact-fp32 MFMA mode): rgb / opacity / sigma atol 2e-5, rtol 1e-4;
gradients rtol 2e-3 of the tensor's max; quantities passing through the 2^9*pi encoding or the
1e10 closing interval are compared relative to their own scale.  Needs a GPU."""
import numpy as np
import pytest
import torch

from oracle import niw_oracle as O
from tests.util import golden, t, check_grad_summary, check_grad_vs_fp64

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def g(x):
    return x.to(DEV) if isinstance(x, torch.Tensor) else torch.as_tensor(x).to(DEV)


def close(a, b, atol=2e-5, rtol=1e-4):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu() if isinstance(b, torch.Tensor) else t(b), atol=atol, rtol=rtol)


def relclose(a, b, rel):
    a, b = a.detach().float().cpu(), (b.detach().float().cpu() if isinstance(b, torch.Tensor) else t(b))
    assert (a - b).abs().max() <= rel * max(float(b.abs().max()), 1e-30), f"max err {(a - b).abs().max():.3e} vs scale {b.abs().max():.3e}"


def load_nerf(module, params):
    sd = module.state_dict()
    with torch.no_grad():
        for k, v in params.items():
            sd[k].copy_(v)


def mk_opt(fn, **kw):
    from neural_invertible_warp_amd import configs
    opt = getattr(configs, fn)(device=DEV)
    for k, v in kw.items():
        node = opt
        ks = k.split(".")
        for kk in ks[:-1]:
            node = node[kk]
        node[ks[-1]] = v
    return opt


def test_library_loads():
    from neural_invertible_warp_amd import _lib
    assert _lib.load().niw_version() >= 100


def test_raygen_golden():
    from neural_invertible_warp_amd import ops
    gd = golden("raygen")
    H, W = int(gd["H"]), int(gd["W"])
    intr, pose, idx = g(t(gd["intr"])), g(t(gd["pose"])), g(torch.from_numpy(gd["ray_idx"]))
    c, gr = ops.raygen(intr, None, idx, H, W, 0)
    close(c, gd["center_unwarped"], atol=2e-6); close(gr, gd["grid_unwarped"], atol=2e-6)
    c, gr = ops.raygen(intr, pose, idx, H, W, 0)
    close(c, gd["center_unwarped_pose"], atol=2e-6); close(gr, gd["grid_unwarped_pose"], atol=2e-6)
    c, r = ops.raygen(intr, pose, None, H, W, 1)
    close(c, gd["center"], atol=2e-6); close(r, gd["ray"], atol=2e-6)
    cn, rn = ops.convert_ndc((c[:, idx] + g(torch.tensor([0., 0., 3.]))).contiguous(), (r[:, idx] + g(torch.tensor([0., 0., 2.]))).contiguous(), intr)
    close(cn, gd["ndc_center"], atol=1e-5); close(rn, gd["ndc_ray"], atol=1e-5)


def test_sampling_golden():
    from neural_invertible_warp_amd import ops
    gd = golden("sampling")
    u, pdf = g(t(gd["u"])), g(t(gd["pdf"]))
    dm = ops.sample_stratified(u.view(-1, 16), 10, 16, (0, 1), "metric", DEV)
    di = ops.sample_stratified(u.view(-1, 16), 10, 16, (1, 0), "inverse", DEV)
    assert torch.equal(dm.cpu().view(2, 5, 16, 1), t(gd["depth_metric"]))          # bit exact
    assert torch.equal(di.cpu().view(2, 5, 16, 1), t(gd["depth_inverse"]))        # bit exact (separately rounded ops, niw_common.h)
    fm, mm = ops.sample_pdf_merge(pdf.view(10, 16), dm, 32, (0, 1))
    fi, mi = ops.sample_pdf_merge(pdf.view(10, 16), di, 32, (1, 0))
    close(fm.view(2, 5, 32, 1), gd["fine_metric"], atol=1e-6); close(fi.view(2, 5, 32, 1), gd["fine_inverse"], atol=1e-6)
    close(mm.view(2, 5, 48, 1), gd["merged_metric"], atol=1e-6); close(mi.view(2, 5, 48, 1), gd["merged_inverse"], atol=1e-6, rtol=1e-6)
    assert torch.all(mm[:, 1:] >= mm[:, :-1])                                      # sortedness


def test_sampling_mid_points_when_not_stratified():
    from neural_invertible_warp_amd import ops
    d = ops.sample_stratified(None, 3, 8, (2, 6), "metric", DEV)
    close(d, O.sample_depth(0.5, 8, (2, 6), "metric")[0, :1, :, 0].expand(3, 8), atol=1e-6)


# up to 256 samples: 16 lanes per ray with 1 .. 4 quads per lane (round 5); 4, 8, 36, 68, 100, 132, 196, 252 leave a lane's later quads or whole
# lanes empty, 256 fills four quads, 260 / 320 are the first sizes on the 64-lane chunked form
@pytest.mark.parametrize("S", [4, 8, 16, 36, 64, 68, 100, 128, 132, 192, 196, 200, 252, 256, 260, 320])
def test_composite_vs_oracle(S):
    from neural_invertible_warp_amd import ops
    rng = np.random.default_rng(S)
    N = 37
    ray = t(rng.standard_normal((1, N, 3))).requires_grad_(True)
    rgb_s = t(rng.uniform(0, 1, (1, N, S, 3))).requires_grad_(True)
    sig = t(rng.uniform(0, 3, (1, N, S)) * (rng.uniform(0, 1, (1, N, S)) > 0.3)).requires_grad_(True)
    dep = t(np.sort(rng.uniform(0.5, 6, (1, N, S, 1)), axis=2))
    ref = O.composite(ray, rgb_s, sig, dep)
    gs = [t(rng.standard_normal(tuple(x.shape))) for x in ref]
    sum((a * b).sum() for a, b in zip(ref, gs)).backward()
    r2, c2, s2 = g(ray.detach()[0]).requires_grad_(True), g(rgb_s.detach()[0]).requires_grad_(True), g(sig.detach()[0]).requires_grad_(True)
    out = ops.composite(r2, c2, s2, g(dep[0, :, :, 0]))
    close(out[0], ref[0][0]); close(out[1], ref[1][0, :, 0], atol=1e-4); close(out[2], ref[2][0, :, 0]); close(out[3], ref[3][0, :, :, 0])
    (out[0] * g(gs[0][0])).sum().add((out[1] * g(gs[1][0, :, 0])).sum()).add((out[2] * g(gs[2][0, :, 0])).sum()).add((out[3] * g(gs[3][0, :, :, 0])).sum()).backward()
    relclose(r2.grad, ray.grad[0], 2e-4); relclose(c2.grad, rgb_s.grad[0], 1e-5); relclose(s2.grad, sig.grad[0], 2e-4)


def test_composite_golden_and_properties():
    from neural_invertible_warp_amd import ops
    gd = golden("composite")
    B, R, S = gd["sig_s"].shape
    out = ops.composite(g(t(gd["ray"]).view(-1, 3)), g(t(gd["rgb_s"]).view(B * R, S, 3)), g(t(gd["sig_s"]).view(B * R, S)), g(t(gd["depth_s"]).view(B * R, S)))
    close(out[0].view(B, R, 3), gd["rgb"]); close(out[1].view(B, R, 1), gd["depth"], atol=1e-5)
    close(out[2].view(B, R, 1), gd["opacity"]); close(out[3].view(B, R, S, 1), gd["prob"])
    close(out[3].sum(-1), out[2], atol=1e-6)            # weights sum to the opacity


def _mlp_inputs(rng, N, S, big=False):
    center = t(rng.uniform(-1, 1, (N, 3)))
    ray = t(rng.standard_normal((N, 3)))
    depth = t(np.sort(rng.uniform(0.5, 4, (N, S)), axis=1))
    if big:
        depth[0, -1] = 3e4
    return center, ray, depth


@pytest.mark.parametrize("activ,S", [("softplus", 32), ("relu", 64), ("softplus", 24)])
def test_mlp_forward_vs_oracle(activ, S):
    from neural_invertible_warp_amd import ops
    rng = np.random.default_rng(7)
    N = 13
    p = O.make_nerf_params(5)
    center, ray, depth = _mlp_inputs(rng, N, S, big=True)
    w3, wv = O.c2f_weights(0.22, (0.1, 0.5), 10), O.c2f_weights(0.22, (0.1, 0.5), 4)
    rgb_ref, sig_ref = O.forward_samples(p, center[None], ray[None], depth[None, :, :, None], density_activ=activ, w3d=w3, wview=wv)
    flat = torch.cat([p[f"{n}.{k}"].reshape(-1) for n, _, _ in O.nerf_layer_shapes() for k in ("weight", "bias")]).to(DEV)
    st = ops.FieldState(flat)
    with torch.no_grad():
        rgb, sig = ops.field_mlp(st, [], g(center), g(ray), g(depth), w3.tolist(), wv.tolist(), activ)
    close(rgb, rgb_ref[0]); close(sig, sig_ref[0], atol=5e-5, rtol=2e-4)


def test_mlp_golden_through_nerf_forward():
    from neural_invertible_warp_amd.model import barf_inn_llff, nerf
    gd = golden("nerf_mlp")
    p = O.make_nerf_params(int(gd["nerf_seed"]))
    pts, dirs = g(t(gd["points"])), g(t(gd["dirs"]))
    o1 = mk_opt("cfg1_nerf_llff_repr")
    n1 = nerf.NeRF(o1)
    load_nerf(n1, p)
    with torch.no_grad():
        rgb, den = n1.forward(o1, pts, ray_unit=dirs)
    close(rgb, gd["relu_rgb"]); close(den, gd["relu_density"], atol=5e-5, rtol=2e-4)
    o3 = mk_opt("cfg3_barf_inn_llff")
    n3 = barf_inn_llff.NeRF(o3)
    load_nerf(n3, p)
    for prog in (0.0, 0.22, 0.8):
        n3.set_progress(prog)
        with torch.no_grad():
            rgb, den = n3.forward(o3, pts, ray_unit=dirs, mode="train")
        close(rgb, gd[f"c2f{prog}_rgb"]); close(den, gd[f"c2f{prog}_density"], atol=5e-5, rtol=2e-4)


@pytest.mark.parametrize("S,activ", [(32, "softplus"), (40, "relu")])
def test_mlp_backward_vs_oracle(S, activ):
    from neural_invertible_warp_amd import ops
    rng = np.random.default_rng(11)
    N = 9
    p = {k: v.requires_grad_(True) for k, v in O.make_nerf_params(6).items()}
    center, ray, depth = _mlp_inputs(rng, N, S)
    center.requires_grad_(True); ray.requires_grad_(True)
    w3, wv = O.c2f_weights(0.3, (0.1, 0.5), 10), O.c2f_weights(0.3, (0.1, 0.5), 4)
    rgb_ref, sig_ref = O.forward_samples(p, center[None], ray[None], depth[None, :, :, None], density_activ=activ, w3d=w3, wview=wv)
    g_rgb, g_sig = t(rng.standard_normal((N, S, 3))), t(rng.standard_normal((N, S)))
    ((rgb_ref[0] * g_rgb).sum() + (sig_ref[0] * g_sig).sum()).backward()
    names = [f"{n}.{k}" for n, _, _ in O.nerf_layer_shapes() for k in ("weight", "bias")]
    flat = torch.cat([p[n].detach().reshape(-1) for n in names]).to(DEV)
    st = ops.FieldState(flat)
    params, off = [], 0
    for n in names:
        params.append(flat[off:off + p[n].numel()].view(p[n].shape).requires_grad_(True))
        off += p[n].numel()
    c2, r2 = g(center.detach()).requires_grad_(True), g(ray.detach()).requires_grad_(True)
    rgb, sig = ops.field_mlp(st, params, c2, r2, g(depth), w3.tolist(), wv.tolist(), activ)
    close(rgb, rgb_ref[0]); close(sig, sig_ref[0], atol=5e-5, rtol=2e-4)
    ((rgb * g(g_rgb)).sum() + (sig * g(g_sig)).sum()).backward()
    for n, prm in zip(names, params):
        relclose(prm.grad, p[n].grad, 2e-3)
    relclose(c2.grad, center.grad, 2e-3); relclose(r2.grad, ray.grad, 2e-3)


def test_mlp_weight_gradient_at_a_batch_that_takes_the_vector_head_pieces():
    """From 131,072 samples the exact-mode weight gradient forms the density row and the colour rows in dw_heads_kernel (vector ALU, second
    stream beside the wide MFMA launch) instead of as two pieces of the skinny MFMA launch.  1100 rays x 128 samples = 140,800 samples
    take that path; comparator = the oracle in FLOAT64 on the GPU (exact to the digits shown), so the bound is the fp32 kernels' own
    rounding over a 140 k-term sum: 2e-4 of each tensor's max."""
    from neural_invertible_warp_amd import ops
    rng = np.random.default_rng(23)
    N, S, activ = 1100, 128, "softplus"
    p32 = O.make_nerf_params(8)
    p = {k: v.double().to(DEV).requires_grad_(True) for k, v in p32.items()}
    center, ray, depth = _mlp_inputs(rng, N, S)
    w3, wv = O.c2f_weights(0.3, (0.1, 0.5), 10), O.c2f_weights(0.3, (0.1, 0.5), 4)
    rgb_ref, sig_ref = O.forward_samples(p, g(center).double()[None], g(ray).double()[None], g(depth).double()[None, :, :, None], density_activ=activ,
                                         w3d=w3.double().to(DEV), wview=wv.double().to(DEV))
    g_rgb, g_sig = g(t(rng.standard_normal((N, S, 3)))), g(t(rng.standard_normal((N, S))))
    ((rgb_ref[0] * g_rgb.double()).sum() + (sig_ref[0] * g_sig.double()).sum()).backward()
    names = [f"{n}.{k}" for n, _, _ in O.nerf_layer_shapes() for k in ("weight", "bias")]
    flat = torch.cat([p32[n].reshape(-1) for n in names]).to(DEV)
    st = ops.FieldState(flat)
    params, off = [], 0
    for n in names:
        params.append(flat[off:off + p32[n].numel()].view(p32[n].shape).requires_grad_(True))
        off += p32[n].numel()
    rgb, sig = ops.field_mlp(st, params, g(center), g(ray), g(depth), w3.tolist(), wv.tolist(), activ)
    close(rgb, rgb_ref[0]); close(sig, sig_ref[0], atol=5e-5, rtol=2e-4)
    ((rgb * g_rgb).sum() + (sig * g_sig).sum()).backward()
    worst = {}
    for n, prm in zip(names, params):
        ref = p[n].grad.float()
        worst[n] = float((prm.grad - ref).abs().max() / ref.abs().max())
    grads = dict(zip(names, params))
    # what the heads kernel owns: the colour rows with their biases, and the density row (row 0 of the eighth layer: nerf.py:427) with its bias.  Their
    # operands (h7, hr, d sigma, d rgb_raw) are forward values and head gradients: no ReLU mask between them and the float64 reference.
    w7, b7 = grads["mlp_feat.7.weight"].grad, grads["mlp_feat.7.bias"].grad
    r7, rb7 = p["mlp_feat.7.weight"].grad.float(), p["mlp_feat.7.bias"].grad.float()
    heads = {"mlp_rgb.1.weight": worst["mlp_rgb.1.weight"], "mlp_rgb.1.bias": worst["mlp_rgb.1.bias"],
             "density row": float((w7[0] - r7[0]).abs().max() / r7[0].abs().max()),
             "density bias": float((b7[0] - rb7[0]).abs() / rb7[0].abs())}
    print("heads kernel, worst relative deviation:", {k: f"{v:.1e}" for k, v in heads.items()})
    print("all tensors:", {k: f"{v:.1e}" for k, v in worst.items()})
    assert max(heads.values()) <= 5e-5, heads
    # the trunk's gradients pass eight ReLU masks evaluated in fp32 (a pre-activation within rounding of zero flips its mask against the
    # float64 reference): conditioning, bounded like tests/test_gpu_baseline_shapes.py bounds it at full shapes
    assert max(worst.values()) <= 2e-2, worst


def test_mlp_backward_with_exactly_zero_preactivations():
    """ReLU'(0) = 0, as in torch (reference model/nerf.py:422-447 uses torch_F.relu): a layer whose weight and bias are zero has the
    pre-activation +0.0 on every sample; no gradient may pass it.  The kernels record (activation > 0) -- not the sign bit of the
    pre-activation, which would let +0.0 through -- so every gradient upstream of the dead layer is EXACTLY zero, the dead layer's
    own weight and bias gradients are zero, and the layers behind it (fed by the skip connection) match the oracle."""
    from neural_invertible_warp_amd import ops
    rng = np.random.default_rng(5)
    N, S = 6, 16
    p = O.make_nerf_params(9)
    p["mlp_feat.2.weight"].zero_(); p["mlp_feat.2.bias"].zero_()
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    center, ray, depth = _mlp_inputs(rng, N, S)
    rgb_ref, sig_ref = O.forward_samples(p, center[None], ray[None], depth[None, :, :, None], density_activ="softplus")
    g_rgb, g_sig = t(rng.standard_normal((N, S, 3))), t(rng.standard_normal((N, S)))
    ((rgb_ref[0] * g_rgb).sum() + (sig_ref[0] * g_sig).sum()).backward()
    names = [f"{n}.{k}" for n, _, _ in O.nerf_layer_shapes() for k in ("weight", "bias")]
    flat = torch.cat([p[n].detach().reshape(-1) for n in names]).to(DEV)
    st = ops.FieldState(flat)
    params, off = [], 0
    for n in names:
        params.append(flat[off:off + p[n].numel()].view(p[n].shape).requires_grad_(True))
        off += p[n].numel()
    rgb, sig = ops.field_mlp(st, params, g(center), g(ray), g(depth), [1.0] * 10, [1.0] * 4, "softplus")
    close(rgb, rgb_ref[0]); close(sig, sig_ref[0], atol=5e-5, rtol=2e-4)
    ((rgb * g(g_rgb)).sum() + (sig * g(g_sig)).sum()).backward()
    for n, prm in zip(names, params):
        layer = int(n.split(".")[1]) if n.startswith("mlp_feat") else 99
        if layer <= 2:
            assert float(p[n].grad.abs().max()) == 0.0                      # the reference's own gradient is zero here ...
            assert float(prm.grad.abs().max()) == 0.0, n                    # ... and so is ours, exactly
        elif layer == 3:
            # input of layer 3 is the dead layer's all-zero output: zero weight gradient, live bias gradient
            if n.endswith("weight"):
                assert float(prm.grad.abs().max()) == 0.0, n
            else:
                relclose(prm.grad, p[n].grad, 2e-3)
        else:
            relclose(prm.grad, p[n].grad, 2e-3)


@pytest.mark.parametrize("alpha,exact", [(0.3, True), (1.0, True), (0.45, False)])
def test_warp_vs_oracle_and_golden(alpha, exact):
    from neural_invertible_warp_amd.model.nvp import nvp_ndr
    gd = golden("warp")
    wp = O.make_warp_params(int(gd["warp_seed"]), float(gd["warp_perturb"]))
    net = nvp_ndr.DeformNetwork(d_feature=128, d_in=3, d_out_1=1, d_out_2=3, n_blocks=3, d_hidden=128, n_layers=1, skip_in=[],
                                multires=6, weight_norm=True, actfn="softplus", reference_exact=exact).to(DEV)
    load_nerf(net, wp)
    code = g(O.make_latent(int(gd["latent_seed"]), 3)).requires_grad_(True)
    pts = g(t(gd["pts"]))
    y = net.forward(code, pts, alpha_ratio=alpha)
    p64 = {k: v.double().requires_grad_(True) for k, v in wp.items()}
    c64 = O.make_latent(int(gd["latent_seed"]), 3).double().requires_grad_(True)
    y64 = O.warp_forward(p64, c64, t(gd["pts"]).double(), alpha, reference_exact=exact)
    close(y, y64.float(), atol=1e-4)                      # 2^5*pi embedding amplifies fp32 roundoff ~100x per block
    if exact:
        close(y, gd[f"fwd_a{alpha}"], atol=1e-4)
    close(net.inverse(code, y.detach(), alpha), pts, atol=2e-4)          # round trip
    # W4 pinned on its own (reference nvp_ndr.py:471-567), not only as the forward's mirror image:
    # (i) the reference's inverse of the reference's forward output, both in the fixture (the HIP kernel sees the reference's input, not its own)
    if exact:
        close(net.inverse(code, g(t(gd[f"fwd_a{alpha}"])), alpha), gd[f"inv_a{alpha}"], atol=2e-4)
    # (ii) a point set that is NOT the image of the forward, against the float64 oracle inverse
    gen = torch.Generator().manual_seed(77)
    arb = (torch.rand(tuple(pts.shape), generator=gen) * 2.4 - 1.2)
    inv64 = O.warp_inverse({k: v.detach() for k, v in p64.items()}, c64.detach(), arb.double(), alpha, reference_exact=exact)
    close(net.inverse(code, g(arb), alpha), inv64.float(), atol=2e-4)
    gw = t(gd["gw_a0.3"])
    (y * g(gw)).sum().backward()
    (y64 * gw.double()).sum().backward()
    for k, prm in net.named_parameters():
        relclose(prm.grad, p64[k].grad.float(), 1e-2)
    relclose(code.grad, c64.grad.float(), 1e-2)


def test_warp_identity_at_reference_init():
    from neural_invertible_warp_amd.model.nvp import nvp_ndr
    net = nvp_ndr.DeformNetwork(d_feature=128, d_in=3, d_out_1=1, d_out_2=3, n_blocks=3, d_hidden=128, n_layers=1, skip_in=[],
                                multires=6, weight_norm=True, actfn="softplus").to(DEV)
    pts = torch.randn(4, 50, 1, 3, device=DEV)
    assert (net.forward(torch.randn(4, 128, device=DEV), pts, 0.3) - pts).abs().max() == 0


def _capture_rng(u, ray_idx):
    import contextlib

    @contextlib.contextmanager
    def cm():
        r, p = torch.rand, torch.randperm
        torch.rand = lambda *a, **k: u.clone()
        torch.randperm = lambda *a, **k: ray_idx.clone()
        try:
            yield
        finally:
            torch.rand, torch.randperm = r, p
    return cm()


def test_render_cfg1_golden():
    from neural_invertible_warp_amd.model import nerf
    from neural_invertible_warp_amd.util import edict
    gd = golden("render_cfg1")
    H, W, S, Sf = (int(gd[k]) for k in ("H", "W", "S", "Sf"))
    opt = mk_opt("cfg1_nerf_llff_repr", H=H, W=W, **{"nerf.sample_intvs": S, "nerf.sample_intvs_fine": Sf, "nerf.density_noise_reg": None})
    graph = nerf.Graph(opt)
    load_nerf(graph.nerf, O.make_nerf_params(int(gd["seed_coarse"])))
    load_nerf(graph.nerf_fine, O.make_nerf_params(int(gd["seed_fine"])))
    idx = g(torch.from_numpy(gd["ray_idx"]))
    with _capture_rng(g(t(gd["u"])), idx):
        ret = graph.render(opt, g(t(gd["pose"])), intr=g(t(gd["intr"])), ray_idx=idx, mode="train")
    # end to end from poses, all ten encoding bands on: holds at the MLP tolerance because sample positions are
    # formed with the reference's two roundings (niw_common.h mul_rn / add_rn); a fused multiply-add there is a 1-ulp
    # position difference that the 2^9*pi band multiplies by 1.6e3 (it cost a factor 20 in this comparison)
    for k in ("rgb", "depth", "opacity", "rgb_fine", "depth_fine", "opacity_fine"):
        close(ret[k], gd[k])
    var = edict(idx=torch.arange(3), image=g(t(gd["image"])), ray_idx=idx)
    var.update(ret)
    loss = graph.compute_loss(opt, var, mode="train")
    close(loss.render, gd["loss_render"], atol=1e-6); close(loss.render_fine, gd["loss_render_fine"], atol=1e-6)
    (loss.render + loss.render_fine).backward()
    for k, prm in graph.named_parameters():
        check_grad_summary(prm.grad, gd, f"grad.{k}", rtol=5e-3)


def test_inn_train_step_dtu_golden():
    """cfg-5 like step through barf_inn_dtu.Graph + INNPoseParams against the reference's golden vectors."""
    from neural_invertible_warp_amd.model import barf_inn_dtu
    from neural_invertible_warp_amd.model.pose_models.inn import INNPoseParams
    from neural_invertible_warp_amd.util import edict
    gd = golden("inn_step_cfg5")
    H, W, S, R = (int(gd[k]) for k in ("H", "W", "S", "R"))
    opt = mk_opt("cfg5_barf_inn_dtu", H=H, W=W, **{"nerf.sample_intvs": S, "nerf.rand_rays": R * 3, "barf_c2f": None})
    pose_net = INNPoseParams(opt, num_poses=3, initial_poses_w2c=g(t(gd["pose_init"])), device=DEV)
    load_nerf(pose_net.pose_embedding, O.make_warp_params(int(gd["seed_warp"]), float(gd["warp_perturb"])))
    with torch.no_grad():
        pose_net.pose_latent.weight.copy_(O.make_latent(int(gd["seed_latent"]), 3))
    graph = barf_inn_dtu.Graph(opt, pose_net)
    load_nerf(graph.nerf, O.make_nerf_params(int(gd["seed_coarse"])))
    var = edict(idx=torch.arange(3), image=g(t(gd["image"])), intr=g(t(gd["intr"])), pose=g(t(gd["pose_init"])),
                depth_range=g(t(gd["depth_range"])))
    with _capture_rng(g(t(gd["u"])), g(torch.from_numpy(gd["ray_idx"]))):
        var = graph.forward(opt, var, mode="train", iter=int(gd["it"]))
    close(var.center_init, gd["center_init"], atol=2e-6); close(var.grid_init, gd["grid_init"], atol=2e-6)
    close(var.center_local, gd["center"], atol=2e-5); close(var.grid_local, gd["grid_3D"], atol=2e-5)
    close(var.rgb, gd["rgb"]); close(var.opacity, gd["opacity"])
    relclose(var.depth, gd["depth"], 2e-4)
    loss = graph.compute_loss(opt, var, mode="train")
    close(loss.render, gd["loss_render"], atol=1e-6)
    assert float(loss.global_alignment.detach()) >= 0            # detached Kabsch pose (nerf_inn_dtu.py:410-414)
    loss.render.backward()
    # gradients against the reference's FLOAT64 gradients of this step (inn_step_cfg5_fp64.npz) under the conditioning bound of
    # tests/util.fp64_bound: all ten bands active, so any fp32 evaluation -- the reference's own included -- scatters at the percent level
    _check_dtu_grads_vs_fp64(graph, pose_net, "cfg5")


def test_inn_train_step_dtu_c2f_golden():
    """cfg 5 with the shipped --barf_c2f=[0.1,0.5] (tests/golden/make_golden_dtu.py): forward values at the LLFF tolerances and ALL
    gradients to 1e-2 of scale (the CPU oracle itself sits 4.5e-3 from the reference on mlp_feat.0.weight: metric depths, points
    3-8 units from the origin) -- the unmasked fixture above can only bound them to 15 %."""
    from neural_invertible_warp_amd.model import barf_inn_dtu
    from neural_invertible_warp_amd.model.pose_models.inn import INNPoseParams
    from neural_invertible_warp_amd.util import edict
    gd = golden("inn_step_cfg5_c2f")
    H, W, S, R = (int(gd[k]) for k in ("H", "W", "S", "R"))
    opt = mk_opt("cfg5_barf_inn_dtu", H=H, W=W, **{"nerf.sample_intvs": S, "nerf.rand_rays": R * 3, "barf_c2f": [0.1, 0.5]})
    pose_net = INNPoseParams(opt, num_poses=3, initial_poses_w2c=g(t(gd["pose_init"])), device=DEV)
    load_nerf(pose_net.pose_embedding, O.make_warp_params(int(gd["seed_warp"]), float(gd["warp_perturb"])))
    with torch.no_grad():
        pose_net.pose_latent.weight.copy_(O.make_latent(int(gd["seed_latent"]), 3))
    graph = barf_inn_dtu.Graph(opt, pose_net)
    load_nerf(graph.nerf, O.make_nerf_params(int(gd["seed_coarse"])))
    graph.nerf.set_progress(float(gd["progress"]))
    var = edict(idx=torch.arange(3), image=g(t(gd["image"])), intr=g(t(gd["intr"])), pose=g(t(gd["pose_init"])),
                depth_range=g(t(gd["depth_range"])))
    with _capture_rng(g(t(gd["u"])), g(torch.from_numpy(gd["ray_idx"]))):
        var = graph.forward(opt, var, mode="train", iter=int(gd["it"]))
    close(var.center_init, gd["center_init"], atol=2e-6); close(var.grid_init, gd["grid_init"], atol=2e-6)
    close(var.center_local, gd["center"], atol=2e-5); close(var.grid_local, gd["grid_3D"], atol=2e-5)
    close(var.rgb, gd["rgb"], atol=3e-5, rtol=2e-4); close(var.opacity, gd["opacity"], atol=3e-5, rtol=2e-4)
    relclose(var.depth, gd["depth"], 2e-4)
    loss = graph.compute_loss(opt, var, mode="train")
    close(loss.render, gd["loss_render"], atol=1e-6)
    loss.render.backward()
    # every gradient against the reference's FLOAT64 gradient (inn_step_cfg5_fp64.npz), bound 2.3 % of max = 16 x the median
    # deviation of the reference's own fp32 gradients from them (tests/util.fp64_bound); rounds 1-2 held the pose network to 15 % / 60 %
    # of ONE fp32 evaluation.  The NeRF tensors are additionally held to 1e-2 of the reference's fp32 sample, as before.
    for k, prm in graph.nerf.named_parameters():
        if f"grad.{k}.norm" in gd:
            check_grad_summary(prm.grad, gd, f"grad.{k}", rtol=1e-2)
    _check_dtu_grads_vs_fp64(graph, pose_net, "cfg5_c2f")


def _check_dtu_grads_vs_fp64(graph, pose_net, tag):
    fx = golden("inn_step_cfg5_fp64")
    rows = []
    for k, prm in graph.nerf.named_parameters():
        if f"{tag}.grad64.{k}.norm" in fx:
            rows.append((k,) + check_grad_vs_fp64(prm.grad, fx, tag, k))
    for k, prm in pose_net.pose_embedding.named_parameters():
        rows.append((k,) + check_grad_vs_fp64(prm.grad, fx, tag, f"pose_embedding.{k}"))
    rows.append(("pose_latent.weight",) + check_grad_vs_fp64(pose_net.pose_latent.weight.grad, fx, tag, "pose_latent.weight"))
    worst = max(rows, key=lambda r: r[1])
    errs = sorted(r[1] for r in rows)
    print(f"{tag}: HIP fp32 gradients vs the reference's float64 gradients over {len(rows)} tensors: median {errs[len(errs) // 2]:.2e}, "
          f"worst {worst[1]:.2e} ({worst[0]}) of max; bound {worst[2]:.2e}; measured worst / bound {worst[1] / worst[2]:.2f}")


@pytest.mark.parametrize("tag", ["cfg3", "cfg2"])
def test_inn_train_step_golden(tag):
    from neural_invertible_warp_amd.model import barf_inn_llff
    from neural_invertible_warp_amd.util import edict
    gd = golden(f"inn_step_{tag}")
    H, W, S, Sf, R = (int(gd[k]) for k in ("H", "W", "S", "Sf", "R"))
    fine = Sf > 0
    over = {"nerf.sample_intvs": S, "nerf.rand_rays": R * 3}
    if fine:
        over.update({"nerf.sample_intvs_fine": Sf})
    opt = mk_opt("cfg2_nerf_inn_llff_hier" if fine else "cfg3_barf_inn_llff", H=H, W=W, **over)
    opt.loss_weight.global_alignment = None
    graph = barf_inn_llff.Graph(opt).attach_warp(opt, 3)
    load_nerf(graph.nerf, O.make_nerf_params(int(gd["seed_coarse"])))
    if fine:
        load_nerf(graph.nerf_fine, O.make_nerf_params(int(gd["seed_fine"])))
        graph.nerf_fine.set_progress(float(gd["progress"]))
    graph.nerf.set_progress(float(gd["progress"]))
    load_nerf(graph.warp_mlp, O.make_warp_params(int(gd["seed_warp"]), float(gd["warp_perturb"])))
    with torch.no_grad():
        graph.warp_latent.weight.copy_(O.make_latent(int(gd["seed_latent"]), 3))
    var = edict(idx=torch.arange(3), image=g(t(gd["image"])), intr=g(t(gd["intr"])))
    with _capture_rng(g(t(gd["u"])), g(torch.from_numpy(gd["ray_idx"]))[:R] if False else g(torch.from_numpy(gd["ray_idx"]))):
        # the fixture stores the already truncated ray_idx the reference used
        var = graph.forward(opt, var, mode="train", iter=int(gd["it"]))
    for k in ("center", "grid_3D"):
        close(var[k], gd[k], atol=2e-5)
    close(var.rgb, gd["rgb"], atol=3e-5, rtol=2e-4); close(var.opacity, gd["opacity"], atol=3e-5, rtol=2e-4)
    relclose(var.depth, gd["depth"], 2e-4)
    loss = graph.compute_loss(opt, var, mode="train")
    close(loss.render, gd["loss_render"], atol=1e-6)
    total = loss.render
    if fine:
        close(var.rgb_fine, gd["rgb_fine"], atol=3e-5, rtol=2e-4)
        close(loss.render_fine, gd["loss_render_fine"], atol=1e-6)
        total = total + loss.render_fine
    total.backward()
    for k, prm in graph.named_parameters():
        if f"grad.{k}.norm" in gd:
            check_grad_summary(prm.grad, gd, f"grad.{k}", rtol=5e-3)
