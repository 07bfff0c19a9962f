"""Edge cases of the HIP path against the oracle: degenerate sizes (one ray, one sample, one point, one view), sizes
that are no multiple of any tile, the 64-view limit, empty / uniform pdfs, inverse-depth extremes, and the error
behaviour of the reference-style classes.  Same tolerances as test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from oracle import niw_oracle as O
from tests.test_gpu_parity import DEV, close, g, load_nerf, relclose
from tests.util import t

pytestmark = pytest.mark.gpu


def _field(seed):
    from neural_invertible_warp_amd import ops
    p = O.make_nerf_params(seed)
    names = [f"{n}.{k}" for n, _, _ in O.nerf_layer_shapes() for k in ("weight", "bias")]
    flat = torch.cat([p[n].reshape(-1) for n in names]).to(DEV)
    params, off = [], 0
    for n in names:
        params.append(flat[off:off + p[n].numel()].view(p[n].shape).requires_grad_(True))
        off += p[n].numel()
    return p, names, ops.FieldState(flat), params


@pytest.mark.parametrize("N,S", [(1, 1), (1, 33), (3, 5), (129, 1)])
def test_mlp_degenerate_shapes_forward_and_backward(N, S):
    from neural_invertible_warp_amd import ops
    rng = np.random.default_rng(N * 100 + S)
    p, names, st, params = _field(8)
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    center, ray = t(rng.uniform(-1, 1, (N, 3))).requires_grad_(True), t(rng.standard_normal((N, 3))).requires_grad_(True)
    depth = t(np.sort(rng.uniform(0.5, 4, (N, S)), axis=1))
    ones3, ones4 = [1.0] * 10, [1.0] * 4
    rgb_ref, sig_ref = O.forward_samples(pr, center[None], ray[None], depth[None, :, :, None], density_activ="softplus")
    (rgb_ref.sum() + 2 * sig_ref.sum()).backward()
    c2, r2 = g(center.detach()).requires_grad_(True), g(ray.detach()).requires_grad_(True)
    rgb, sig = ops.field_mlp(st, params, c2, r2, g(depth), ones3, ones4, "softplus")
    assert rgb.shape == (N, S, 3) and sig.shape == (N, S)
    close(rgb, rgb_ref[0]); close(sig, sig_ref[0], atol=5e-5, rtol=2e-4)
    (rgb.sum() + 2 * sig.sum()).backward()
    for n, prm in zip(names, params):
        relclose(prm.grad, pr[n].grad, 5e-3)
    relclose(c2.grad, center.grad, 5e-3); relclose(r2.grad, ray.grad, 5e-3)


def test_mlp_inverse_depth_extremes_stay_finite_and_match():
    """inverse-depth sampling reaches 1/(0+1e-8) = 1e8 (nerf.py:343): the band arguments are ~1e11 rad"""
    from neural_invertible_warp_amd import ops
    p, names, st, _ = _field(9)
    center, ray = torch.tensor([[0.1, -0.2, 0.3]]), torch.tensor([[0.3, 0.2, 0.9]])
    depth = torch.tensor([[1.0, 7.5, 3.2e3, 4.0e5, 1.0e8]])
    rgb_ref, sig_ref = O.forward_samples(p, center[None], ray[None], depth[None, :, :, None], density_activ="softplus")
    with torch.no_grad():
        rgb, sig = ops.field_mlp(st, [], g(center), g(ray), g(depth), [1.0] * 10, [1.0] * 4, "softplus")
    assert torch.isfinite(rgb).all() and torch.isfinite(sig).all()
    # up to 3.2e3 the fp32 argument x*2^9*pi is still resolved to ~0.1 rad or better on both sides
    close(rgb[:, :2], rgb_ref[0][:, :2]); close(sig[:, :2], sig_ref[0][:, :2], atol=5e-5, rtol=2e-4)


@pytest.mark.parametrize("bg", [None, 0.75])
def test_composite_single_sample_degenerates_like_the_reference(bg):
    """reference nerf.py:461-462: the closing 1e10 interval is empty_like(intervals[..., :1]); with S = 1 that slice is empty, the
    sample gets no interval, the weights are an EMPTY tensor and rgb / depth / opacity are sums over nothing (rgb = the background
    colour under setbg_opaque).  Rounds 1-2 refused S = 1; now the boundary returns what the reference returns, gradients included."""
    from neural_invertible_warp_amd import ops
    ray, rgb_s, sig, dep = torch.randn(1, 2, 3), torch.rand(1, 2, 1, 3), torch.rand(1, 2, 1), torch.rand(1, 2, 1, 1)
    ref = O.composite(ray, rgb_s, sig, dep)
    assert ref[3].numel() == 0 and float(ref[0].abs().max()) == 0.0
    a, b, c = g(ray[0]).requires_grad_(True), g(rgb_s[0]).requires_grad_(True), g(sig[0]).requires_grad_(True)
    rgb, depth, opacity, prob = ops.composite(a, b, c, g(dep[0, :, :, 0]), bg)
    assert prob.shape == (2, 0)
    assert torch.equal(rgb.cpu(), ref[0][0] + (bg or 0.0)) and torch.equal(depth.cpu(), ref[1][0, :, 0]) and torch.equal(opacity.cpu(), ref[2][0, :, 0])
    (rgb.sum() + depth.sum() + opacity.sum()).backward()
    for x in (a, b, c):
        assert x.grad is not None and float(x.grad.abs().max()) == 0.0


@pytest.mark.parametrize("S", [2, 3, 65])
def test_composite_tiny_and_chunk_boundary(S):
    from neural_invertible_warp_amd import ops
    rng = np.random.default_rng(S)
    N = 5
    ray, rgb_s = t(rng.standard_normal((1, N, 3))), t(rng.uniform(0, 1, (1, N, S, 3)))
    sig, dep = t(rng.uniform(0, 3, (1, N, S))), t(np.sort(rng.uniform(0.5, 6, (1, N, S, 1)), axis=2))
    ref = O.composite(ray, rgb_s, sig, dep)
    out = ops.composite(g(ray[0]), g(rgb_s[0]), g(sig[0]), g(dep[0, :, :, 0]))
    close(out[0], ref[0][0]); close(out[1], ref[1][0, :, 0], atol=1e-4); close(out[2], ref[2][0, :, 0]); close(out[3], ref[3][0, :, :, 0])


def test_pdf_resampling_zero_and_uniform_pdf():
    """zero pdf: every quantile lies beyond the cdf -> all fine samples clamp to the far bound; uniform pdf: the
    mid-point quantiles land on the bin centres' lerp (reference nerf.py:346-365)"""
    from neural_invertible_warp_amd import ops
    S, Sf, rng_ = 16, 24, [1.0, 0.0]
    for pdf in (torch.zeros(3, S), torch.full((3, S), 1.0 / S), torch.rand(3, S) * (torch.rand(3, S) > 0.5)):
        coarse = O.sample_depth(0.5, S, rng_, "inverse")[0, :1].repeat(3, 1, 1, 1)[..., 0].reshape(3, S)
        fine_ref = O.sample_depth_from_pdf(pdf[None], S, Sf, rng_)[0, :, :, 0]
        fine, merged = ops.sample_pdf_merge(g(pdf), g(coarse), Sf, rng_)
        close(fine, fine_ref, atol=1e-6)
        assert (merged[:, 1:] >= merged[:, :-1]).all() and merged.shape == (3, S + Sf)
        close(merged, torch.cat([coarse, fine_ref], 1).sort(dim=1).values, atol=1e-6)
    fine0, _ = ops.sample_pdf_merge(g(torch.zeros(2, S)), g(coarse[:2]), Sf, rng_)
    assert (fine0 == rng_[1]).all()


def _warp_net(exact=True):
    from neural_invertible_warp_amd.model.nvp import nvp_ndr
    net = nvp_ndr.DeformNetwork(d_feature=128, d_in=3, d_out_1=1, d_out_2=3, n_blocks=3, d_hidden=128, n_layers=1, skip_in=[],
                                multires=6, weight_norm=True, actfn="softplus", reference_exact=exact).to(DEV)
    wp = O.make_warp_params(77, 0.02)
    load_nerf(net, wp)
    return net, wp


@pytest.mark.parametrize("B,P", [(1, 1), (2, 17), (5, 3)])
def test_warp_tiny_point_sets(B, P):
    net, wp = _warp_net()
    code, pts = O.make_latent(5, B), torch.randn(B, P, 1, 3, generator=torch.Generator().manual_seed(P))
    cg = g(code).requires_grad_(True)
    y = net.forward(cg, g(pts), alpha_ratio=0.6)
    p64 = {k: v.double().requires_grad_(True) for k, v in wp.items()}
    c64 = code.double().requires_grad_(True)
    y64 = O.warp_forward(p64, c64, pts.double(), 0.6, reference_exact=True)
    close(y, y64.float(), atol=1e-4)
    y.sum().backward(); y64.sum().backward()
    for k, prm in net.named_parameters():
        relclose(prm.grad, p64[k].grad.float(), 1e-2)
    relclose(cg.grad, c64.grad.float(), 1e-2)
    close(net.inverse(cg.detach(), y.detach(), 0.6), pts, atol=2e-4)


def test_warp_view_limit_and_shape_errors():
    from neural_invertible_warp_amd._lib import NiwError
    net, _ = _warp_net(exact=False)
    y = net.forward(torch.randn(64, 128, device=DEV) * 0.1, torch.randn(64, 8, 1, 3, device=DEV), alpha_ratio=1.0)      # the maximum
    assert torch.isfinite(y).all() and y.shape == (64, 8, 1, 3)
    with pytest.raises(NiwError, match="views"):
        net.forward(torch.randn(65, 128, device=DEV), torch.randn(65, 8, 1, 3, device=DEV), alpha_ratio=1.0)
    with pytest.raises(NiwError, match="input_pts"):
        net.forward(torch.randn(2, 128, device=DEV), torch.randn(2, 8, 3, device=DEV), alpha_ratio=1.0)
    with pytest.raises(NiwError, match="deformation_code"):
        net.forward(torch.randn(3, 128, device=DEV), torch.randn(2, 8, 1, 3, device=DEV), alpha_ratio=1.0)
    from neural_invertible_warp_amd.model.nvp import nvp_ndr
    with pytest.raises(NiwError, match="configuration"):
        nvp_ndr.DeformNetwork(d_feature=64, d_in=3, d_out_1=1, d_out_2=3, n_blocks=3, d_hidden=128, n_layers=1, skip_in=[], multires=6)


def test_kabsch_rotation_matches_svd_and_its_autograd():
    """niw_kabsch_rotation_fwd/_bwd vs U diag(1,1,det) V^T from torch.linalg.svd (float64) incl. reflections, a rank-2 and a
    near-degenerate matrix; gradient vs autograd through the float64 SVD."""
    from neural_invertible_warp_amd import ops
    gen = torch.Generator().manual_seed(3)
    M = torch.randn(40, 3, 3, generator=gen)
    M[1] = torch.diag(torch.tensor([2.0, 1.0, -0.5]))                     # det < 0: reflection fix
    M[2] = torch.outer(torch.tensor([1.0, 2, 3]), torch.tensor([0.5, -1, 2])) + torch.outer(torch.tensor([0.0, 1, -1]), torch.tensor([1.0, 1, 0]))   # rank 2
    M[3] = torch.eye(3) * 3 + 1e-3 * torch.randn(3, 3, generator=gen)     # nearly equal singular values
    Md = M.double().requires_grad_(True)
    U, _, Vt = torch.linalg.svd(Md)
    det = torch.det(U @ Vt)
    Rref = U @ torch.diag_embed(torch.stack([torch.ones_like(det), torch.ones_like(det), det], dim=-1)) @ Vt
    G = torch.randn(40, 3, 3, generator=gen)
    keep = [i for i in range(40) if i != 2]                               # rank-deficient: R is not unique, only check orthogonality
    (Rref[keep] * G[keep].double()).sum().backward()
    Mg = M.to(DEV).requires_grad_(True)
    R = ops.kabsch_rotation(Mg)
    close(R[keep], Rref[keep].float(), atol=2e-6, rtol=1e-5)
    eye = torch.eye(3, device=DEV).expand(40, 3, 3)
    close(R @ R.transpose(1, 2), eye, atol=1e-6)
    assert (torch.det(R) > 0.999).all()
    (R[keep] * G[keep].to(DEV)).sum().backward()
    relclose(Mg.grad[keep], Md.grad[keep].float(), 1e-4)


def test_c_program_calls_the_boundary(tmp_path):
    """examples/composite_c_abi.c: a plain C11 program (gcc, HIP runtime C API for memory) linked against libniw_hip.so"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "neural_invertible_warp_amd")
    exe = str(tmp_path / "composite_c_abi")
    subprocess.run(["gcc", "-std=c11", "-D__HIP_PLATFORM_AMD__", os.path.join(root, "examples", "composite_c_abi.c"), "-I" + os.path.join(root, "include"),
                    "-I/opt/rocm/include", "-L" + pkg, "-L/opt/rocm/lib", "-lniw_hip", "-lamdhip64", "-lm", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib",
                    "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr


def test_composite_with_opaque_background():
    """`nerf.setbg_opaque` (nerf.py:470-472): rgb += bgcolor * (1 - opacity); forward and backward vs the oracle"""
    from neural_invertible_warp_amd import ops
    rng = np.random.default_rng(4)
    N, S = 11, 40
    ray = t(rng.standard_normal((1, N, 3))).requires_grad_(True)
    rgb_s = t(rng.uniform(0, 1, (1, N, S, 3))).requires_grad_(True)
    sig = t(rng.uniform(0, 0.6, (1, N, S))).requires_grad_(True)          # thin medium: the background shows through
    dep = t(np.sort(rng.uniform(0.5, 3, (1, N, S, 1)), axis=2))
    dep[..., -1, :] = dep[..., -2, :]                                      # zero-length last interval: opacity stays < 1
    ref = O.composite(ray, rgb_s, sig * 0 + sig, dep, bgcolor=0.8)
    gs = t(rng.standard_normal((N, 3)))
    (ref[0][0] * gs).sum().backward()
    r2, c2, s2 = g(ray.detach()[0]).requires_grad_(True), g(rgb_s.detach()[0]).requires_grad_(True), g(sig.detach()[0]).requires_grad_(True)
    out = ops.composite(r2, c2, s2, g(dep[0, :, :, 0]), bg=0.8)
    close(out[0], ref[0][0]); close(out[2], ref[2][0, :, 0])
    (out[0] * g(gs)).sum().backward()
    relclose(r2.grad, ray.grad[0], 2e-4); relclose(c2.grad, rgb_s.grad[0], 1e-5); relclose(s2.grad, sig.grad[0], 2e-4)


def test_density_noise_regularisation_path():
    """cfg 1 trains with `nerf.density_noise_reg = 1` (nerf.py:428-429): sigma = relu(raw + noise); the noise tensor enters the
    kernel per sample, forward and backward vs the oracle with the same draw."""
    from neural_invertible_warp_amd import ops
    rng = np.random.default_rng(21)
    N, S = 7, 24
    p, names, st, params = _field(12)
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    center, ray = t(rng.uniform(-1, 1, (N, 3))), t(rng.standard_normal((N, 3)))
    depth, noise = t(np.sort(rng.uniform(0.5, 4, (N, S)), axis=1)), t(rng.standard_normal((N, S)))
    w3, wv = O.c2f_weights(0.3, (0.1, 0.5), 10), O.c2f_weights(0.3, (0.1, 0.5), 4)
    rgb_ref, sig_ref = O.forward_samples(pr, center[None], ray[None], depth[None, :, :, None], density_activ="relu", w3d=w3, wview=wv,
                                         density_noise=noise[None])
    (rgb_ref.sum() + sig_ref.sum()).backward()
    rgb, sig = ops.field_mlp(st, params, g(center), g(ray), g(depth), w3.tolist(), wv.tolist(), "relu", noise=g(noise))
    close(rgb, rgb_ref[0]); close(sig, sig_ref[0], atol=5e-5, rtol=2e-4)
    assert (sig == 0).any() and (sig > 0).any()                     # the noise moves samples across the ReLU threshold
    (rgb.sum() + sig.sum()).backward()
    for n, prm in zip(names, params):
        relclose(prm.grad, pr[n].grad, 5e-3)


def test_indexed_weight_packing_equals_the_direct_packing():
    from neural_invertible_warp_amd import ops
    _, _, st, _ = _field(31)
    assert torch.equal(st._pack(), st._pack_decode())
    idx = ops.FieldState._index[str(st.flat.device)]
    assert int((idx >= 0).sum()) >= 2 * 527872 and int(idx.max()) < st.flat.numel()      # every weight appears in both packings


def test_warp_operand_preparation_against_float64_autograd():
    """niw_warp_prep_fwd / _bwd (weight norm, code projection, latent folding) directly against the same formulas in float64
    torch with autograd: outputs 1e-6, every parameter / latent gradient 1e-5 of its scale."""
    from neural_invertible_warp_amd import ops
    net, wp = _warp_net()
    B = 7
    code = O.make_latent(9, B)
    cg = g(code).requires_grad_(True)
    w_emb, view_b, w_head = ops.warp_prepare(net.flat_params, list(net.parameters()), cg)
    p = {k: v.double().requires_grad_(True) for k, v in wp.items()}
    c64 = code.double().requires_grad_(True)
    emb, vb, head = [], [], []
    for b in range(3):
        cb = c64 @ p[f"lin{b}_c.weight"].t() + p[f"lin{b}_c.bias"] + c64
        per_view = []
        for part, E, S in (("a", 26, 28), ("b", 13, 16)):           # w_emb rows are padded to 28 / 16 floats (include/niw.h)
            v_, g_ = p[f"lin{b}_{part}_0.weight_v"], p[f"lin{b}_{part}_0.weight_g"]
            w = v_ * (g_ / v_.norm(dim=1, keepdim=True))
            emb.append(torch.nn.functional.pad(w[:, :E], (0, S - E)).reshape(-1))
            per_view.append(cb @ w[:, E:].t() + p[f"lin{b}_{part}_0.bias"])                 # [B,128]
            head += [p[f"lin{b}_{part}_1.weight"].reshape(-1), p[f"lin{b}_{part}_1.bias"].reshape(-1)]
        vb.append(torch.stack(per_view, dim=1))                                           # [B,2,128]
    emb64, vb64, head64 = torch.cat(emb), torch.stack(vb, dim=1), torch.cat(head)         # vb64 [B,3,2,128]
    close(w_emb, emb64.float(), atol=1e-6, rtol=1e-5); close(view_b, vb64.float(), atol=1e-6, rtol=1e-5); close(w_head, head64.float(), atol=0, rtol=0)
    gen = torch.Generator().manual_seed(1)
    ge, gv, gh = torch.randn(emb64.shape, generator=gen), torch.randn(vb64.shape, generator=gen), torch.randn(head64.shape, generator=gen)
    ((emb64 * ge.double()).sum() + (vb64 * gv.double()).sum() + (head64 * gh.double()).sum()).backward()
    ((w_emb * g(ge)).sum() + (view_b * g(gv)).sum() + (w_head * g(gh)).sum()).backward()
    for k, prm in net.named_parameters():
        relclose(prm.grad, p[k].grad.float(), 1e-5)
    relclose(cg.grad, c64.grad.float(), 1e-5)


def test_fused_adam_matches_torch_adam():
    """niw_adam_step over a flat buffer vs torch.optim.Adam on the CPU with the same gradients, 20 steps, odd length"""
    from neural_invertible_warp_amd import ops
    gen = torch.Generator().manual_seed(5)
    n = 100003
    p0 = torch.randn(n, generator=gen)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=3e-3)
    flat, m, v = g(p0.clone()), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for it in range(1, 21):
        grad = torch.randn(n, generator=gen) * (1.0 if it % 3 else 1e-3)
        ref.grad = grad.clone()
        opt.step()
        ops.adam_step(flat, g(grad), m, v, 3e-3, it)
    relclose(flat, ref.detach(), 2e-6)
    relclose(m, opt.state[ref]["exp_avg"], 1e-6); relclose(v, opt.state[ref]["exp_avg_sq"], 1e-6)      # fma vs mul + add: last-bit differences


def test_photometric_loss_kernel_direct():
    """niw_mse_fwd_bwd vs torch: gather of image[:, :, ray_idx], mean over 3*B*R (or the global count under sharding), gradient"""
    from neural_invertible_warp_amd import ops
    gen = torch.Generator().manual_seed(8)
    B, H, W, R = 3, 10, 12, 17
    image, rgb = torch.rand(B, 3, H, W, generator=gen), torch.rand(B, R, 3, generator=gen).requires_grad_(True)
    idx = torch.randperm(H * W, generator=gen)[:R]
    target = image.view(B, 3, H * W).permute(0, 2, 1)[:, idx]
    ref = ((rgb - target) ** 2).mean()
    ref.backward()
    r2 = g(rgb.detach()).requires_grad_(True)
    loss = ops.mse_gather(r2, g(image), g(idx))
    close(loss, ref, atol=1e-7, rtol=1e-6)
    loss.backward()
    relclose(r2.grad, rgb.grad, 1e-6)
    # global normaliser of a sharded batch (twice the local element count): loss and gradient halve
    r3 = g(rgb.detach()).requires_grad_(True)
    half = ops.mse_gather(r3, g(image), g(idx), n_norm=2 * 3 * B * R)
    half.backward()
    close(half, ref / 2, atol=1e-7, rtol=1e-6); relclose(r3.grad, rgb.grad / 2, 1e-6)
    # full image, no ray_idx (val / eval)
    full = torch.rand(B, H * W, 3, generator=gen)
    close(ops.mse_gather(g(full), g(image)), ((full - image.view(B, 3, H * W).permute(0, 2, 1)) ** 2).mean(), atol=1e-7, rtol=1e-6)


def test_gradient_sinks_receive_what_autograd_would_accumulate():
    """engine.INNTrainer hands the kernels its bucket segments as gradient sinks (ops.field_mlp / ops.warp_prepare grad_sink): the
    sink must hold exactly the gradients that the Parameters' .grad would have received, and .grad must stay None."""
    from neural_invertible_warp_amd import ops
    p, names, state, params = _field(3)
    gen = torch.Generator().manual_seed(5)
    N, S = 37, 24
    center, ray = g(torch.randn(N, 3, generator=gen) * 0.1), g(torch.randn(N, 3, generator=gen))
    depth = g((torch.rand(N, S, generator=gen).sort(dim=1).values * 4 + 0.5).contiguous())
    w_rgb, w_sig = g(torch.randn(N, S, 3, generator=gen)), g(torch.randn(N, S, generator=gen))

    def run(sink):
        for q in params:
            q.grad = None
        rgb, sigma = ops.field_mlp(state, params, center, ray, depth, [1.0] * 10, [1.0] * 4, "softplus", grad_sink=sink)
        ((rgb * w_rgb).sum() + (sigma * w_sig).sum()).backward()

    run(None)
    ref = torch.cat([q.grad.reshape(-1) for q in params])
    sink = torch.full((ops.NERF_PARAM_FLOATS,), float("nan"), device=DEV)
    run(sink)
    assert all(q.grad is None for q in params)
    assert torch.equal(sink, ref)                      # same kernels, same order: bit-identical

    # the warp's operand preparation: parameters and the per-view codes
    net, wp = _warp_net()
    code = g(O.make_latent(5, 4)).requires_grad_(True)
    pts = g(torch.randn(4, 19, 1, 3, generator=gen))

    def run_warp(sink):
        net.grad_sink = sink
        for q in net.parameters():
            q.grad = None
        code.grad = None
        net.forward(code, pts, alpha_ratio=0.6).square().sum().backward()
        net.grad_sink = None

    run_warp(None)
    ref_p, ref_c = torch.cat([q.grad.reshape(-1) for q in net.parameters()]), code.grad.reshape(-1).clone()
    sp, sc = torch.full_like(ref_p, float("nan")), torch.full_like(ref_c, float("nan"))
    run_warp((sp, sc))
    assert all(q.grad is None for q in net.parameters()) and code.grad is None
    assert torch.equal(sp, ref_p) and torch.equal(sc, ref_c)


def test_warp_backward_from_saved_block_inputs_equals_recomputation():
    """niw_warp_fwd may leave every coupling block's input point behind (xin_save) so that niw_warp_bwd skips its own forward sweep:
    both routes must give the same gradients bit for bit, and the saved points must be the block inputs (block 0: the points)."""
    import ctypes
    from neural_invertible_warp_amd import _lib, ops
    B, N = 3, 41
    gen = torch.Generator().manual_seed(9)
    w_emb = g(torch.randn(ops.WARP_WEMB_FLOATS, generator=gen) * 0.1)
    view_b = g(torch.randn(B, 3, 2, 128, generator=gen) * 0.1)
    w_head = g(torch.randn(ops.WARP_WHEAD_FLOATS, generator=gen) * 0.02)
    pts, d_out = g(torch.randn(B, N, 3, generator=gen)), g(torch.randn(B, N, 3, generator=gen))
    out, xin = torch.empty_like(pts), torch.empty(B, N, 3, 3, device=DEV)
    P, st = ops._p, ops._stream()
    cw, iw = ops._farr([1.0, 1.0, 0.7, 0.2, 0.0, 0.0], 6), ops._farr([0.3, 0.6, 1, 1, 1, 1], 6)
    _lib.call("niw_warp_fwd", P(w_emb), P(view_b), P(w_head), P(pts), B, N, cw, iw, None, 0, None, None, 0, P(out), P(xin), st)
    assert torch.equal(xin[:, :, 0], pts)
    ws = torch.empty(_lib.load().niw_warp_bwd_workspace_floats(B, N), device=DEV)

    def bwd(saved):
        o = [torch.empty_like(w_emb), torch.empty_like(view_b), torch.empty_like(w_head), torch.empty_like(pts)]
        _lib.call("niw_warp_bwd", P(w_emb), P(view_b), P(w_head), P(pts), B, N, cw, iw, None, 0, None, None, P(saved), P(d_out), P(ws),
                  P(o[0]), P(o[1]), P(o[2]), P(o[3]), st)
        return o

    # (not bit for bit: the forward kernel and the backward kernel's own sweep are separately compiled copies of the same
    # arithmetic, and the 2^5 pi band of the embedding amplifies a last-bit difference of a block input; measured 1e-6)
    for a, b in zip(bwd(None), bwd(xin)):
        assert (a - b).abs().max() <= 1e-5 * a.abs().max(), float((a - b).abs().max() / a.abs().max())
    # the inverse warp has no saved inputs
    with pytest.raises(_lib.NiwError):
        _lib.call("niw_warp_fwd", P(w_emb), P(view_b), P(w_head), P(pts), B, N, cw, iw, None, 0, None, None, 1, P(out), P(xin), st)


def test_ray_split_node_matches_plain_autograd():
    """nvp_ndr._SplitRays ([grid ; centre] -> ray, centre, grid in one node) against slices and a subtraction, with every
    combination of used / unused outputs"""
    from neural_invertible_warp_amd.model.nvp.nvp_ndr import _SplitRays
    gen = torch.Generator().manual_seed(2)
    w = g(torch.randn(3, 10, 3, generator=gen))
    gs = [g(torch.randn(3, 5, 3, generator=gen)) for _ in range(3)]
    for use in ((1, 1, 1), (1, 0, 0), (0, 1, 1), (1, 1, 0), (0, 0, 1)):
        a = w.clone().requires_grad_(True)
        outs = _SplitRays.apply(a, 5)
        sum((o * gi).sum() for o, gi, u in zip(outs, gs, use) if u).backward()
        b = w.clone().requires_grad_(True)
        ref = (b[:, :5] - b[:, 5:], b[:, 5:], b[:, :5])
        sum((o * gi).sum() for o, gi, u in zip(ref, gs, use) if u).backward()
        for o, r in zip(outs, ref):
            assert torch.equal(o, r)
        close(a.grad, b.grad, atol=1e-7)


@pytest.mark.parametrize("N,S", [(750, 128), (751, 128), (749, 128), (1101, 120)])
def test_weight_gradient_is_the_same_on_both_sides_of_the_heads_threshold(N, S):
    """niw_mlp_bwd_dw forms the density row and the colour rows in dw_heads_kernel (vector ALU, second stream) from 96,000 samples (round 5;
    131,072 in round 4) and as two pieces of the skinny MFMA launch below.  The same rays evaluated in ONE call (N x 128 samples: 96,000 at
    N = 750 -- the first size on the heads path -- 96,128 with a ragged last chunk, 95,872 just below the threshold, and 1101 x 120 =
    132,120 samples, which the workspaces pad to 132,224: the kernel reads the padding columns too) and as two half batches (always below)
    must give the same parameter gradients up to the order of an fp32 sum over the samples."""
    from neural_invertible_warp_amd import ops
    rng = np.random.default_rng(N)
    center, ray = g(t(rng.uniform(-1, 1, (N, 3)))), g(t(rng.standard_normal((N, 3))))
    depth = g(t(np.sort(rng.uniform(0.5, 4, (N, S)), axis=1)))
    g_rgb, g_sig = g(t(rng.standard_normal((N, S, 3)))), g(t(rng.standard_normal((N, S))))
    ones3, ones4 = [1.0] * 10, [1.0] * 4

    def grads(parts):
        _, names, st, params = _field(8)
        for lo, hi in parts:
            rgb, sig = ops.field_mlp(st, params, center[lo:hi].contiguous(), ray[lo:hi].contiguous(), depth[lo:hi].contiguous(), ones3, ones4, "softplus")
            ((rgb * g_rgb[lo:hi]).sum() + (sig * g_sig[lo:hi]).sum()).backward()
        return names, [prm.grad.clone() for prm in params]

    names, whole = grads([(0, N)])
    _, halves = grads([(0, N // 2), (N // 2, N)])
    for n, a, b in zip(names, whole, halves):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * scale, (n, float((a - b).abs().max()) / scale)
