"""Size-independent properties checked at BASELINE.json's full shapes (18 views x 227 rays x 64 / 192
samples, 8172 warp points), where the CPU oracle would take minutes: round trips, sortedness,
conservation, determinism, linearity of the backward.  Needs a GPU."""
import pytest
import torch

from oracle import niw_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, R, S, SF = 18, 227, 64, 128
N = B * R


def _flat_params(seed):
    p = O.make_nerf_params(seed)
    names = [f"{n}.{k}" for n, _, _ in O.nerf_layer_shapes() for k in ("weight", "bias")]
    flat = torch.cat([p[n].reshape(-1) for n in names]).to(DEV)
    views, off = [], 0
    for n in names:
        views.append(flat[off:off + p[n].numel()].view(p[n].shape).requires_grad_(True))
        off += p[n].numel()
    return flat, views


def _rays(seed):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    center = torch.randn(N, 3, generator=gen).mul(0.1).to(DEV)
    ray = torch.randn(N, 3, generator=gen).to(DEV)
    return center, ray


def test_composite_conservation_full_size():
    from neural_invertible_warp_amd import ops
    gen = torch.Generator().manual_seed(0)
    ray = torch.randn(N, 3, generator=gen).to(DEV)
    rgb_s = torch.rand(N, S + SF, 3, generator=gen).to(DEV)
    sig = (torch.rand(N, S + SF, generator=gen) * 3 + 1e-3).to(DEV)
    depth = ops.sample_stratified(torch.rand(N, S + SF, generator=gen).to(DEV), N, S + SF, (1, 0), "inverse", DEV)
    rgb, d, opa, prob = ops.composite(ray, rgb_s, sig, depth)
    assert torch.isfinite(rgb).all() and torch.isfinite(d).all()
    torch.testing.assert_close(prob.sum(-1), opa, atol=2e-6, rtol=0)          # weights sum to the opacity
    assert (opa - 1).abs().max() < 1e-6                                        # sigma > 0 and a 1e10 closing interval
    assert (rgb.min() >= 0) and (rgb.max() <= 1 + 1e-5)                        # convex combination of colours


def test_hierarchical_resampling_sorted_and_bounded_full_size():
    from neural_invertible_warp_amd import ops
    gen = torch.Generator().manual_seed(1)
    coarse = ops.sample_stratified(torch.rand(N, S, generator=gen).to(DEV), N, S, (1, 0), "inverse", DEV)
    pdf = (torch.rand(N, S, generator=gen) * (torch.rand(N, S, generator=gen) > 0.5) / S).to(DEV)
    pdf[0] = 0                                                                 # all-zero pdf: every quantile at the far bound
    fine, merged = ops.sample_pdf_merge(pdf, coarse, SF, (1, 0))
    assert merged.shape == (N, S + SF)
    assert torch.all(merged[:, 1:] >= merged[:, :-1])                          # ascending
    assert torch.all(fine[0] == 0.0) and fine.min() >= 0 and fine.max() <= 1   # bins live in the parametrised space [1, 0]
    # merged is a permutation of coarse ++ fine
    torch.testing.assert_close(merged.sum(-1, dtype=torch.float64), (coarse.sum(-1, dtype=torch.float64) + fine.sum(-1, dtype=torch.float64)), rtol=1e-9, atol=0)


def test_warp_round_trip_full_size():
    from neural_invertible_warp_amd.model.nvp import nvp_ndr
    net = nvp_ndr.DeformNetwork(d_feature=128, d_in=3, d_out_1=1, d_out_2=3, n_blocks=3, d_hidden=128, n_layers=1, skip_in=[],
                                multires=6, weight_norm=True, actfn="softplus").to(DEV)
    sd = net.state_dict()
    with torch.no_grad():
        for k, v in O.make_warp_params(9, 0.02).items():
            sd[k].copy_(v)
    code = O.make_latent(3, B).to(DEV)
    pts = (torch.rand(B, 2 * R, 1, 3, device=DEV) - 0.5)
    for alpha in (0.0, 0.37, 1.0):
        y = net.forward(code, pts, alpha_ratio=alpha)
        assert (y - pts).abs().max() > 1e-3                                    # the warp is not the identity here
        back = net.inverse(code, y, alpha)
        assert (back - pts).abs().max() < 5e-5


def test_mlp_forward_deterministic_and_backward_linear_full_size():
    from neural_invertible_warp_amd import ops
    flat, params = _flat_params(4)
    st = ops.FieldState(flat)
    center, ray = _rays(5)
    depth = ops.sample_stratified(torch.rand(N, S, device=DEV), N, S, (1, 0), "inverse", DEV)
    w3, wv = [1.0] * 10, [1.0] * 4
    with torch.no_grad():
        a = ops.field_mlp(st, [], center, ray, depth, w3, wv, "softplus")
        b = ops.field_mlp(st, [], center, ray, depth, w3, wv, "softplus")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])                # bit-reproducible forward
    assert torch.isfinite(a[0]).all() and torch.isfinite(a[1]).all()
    assert a[0].min() >= 0 and a[0].max() <= 1 and a[1].min() >= 0            # sigmoid colours, softplus density
    g_rgb, g_sig = torch.randn(N, S, 3, device=DEV), torch.randn(N, S, device=DEV)

    def grads(scale):
        for p in params:
            p.grad = None
        rgb, sig = ops.field_mlp(st, params, center, ray, depth, w3, wv, "softplus")
        ((rgb * g_rgb).sum() * scale + (sig * g_sig).sum() * scale).backward()
        return torch.cat([p.grad.reshape(-1) for p in params])

    g1, g1b, g2 = grads(1.0), grads(1.0), grads(2.0)
    assert torch.equal(g1, g1b)                                                # dW has no float atomics: bit-reproducible
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    assert (g2 - 2 * g1).abs().max() <= 1e-5 * g1.abs().max()                  # the backward is linear in the upstream gradient


def test_full_train_step_cfg2_finite_and_learns():
    """cfg2 shapes through the engine: losses finite, all gradients finite, the loss goes down."""
    from neural_invertible_warp_amd import configs, engine
    opt = configs.cfg2_nerf_inn_llff_hier(device=DEV)
    tr = engine.INNTrainer(opt, B, warp_perturb=0.02)
    var0 = engine.synthetic_scene(opt, B)
    losses = []
    for _ in range(6):
        loss = tr.train_iteration(type(var0)(var0))
        losses.append(float(loss.all.detach()))
        assert torch.isfinite(tr.bucket.flat).all()
    assert all(l == l and l < 1 for l in losses)
    assert losses[-1] < losses[0]


def test_mlp_batches_beyond_one_launch_are_split_over_rays():
    """> 2^24 samples in one call (here 70,000 rays x 256): the wrapper splits over the rays; the pieces must agree with direct
    calls on sub-ranges (size-independent property: per-ray independence)."""
    from neural_invertible_warp_amd import ops
    from oracle import niw_oracle as O
    p = O.make_nerf_params(2)
    flat = torch.cat([p[f"{n}.{k}"].reshape(-1) for n, _, _ in O.nerf_layer_shapes() for k in ("weight", "bias")]).to(DEV)
    st = ops.FieldState(flat)
    N, S = 70000, 256
    gen = torch.Generator(device=DEV).manual_seed(0)
    center, ray = torch.randn(N, 3, device=DEV, generator=gen), torch.randn(N, 3, device=DEV, generator=gen)
    depth = torch.rand(N, S, device=DEV, generator=gen).sort(dim=1).values * 4 + 0.5
    with torch.no_grad():
        rgb, sig = ops.field_mlp(st, [], center, ray, depth, [1.0] * 10, [1.0] * 4, "softplus")
        assert rgb.shape == (N, S, 3) and torch.isfinite(rgb).all() and torch.isfinite(sig).all()
        for a, b in ((0, 100), (65000, 65600), (N - 50, N)):                    # across and beyond the split point
            r2, s2 = ops.field_mlp(st, [], center[a:b], ray[a:b], depth[a:b], [1.0] * 10, [1.0] * 4, "softplus")
            assert torch.equal(rgb[a:b], r2) and torch.equal(sig[a:b], s2)


def test_mlp_training_batches_beyond_the_gradient_launch_limit_are_split():
    """With gradients one launch takes fewer than 2^31 / (288 * 4) = 1.86 M samples (the dW GEMM reaches a 288-row operand through one
    2 GiB buffer descriptor).  2.36 M samples in one differentiable call: the wrapper splits over the rays and autograd sums the pieces; the
    parameter gradient must equal the sum of the gradients of the same pieces run by hand (and be finite), and a gradient sink
    -- which cannot be summed into -- must be refused loudly."""
    from neural_invertible_warp_amd import _lib, ops
    from oracle import niw_oracle as O
    p = O.make_nerf_params(2)
    names = [f"{n}.{k}" for n, _, _ in O.nerf_layer_shapes() for k in ("weight", "bias")]
    flat = torch.cat([p[n].reshape(-1) for n in names]).to(DEV)
    params, off = [], 0
    for n in names:
        params.append(flat[off:off + p[n].numel()].view(p[n].shape).requires_grad_(True))
        off += p[n].numel()
    st = ops.FieldState(flat)
    N, S = 18432, 128                                  # 2,359,296 samples
    gen = torch.Generator(device=DEV).manual_seed(0)
    center, ray = torch.randn(N, 3, device=DEV, generator=gen) * 0.1, torch.randn(N, 3, device=DEV, generator=gen)
    depth = (torch.rand(N, S, device=DEV, generator=gen).sort(dim=1).values * 4 + 0.5).contiguous()
    w = torch.randn(N, S, 3, device=DEV, generator=gen)

    def grads(pieces):
        for q in params:
            q.grad = None
        for a, b in pieces:
            rgb, sig = ops.field_mlp(st, params, center[a:b], ray[a:b], depth[a:b], [1.0] * 10, [1.0] * 4, "softplus")
            ((rgb * w[a:b]).sum() + sig.sum()).backward()
        return torch.cat([q.grad.reshape(-1) for q in params])

    whole = grads([(0, N)])
    cut = (ops.TRAIN_LAUNCH_SAMPLES - 256) // S        # where the wrapper cuts
    by_hand = grads([(0, cut), (cut, N)])
    assert torch.isfinite(whole).all()
    assert (whole - by_hand).abs().max() <= 1e-6 * whole.abs().max()
    with pytest.raises(_lib.NiwError):
        ops.field_mlp(st, params, center, ray, depth, [1.0] * 10, [1.0] * 4, "softplus", grad_sink=torch.empty(ops.NERF_PARAM_FLOATS, device=DEV))
