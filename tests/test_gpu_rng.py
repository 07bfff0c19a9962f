"""The in-kernel stratified draw (niw_sample_stratified_rng: Philox4x32-10 keyed by seed, counter = (sample group, draw number)) --
what replaces the reference's device-side torch.rand (model/nerf.py:337) under the engine.  Known-answer vectors of the generator, the
stream properties the engine relies on (pure function of seed / draw / sample index; device-resident draw number), uniformity, and
the depths against the oracle on the kernel's own draws (bit-exact, like niw_sample_stratified).  Needs a GPU."""
import numpy as np
import pytest
import torch

from oracle import niw_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _philox4x32_10(counter, key):
    """host restatement of the published algorithm (Salmon, Moraes, Dror, Shaw: Parallel random numbers: as easy as 1, 2, 3; SC'11)"""
    c, (k0, k1) = list(counter), key
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k0, p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k1, p0 & 0xffffffff]
        k0, k1 = (k0 + 0x9E3779B9) & 0xffffffff, (k1 + 0xBB67AE85) & 0xffffffff
    return c


def test_host_restatement_reproduces_the_published_known_answers():
    assert _philox4x32_10((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox4x32_10((0xffffffff,) * 4, (0xffffffff,) * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


@pytest.mark.parametrize("seed,draw", [(0, 0), (0x299f31d0a4093822, 0x0370734413198a2e), (7, 123456)])
def test_kernel_draws_are_philox_of_seed_draw_and_sample_group(seed, draw):
    from neural_invertible_warp_amd import ops
    n_rays, S = 5, 12                                       # 60 samples = 15 groups of four
    _, u = ops.sample_stratified_rng(seed, draw, n_rays, S, (0.0, 1.0), "metric", DEV, return_u=True)
    u = u.cpu().numpy().reshape(-1)
    for g in range(15):
        words = _philox4x32_10((g, 0, draw & 0xffffffff, draw >> 32), (seed & 0xffffffff, seed >> 32))
        expect = np.array([(w >> 8) * 2.0 ** -24 for w in words], dtype=np.float32)
        assert np.array_equal(u[4 * g:4 * g + 4], expect), (g, u[4 * g:4 * g + 4], expect)


@pytest.mark.parametrize("param,rng", [("inverse", (1.0, 0.0)), ("metric", (1.2, 5.2))])
def test_depths_equal_the_oracle_on_the_kernels_own_draws(param, rng):
    from neural_invertible_warp_amd import ops
    n_rays, S = 4086, 64
    d, u = ops.sample_stratified_rng(11, 3, n_rays, S, rng, param, DEV, return_u=True)
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0
    ref = O.sample_depth(u.cpu().view(1, n_rays, S, 1), S, rng, param)
    assert torch.equal(d.cpu(), ref.view(n_rays, S))                        # bit-exact, like the u-fed kernel
    assert torch.equal(d, ops.sample_stratified(u, n_rays, S, rng, param, DEV))
    assert bool((d[:, 1:] > d[:, :-1]).all())                # one sample per stratum, ascending (the inverse range is given as (1, 0))


def test_stream_properties():
    from neural_invertible_warp_amd import ops
    a = ops.sample_stratified_rng(5, 9, 300, 128, (1.0, 0.0), "inverse", DEV, return_u=True)[1]
    b = ops.sample_stratified_rng(5, 9, 300, 128, (1.0, 0.0), "inverse", DEV, return_u=True)[1]
    assert torch.equal(a, b)                                                                    # reproducible
    c = ops.sample_stratified_rng(5, 9, 77, 128, (1.0, 0.0), "inverse", DEV, return_u=True)[1]
    assert torch.equal(a.view(-1)[:77 * 128], c.view(-1))                                       # independent of the launch size
    other_draw = ops.sample_stratified_rng(5, 10, 300, 128, (1.0, 0.0), "inverse", DEV, return_u=True)[1]
    other_seed = ops.sample_stratified_rng(6, 9, 300, 128, (1.0, 0.0), "inverse", DEV, return_u=True)[1]
    assert float((a == other_draw).float().mean()) < 1e-3 and float((a == other_seed).float().mean()) < 1e-3
    # the draw number read from device memory (captured-graph replays) overrides the by-value one
    word = torch.tensor([10], dtype=torch.int64, device=DEV)
    assert torch.equal(ops.sample_stratified_rng(5, 0, 300, 128, (1.0, 0.0), "inverse", DEV, draw_dev=word, return_u=True)[1], other_draw)


def test_uniformity_of_two_million_draws():
    from neural_invertible_warp_amd import ops
    u = ops.sample_stratified_rng(2024, 1, 16384, 128, (0.0, 1.0), "metric", DEV, return_u=True)[1].double().view(-1)
    n = u.numel()
    assert abs(float(u.mean()) - 0.5) < 4 * (1 / 12 / n) ** 0.5
    assert abs(float(u.var()) - 1 / 12) < 1e-3
    hist = torch.histc(u.float(), bins=64, min=0.0, max=1.0).double()
    chi2 = float(((hist - n / 64) ** 2 / (n / 64)).sum())
    assert chi2 < 63 + 6 * (2 * 63) ** 0.5, chi2                                               # chi-square, 63 dof, +6 sigma
    for lag in (1, 2, 3, 4, 128):                                                               # neighbours inside and across groups / rays
        r = float(((u[:-lag] - 0.5) * (u[lag:] - 0.5)).mean() * 12)
        assert abs(r) < 5 / n ** 0.5, (lag, r)


def test_engine_uses_the_in_kernel_draw_and_resumes_its_stream():
    """engine default on the GPU: Feistel pixel draw + Philox depth draw, both keyed by the iteration -- two trainers that reach
    iteration k by different routes (from 0, or set there like a resumed run) draw the same samples at k"""
    from neural_invertible_warp_amd import configs, engine

    def trainer(fused=False):
        opt = configs.cfg3_barf_inn_llff(device=DEV)
        opt.H, opt.W = 12, 16
        opt.nerf.rand_rays, opt.nerf.sample_intvs, opt.max_iter = 5 * 16, 32, 40
        return opt, engine.synthetic_scene(opt, 5), engine.INNTrainer(opt, 5, warp_perturb=0.02, seed=5, fused_step=fused)

    # the one-call iteration (niw_train_step) numbers its draws the same way: draw = iteration + 1, stream key of call 0
    _, varf, f = trainer(fused=True)
    f.it = 2
    f.train_iteration(type(varf)(varf))
    assert f.fused is not None and f.fused.desc.draw == 3 and f.fused.desc.stratified == 1
    opt, var0, a = trainer()
    assert opt.nerf.stratified_rng == "philox" and opt.nerf.ray_sampler == "feistel"
    import neural_invertible_warp_amd.ops as ops_mod
    seen = []
    orig = ops_mod.sample_stratified_rng
    ops_mod.sample_stratified_rng = lambda *a_, **k: (seen.append(a_[1]) or orig(*a_, **k))
    try:
        for _ in range(3):
            a.train_iteration(type(var0)(var0))
        _, var1, b = trainer()
        b.it = 2
        b.train_iteration(type(var1)(var1))
    finally:
        ops_mod.sample_stratified_rng = orig
    assert seen == [1, 2, 3, 3]                              # draw number = iteration + 1, also after the jump


def test_normal_rng_is_box_muller_over_the_keyed_philox_stream():
    """niw_normal_rng (round 6; the density noise of reference model/nerf.py:428-429 drawn on the device): values 4g .. 4g+3 come from
    the Philox words of counter (g, draw) under key seed -- r = sqrt(-2 ln u1), theta = 2 pi u2 -- so a numpy restatement reproduces them
    to fp32 libm accuracy; moments of a million draws; the draw number may come from device memory; a pure function of its key."""
    import math
    import numpy as np
    from neural_invertible_warp_amd import ops
    seed, draw, scale = 0x1234567890ABCDEF, 7, 0.5
    z = ops.normal_rng(seed, draw, 1003, scale, DEV).cpu().numpy()
    ref = np.zeros(1004, dtype=np.float64)
    for g in range(251):
        w = _philox4x32_10((g, 0, draw & 0xffffffff, draw >> 32), (seed & 0xffffffff, seed >> 32))
        for t in range(2):
            u1, u2 = ((w[2 * t] >> 8) + 1) * 2.0 ** -24, (w[2 * t + 1] >> 8) * 2.0 ** -24
            r = math.sqrt(-2.0 * math.log(u1))
            ref[4 * g + 2 * t], ref[4 * g + 2 * t + 1] = scale * r * math.cos(2 * math.pi * u2), scale * r * math.sin(2 * math.pi * u2)
    assert np.abs(z - ref[:1003]).max() <= 2e-6 * scale * 6
    big = ops.normal_rng(seed, draw, 1 << 20, 1.0, DEV).double()
    m, v = float(big.mean()), float(big.var())
    k = float(((big - m) ** 4).mean() / v ** 2)
    assert abs(m) < 4e-3 and abs(v - 1) < 6e-3 and abs(k - 3) < 0.05, (m, v, k)
    assert torch.isfinite(big).all() and float(big.abs().max()) < 6.0                  # u1 >= 2^-24: |z| <= sqrt(2 * 24 ln 2) = 5.77
    assert torch.equal(ops.normal_rng(seed, draw, 4096, 1.0, DEV), big[:4096].float())
    dev_draw = torch.tensor([draw], dtype=torch.int64, device=DEV)
    assert torch.equal(ops.normal_rng(seed, 99, 4096, 1.0, DEV, draw_dev=dev_draw), big[:4096].float())
    assert not torch.equal(ops.normal_rng(seed + 1, draw, 4096, 1.0, DEV), big[:4096].float())
