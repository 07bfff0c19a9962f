"""The HIP path against the oracle at the FULL shapes of the BASELINE configurations.

The oracle is plain PyTorch and follows the device of its inputs, so at these sizes it runs on the same MI355X through torch's own
ROCm kernels (seconds; on the host cores it would take minutes).  It is still the checker: the product never calls it.

    cfg2   18 views x 227 rays x (64 coarse + 192 fine) = 1,046,016 MLP evaluations, inverse depth, hierarchical resampling
    cfg3   18 views x 113 rays x 128, c2f encoding at progress 0.3, alignment term x 1e4
    cfg5   3 views x 682 rays x 128, metric depth [1.2, 5.2], poses composed with noisy initial poses (unwarped rays in the world frame)

Forward values at the small-shape tolerances (rgb / opacity atol 3e-5 rtol 2e-4, warped points 2e-5, loss 1e-6) against the oracle's
fp32 evaluation.  GRADIENTS (round 4) against the oracle evaluated in FLOAT64 on the same GPU at the same full shapes -- the same
function with 29 more bits, ~25 GB of autograd state at cfg2 -- and held to a conditioning bound instead of rounds 1-3's blanket
2e-2 / 5e-2 against another fp32 evaluation.  The yardstick is the COMPARATOR'S OWN fp32 evaluation (torch's kernels, same inputs)
against the same float64 gradients: a tensor of the HIP path may deviate from the float64 gradient by max(3 x what torch's fp32 does
on THAT tensor, 3 x the median of what it does over the tensors of the group) + 2e-4 of the tensor's scale -- i.e. the HIP path must
be as good an fp32 evaluation of these gradients as torch's, tensor by tensor (conditioning differs by 20 x between tensors of one
network: the first layer sees the 2^9 pi band).  Measured worst / bound and worst HIP / comparator ratios are printed per group.
Why a group-median branch at all (round 5, profiles/r5_fp64_parity_table.txt): the comparator's deviation on ONE tensor is itself a draw
of rounding noise.  Round 4 reported cfg2 warp_mlp.lin1_c.weight at 5.45 x torch's own deviation; in this round's run the same tensor
sits at 1.11 x (2.45e-2 against torch's 2.21e-2) and the largest ratio, 5.47, belongs to the one-element lin0_a_1.bias, where torch
happened to land within 8e-4 of float64 and the HIP path at 0.43 x the group median: the maximum over 36 tensors of a ratio of two
noisy numbers.  No tensor of any group lies above 2.7 x its group's median, so the group factor is 3 (round 4: 6).
Needs a GPU."""
import numpy as np
import pytest
import torch

from oracle import niw_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H, W = 300, 400


def _rel(a, b):
    return float((a.detach() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-30))


def _load(module, params):
    sd = module.state_dict()
    with torch.no_grad():
        for k, v in params.items():
            sd[k].copy_(v)


def _inputs(B, R, S, seed):
    rng = np.random.default_rng(seed)
    image = torch.from_numpy(rng.uniform(0, 1, (B, 3, H, W)).astype(np.float32)).to(DEV)
    intr = torch.tensor([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]], dtype=torch.float32).repeat(B, 1, 1).to(DEV)
    u = torch.from_numpy(rng.uniform(0, 1, (B, R, S, 1)).astype(np.float32)).to(DEV)
    ray_idx = torch.from_numpy(rng.permutation(H * W)[:R].astype(np.int64)).to(DEV)
    return image, intr, u, ray_idx


class _Rng:
    """the reference draws torch.rand / torch.randperm inside the path (nerf.py:258, 337): hand it the test's draws"""

    def __init__(self, u, ray_idx):
        self.u, self.ray_idx = u, ray_idx

    def __enter__(self):
        self.saved = torch.rand, torch.randperm
        torch.rand, torch.randperm = (lambda *a, **k: self.u.clone()), (lambda *a, **k: self.ray_idx.clone())

    def __exit__(self, *exc):
        torch.rand, torch.randperm = self.saved


def _dev_params(seed_c, seed_w, seed_l, B, seed_f=None):
    req = lambda d: {k: v.to(DEV).requires_grad_(True) for k, v in d.items()}
    pc, wp = req(O.make_nerf_params(seed_c)), req(O.make_warp_params(seed_w, 0.02))
    pf = req(O.make_nerf_params(seed_f)) if seed_f is not None else None
    lat = O.make_latent(seed_l, B).to(DEV).requires_grad_(True)
    return pc, pf, wp, lat


import os

FACTOR_TENSOR, FACTOR_GROUP, FLOOR = 3.0, float(os.environ.get("NIW_PARITY_FACTOR_GROUP", 3.0)), 2e-4
TABLE = os.environ.get("NIW_PARITY_TABLE")          # a file: every tensor's row is appended (tools/collect_profiles.sh -> profiles/r5_fp64_parity_table.txt)


def _err(g, g64, scale64):
    return float((g.detach().double() - g64).abs().max() / scale64)


def _scale64(key, grads64):
    """max |g64| of the tensor; for the 1- and 3-element head biases of the warp (sums over all points that largely cancel) the max of
    the same layer's WEIGHT gradient, the scale of the terms they are summed from (tests/util.check_grad_vs_fp64 does the same)"""
    s = float(grads64[key].abs().max())
    if key.endswith("_1.bias") and grads64[key].numel() <= 16:
        s = max(s, float(grads64[key[:-len("bias")] + "weight"].abs().max()))
    return max(s, 1e-300)


def _check_group_vs_fp64(name, hip, torch32, grads64, report):
    """hip / torch32 / grads64: {key: gradient} of one optimizer group (HIP path, oracle fp32 on the GPU, oracle float64 on the GPU)"""
    keys = [k for k in grads64 if grads64[k] is not None and float(grads64[k].abs().max()) > 0]
    cond = {k: _err(torch32[k], grads64[k], _scale64(k, grads64)) for k in keys}
    mine = {k: _err(hip[k], grads64[k], _scale64(k, grads64)) for k in keys}
    med = float(np.median(list(cond.values())))
    bound = {k: max(FACTOR_TENSOR * cond[k], FACTOR_GROUP * med) + FLOOR for k in keys}
    worst = max(keys, key=lambda k: mine[k] / bound[k])
    report.append(f"  {name:<14} {len(keys):3d} tensors | comparator fp32 vs fp64: median {med:.2e} worst {max(cond.values()):.2e} | HIP vs fp64: median "
                  f"{np.median(list(mine.values())):.2e} worst {max(mine.values()):.2e} | worst error / bound {mine[worst] / bound[worst]:.2f} ({worst}: "
                  f"{mine[worst]:.2e} of {bound[worst]:.2e}) | worst HIP / comparator {max(mine[k] / max(cond[k], 1e-12) for k in keys):.2f}")
    if TABLE:
        with open(TABLE, "a") as f:
            f.write(f"# {report[0]} | group {name}: median of the comparator's deviations {med:.3e}\n")
            f.write(f"# {'tensor':<34} {'elements':>9} {'scale (max |g64|)':>18} {'torch fp32 vs fp64':>19} {'HIP vs fp64':>12} {'HIP / torch':>12} {'HIP / median':>13} {'bound':>10} {'HIP / bound':>12}\n")
            for k in sorted(keys, key=lambda k: -mine[k] / bound[k]):
                f.write(f"  {k:<34} {grads64[k].numel():9d} {_scale64(k, grads64):18.3e} {cond[k]:19.3e} {mine[k]:12.3e} {mine[k] / max(cond[k], 1e-30):12.2f} "
                        f"{mine[k] / max(med, 1e-30):13.2f} {bound[k]:10.3e} {mine[k] / bound[k]:12.2f}\n")
    for k in keys:
        assert mine[k] <= bound[k], f"{name}.{k}: {mine[k]:.3e} of scale from the float64 gradient, bound {bound[k]:.3e} (comparator's own fp32: {cond[k]:.3e})"
    return mine[worst] / bound[worst]


def _oracle_step64(pc, pf, wp, lat, image, intr, ray_idx, u, S, Sf, depth_range, param, alpha, ga, w3, wv, pose_init=None):
    """the oracle's INN train step in float64 on the device of its inputs: parameters and inputs cast, the fp32 band tables kept, the
    un-warped points formed in fp32 and cast -- exactly how tests/golden/make_golden_dtu_fp64.py runs the REFERENCE in float64
    (tests/test_oracle_golden.py pins that evaluation of the oracle to the reference's float64 gradients to 4e-8)
    -> {group: {key: gradient}}"""
    dt = torch.float64
    d = lambda prm: {k: v.detach().to(dt).requires_grad_(True) for k, v in prm.items()}
    pc64, wp64 = d(pc), d(wp)
    pf64 = d(pf) if pf is not None else None
    lat64 = lat.detach().to(dt).requires_grad_(True)
    c0, g0 = O.unwarped_center_and_grid(H, W, intr, ray_idx, pose_init)
    ray, center, grid3 = O.warped_rays(wp64, lat64, c0.to(dt), g0.to(dt), alpha, True)
    kw = {} if w3 is None else dict(w3d=w3.to(dt).to(image.device), wview=wv.to(dt).to(image.device))
    out = O.render_rays(pc64, center, ray, u.to(dt), S, depth_range, param, p_fine=pf64, Sf=Sf, **kw)
    target = O.gather_pixels(image.to(dt), ray_idx)
    total = O.mse_loss(out["rgb"], target)
    if pf64 is not None:
        total = total + O.mse_loss(out["rgb_fine"], target)
    if ga is not None:
        total = total + (10.0 ** ga) * O.global_alignment_loss(g0.to(dt), c0.to(dt), grid3, center)[0]
    total.backward()
    g = lambda prm: {k: v.grad for k, v in prm.items()}
    groups = dict(nerf=g(pc64), warp=g(wp64), latent=dict(weight=lat64.grad))
    if pf64 is not None:
        groups["nerf_fine"] = g(pf64)
    del out, total
    return groups


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_llff_train_step_at_baseline_shapes_vs_oracle_on_the_gpu(cfg):
    from neural_invertible_warp_amd import configs
    from neural_invertible_warp_amd.model import barf_inn_llff
    from neural_invertible_warp_amd.util import edict
    B, it, progress = 18, 30000, 0.3
    if cfg == "cfg2":
        opt = configs.cfg2_nerf_inn_llff_hier(device=DEV)
        R, S, Sf, ga = 4096 // B, 64, 128, None
        opt.barf_c2f = [0.1, 0.5]                          # exercise the band mask at full size as well (cfg2's yaml leaves it off)
    else:
        opt = configs.cfg3_barf_inn_llff(device=DEV)
        R, S, Sf, ga = 2048 // B, 128, 0, 4
    assert (opt.H, opt.W) == (H, W) and opt.nerf.rand_rays // B == R and opt.nerf.sample_intvs == S
    graph = barf_inn_llff.Graph(opt).attach_warp(opt, B)
    pc, pf, wp, lat = _dev_params(1, 3, 4, B, seed_f=2 if Sf else None)
    _load(graph.nerf, pc); _load(graph.warp_mlp, wp)
    graph.nerf.set_progress(progress)
    if Sf:
        _load(graph.nerf_fine, pf)
        graph.nerf_fine.set_progress(progress)
    with torch.no_grad():
        graph.warp_latent.weight.copy_(lat)
    image, intr, u, ray_idx = _inputs(B, R, S, seed=17)
    var = edict(idx=torch.arange(B), image=image, intr=intr)
    with _Rng(u, ray_idx):
        var = graph.forward(opt, var, mode="train", iter=it)
    loss = graph.compute_loss(opt, var, mode="train")
    total = loss.render + (loss.render_fine if Sf else 0) + ((10.0 ** ga) * loss.global_alignment if ga is not None else 0)
    total.backward()

    w3, wv = O.c2f_weights(progress, (0.1, 0.5), 10), O.c2f_weights(progress, (0.1, 0.5), 4)
    ref = O.inn_train_step(pc, wp, lat, image, intr, ray_idx, u, H, W, S, (1, 0), "inverse", it / opt.inn.real_nvp.max_pe_iter,
                           nerf_fine_p=pf, Sf=Sf, ga_weight=ga, w3d=w3, wview=wv)
    ref["loss"].backward()
    report = [f"{cfg}: {B} x {R} x {S}" + (f" + {S + Sf}" if Sf else "") + f" = {B * R * (S + (S + Sf if Sf else 0))} evaluations"]
    torch.testing.assert_close(var.center, ref["center"], atol=2e-5, rtol=0)
    torch.testing.assert_close(var.grid_3D, ref["grid_3D"], atol=2e-5, rtol=0)
    torch.testing.assert_close(var.rgb, ref["rgb"], atol=3e-5, rtol=2e-4)
    torch.testing.assert_close(var.opacity, ref["opacity"], atol=3e-5, rtol=2e-4)
    report.append(f"max |rgb - oracle| {float((var.rgb - ref['rgb']).abs().max()):.2e}")
    assert abs(float(loss.render.detach()) - float(ref["loss_render"].detach())) < 1e-6
    if Sf:
        # (the inverse-CDF positions come from a cumulative sum -- here the sequential fp64 sum of the CPU reference, in torch on the
        # GPU a parallel fp32 scan -- and still the fine image agrees at the coarse tolerance: measured 5.9e-6)
        torch.testing.assert_close(var.rgb_fine, ref["rgb_fine"], atol=3e-5, rtol=2e-4)
        report.append(f"max |rgb_fine - oracle| {float((var.rgb_fine - ref['rgb_fine']).abs().max()):.2e}")
        assert abs(float(loss.render_fine.detach()) - float(ref["loss_render_fine"].detach())) < 1e-6
    if ga is not None:
        assert abs(float(loss.global_alignment.detach()) - float(ref["loss_ga"].detach())) <= 1e-4 * float(ref["loss_ga"].detach()) + 1e-9
    # gradients: every optimizer group against the oracle in FLOAT64 at these shapes, bounded by the comparator's own fp32 conditioning
    ref64 = _oracle_step64(pc, pf, wp, lat, image, intr, ray_idx, u, S, Sf, (1, 0), "inverse", it / opt.inn.real_nvp.max_pe_iter, ga, w3, wv)
    grads = lambda named: {k: p.grad for k, p in named}
    tgrads = lambda prm: {k: v.grad for k, v in prm.items()}
    report.append(f"gradients vs the oracle in float64 (bound per tensor = max({FACTOR_TENSOR} x the comparator's own fp32 deviation on it, {FACTOR_GROUP} x the group's median) + {FLOOR}):")
    ratios = [_check_group_vs_fp64("nerf", grads(graph.nerf.named_parameters()), tgrads(pc), ref64["nerf"], report)]
    if Sf:
        ratios.append(_check_group_vs_fp64("nerf_fine", grads(graph.nerf_fine.named_parameters()), tgrads(pf), ref64["nerf_fine"], report))
    ratios.append(_check_group_vs_fp64("warp_mlp", grads(graph.warp_mlp.named_parameters()), tgrads(wp), ref64["warp"], report))
    ratios.append(_check_group_vs_fp64("warp_latent", dict(weight=graph.warp_latent.weight.grad), dict(weight=lat.grad), ref64["latent"], report))
    print("\n".join(report))


def test_dtu_train_step_at_baseline_shape_vs_oracle_on_the_gpu():
    """cfg5: barf_inn_dtu.Graph + INNPoseParams, 3 x 682 x 128, metric depth from the data range, un-warped rays taken to the world
    frame with the noisy initial poses (pose_models/inn.py:63-93), c2f mask at progress 0.3 as shipped (scripts/train_dtu.sh:6)."""
    from neural_invertible_warp_amd import configs, engine
    from neural_invertible_warp_amd.model import barf_inn_dtu
    from neural_invertible_warp_amd.model.pose_models.inn import INNPoseParams
    from neural_invertible_warp_amd.util import edict
    B, it, progress = 3, 30000, 0.3
    opt = configs.cfg5_barf_inn_dtu(device=DEV)
    R, S = opt.nerf.rand_rays // B, opt.nerf.sample_intvs
    assert (R, S) == (682, 128)
    var0, init = engine.synthetic_dtu_scene(opt, B)
    pose_net = INNPoseParams(opt, num_poses=B, initial_poses_w2c=init, device=DEV)
    pc, _, wp, lat = _dev_params(21, 23, 24, B)
    _load(pose_net.pose_embedding, wp)
    with torch.no_grad():
        pose_net.pose_latent.weight.copy_(lat)
    graph = barf_inn_dtu.Graph(opt, pose_net)
    _load(graph.nerf, pc)
    graph.nerf.set_progress(progress)
    image, intr, u, ray_idx = _inputs(B, R, S, seed=29)
    var = edict(idx=torch.arange(B), image=image, intr=intr, pose=var0.pose, depth_range=var0.depth_range)
    with _Rng(u, ray_idx):
        var = graph.forward(opt, var, mode="train", iter=it)
    loss = graph.compute_loss(opt, var, mode="train")
    loss.render.backward()
    c2f = opt.barf_c2f
    w3 = O.c2f_weights(progress, c2f, 10) if c2f else None
    wv = O.c2f_weights(progress, c2f, 4) if c2f else None
    ref = O.inn_train_step(pc, wp, lat, image, intr, ray_idx, u, H, W, S, (1.2, 5.2), "metric", it / opt.inn.real_nvp.max_pe_iter,
                           pose_init=init, w3d=w3, wview=wv)
    ref["loss_render"].backward()
    torch.testing.assert_close(var.center_local, ref["center"], atol=2e-5, rtol=0)
    torch.testing.assert_close(var.grid_local, ref["grid_3D"], atol=2e-5, rtol=0)
    torch.testing.assert_close(var.rgb, ref["rgb"], atol=3e-5, rtol=2e-4)
    torch.testing.assert_close(var.opacity, ref["opacity"], atol=3e-5, rtol=2e-4)
    assert abs(float(loss.render.detach()) - float(ref["loss_render"].detach())) < 1e-6
    report = [f"cfg5: {B} x {R} x {S}, max |rgb - oracle| {float((var.rgb - ref['rgb']).abs().max()):.2e}"]
    # gradients against the oracle in FLOAT64 at the full shape (the pose network's gradients at world-scale inputs scatter at the
    # percent level between ANY two fp32 evaluations: rounds 1-3 held this test to 5e-2 against torch's fp32 kernels)
    ref64 = _oracle_step64(pc, None, wp, lat, image, intr, ray_idx, u, S, 0, (1.2, 5.2), "metric", it / opt.inn.real_nvp.max_pe_iter, None, w3, wv,
                           pose_init=init)
    grads = lambda named: {k: p.grad for k, p in named}
    tgrads = lambda prm: {k: v.grad for k, v in prm.items()}
    report.append(f"gradients vs the oracle in float64 (bound per tensor = max({FACTOR_TENSOR} x the comparator's own fp32 deviation on it, {FACTOR_GROUP} x the group's median) + {FLOOR}):")
    _check_group_vs_fp64("nerf", grads(graph.nerf.named_parameters()), tgrads(pc), ref64["nerf"], report)
    _check_group_vs_fp64("pose_embedding", grads(pose_net.pose_embedding.named_parameters()), tgrads(wp), ref64["warp"], report)
    _check_group_vs_fp64("pose_latent", dict(weight=pose_net.pose_latent.weight.grad), dict(weight=lat.grad), ref64["latent"], report)
    print("\n".join(report))
