"""The engine's one-call train iteration (niw_train_step, csrc/niw_step.hip; engine.FusedStep) against the autograd mirror of the
reference's call sequence (Graph.forward + compute_loss + backward over the per-stage entry points, `fused_step=False`): the two run
the same kernels on the same operands and sum the gradient routes in the same order, so losses, gradients and trained parameters
must be IDENTICAL -- unsharded and as one rank of a sharded job, LLFF and DTU, with and without the fine network, eager and as a
captured graph.  The mirror itself is held to the oracle by tests/test_gpu_parity.py and tests/test_gpu_baseline_shapes.py.
Needs a GPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _trainer(cfg, fused, rank=0, world=1, hip_graph=False, stratified=True, B=None, rays=None, S=16, ndc=False, noise=None, bg=None):
    from neural_invertible_warp_amd import configs, engine
    if cfg == "dtu":
        opt = configs.cfg5_barf_inn_dtu(device=DEV)
        B = B or 3
        var0, init = engine.synthetic_dtu_scene(opt, B)
    else:
        opt = getattr(configs, cfg)(device=DEV)
        B = B or 5
        var0, init = engine.synthetic_scene(opt, B), None
    opt.nerf.sample_stratified = stratified
    opt.camera.ndc, opt.nerf.density_noise_reg = ndc, noise
    if bg is not None:
        opt.nerf.setbg_opaque, opt.data.bgcolor = True, bg
    opt.nerf.rand_rays, opt.nerf.sample_intvs, opt.max_iter = rays or B * 37, S, 40
    opt.inn.real_nvp.max_pe_iter = 20                       # bands, windows and learning rates all move within the run
    if opt.nerf.fine_sampling:
        opt.nerf.sample_intvs_fine = S
    tr = engine.INNTrainer(opt, B, rank=rank, world=world, warp_perturb=0.02, seed=4, initial_poses_w2c=init, hip_graph=hip_graph, fused_step=fused)
    assert (tr.fused is not None) == bool(fused), tr.fused_fallback_reason
    return tr, var0


def _run(cfg, fused, steps=5, **kw):
    tr, var0 = _trainer(cfg, fused, **kw)
    losses = []
    for _ in range(steps):
        loss = tr.train_iteration(type(var0)(var0))
        losses.append({k: float(v.detach()) for k, v in loss.items()})
    torch.cuda.synchronize()
    return tr, losses


@pytest.mark.parametrize("cfg", ["cfg3_barf_inn_llff", "cfg2_nerf_inn_llff_hier", "dtu"])
def test_fused_iteration_trains_bit_for_bit_like_the_autograd_mirror(cfg):
    a, la = _run(cfg, False)
    b, lb = _run(cfg, True)
    assert [sorted(x) for x in la] == [sorted(x) for x in lb]
    for x, y in zip(la, lb):
        for k in x:
            if k == "all":          # the weighted total: the mirror forms it with torch.add(alpha=), the call with one fma per term
                assert abs(x[k] - y[k]) <= 1e-6 * max(abs(x[k]), 1e-6), (k, x[k], y[k])
            else:
                assert x[k] == y[k], (k, x[k], y[k])
    assert torch.equal(a.bucket.flat, b.bucket.flat), float((a.bucket.flat - b.bucket.flat).abs().max())
    for fa, fb in zip(a._flats(), b._flats()):
        assert torch.equal(fa, fb)
    for ma, mb in zip(a.m + a.v, b.m + b.v):
        assert torch.equal(ma, mb)
    # the registered per-view poses (global_rigid / pose_global) are refreshed by both
    ta = a.pose_net.pose_global.weight if cfg == "dtu" else a.graph.global_rigid.weight
    tb = b.pose_net.pose_global.weight if cfg == "dtu" else b.graph.global_rigid.weight
    if cfg != "cfg2_nerf_inn_llff_hier":                    # (no alignment term there: the LLFF table keeps its initial identity poses)
        assert not torch.equal(tb.data, torch.eye(3, 4, device=DEV).reshape(1, 12).repeat(tb.shape[0], 1))
    assert torch.equal(ta.data, tb.data)


@pytest.mark.parametrize("cfg,world", [("cfg3_barf_inn_llff", 3), ("cfg2_nerf_inn_llff_hier", 2), ("dtu", 4)])
def test_fused_iteration_of_a_rank_equals_the_mirror_under_ray_sharding(cfg, world):
    """every rank of a sharded job (run one after the other, no process group): window of whole views, contiguous share of the rays,
    alignment terms of the owned views, zero latent rows outside the window"""
    for rank in range(world):
        outs = []
        for fused in (False, True):
            tr, var0 = _trainer(cfg, fused, rank=rank, world=world, stratified=False)
            tr.it = 7
            for n in tr.nets:
                n.set_progress(0.3)
            # the latent columns of the warp's first layers are zero-initialised (nvp_ndr.py:278-282), which makes the latent gradient
            # exactly zero at the first step: move every warp parameter off its initial value
            with torch.no_grad():
                tr.warp_mlp.flat_params.add_(0.01 * torch.randn(tr.warp_mlp.flat_params.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9)))
            loss = tr._forward_backward(type(var0)(var0), tr.it)
            outs.append(({k: float(v.detach()) for k, v in loss.items()}, tr.bucket.flat.clone(), tr))
        (la, ga, ta), (lb, gb, tb) = outs
        for k in la:
            assert abs(la[k] - lb[k]) <= 1e-6 * max(abs(la[k]), 1e-6), (rank, k, la[k], lb[k])
        assert torch.equal(ga, gb), (rank, float((ga - gb).abs().max()))
        n_nets = len(ta.nets)
        lat = gb[tb.bucket.starts[n_nets + 1]:tb.bucket.starts[n_nets + 1] + tb.bucket.sizes[n_nets + 1]].view(-1, 128)
        win = tb.graph._last_window
        assert bool((lat[:win.v0] == 0).all()) and bool((lat[win.v1:] == 0).all()) and float(lat[win.v0:win.v1].abs().max()) > 0


def test_fused_iteration_stage_by_stage_equals_the_single_call():
    """bench.py's kernel table times the stages of the call one by one (niw_train_step(stage, stage + 1)): same numbers"""
    from neural_invertible_warp_amd import ops
    a, la = _run("cfg2_nerf_inn_llff_hier", True, steps=3)
    ops.TIMING.enabled = True
    ops.TIMING.reset()
    try:
        b, lb = _run("cfg2_nerf_inn_llff_hier", True, steps=3)
    finally:
        ops.TIMING.enabled = False
    table = ops.TIMING.summary()
    assert la == lb and torch.equal(a.bucket.flat, b.bucket.flat)
    # (round 5: compositing + photometric residual + their backward are ONE launch per pass -- its stage is listed as composite_train and the
    # pass's composite_bwd stage, which launches nothing, is not listed)
    for name in ("mlp_fwd_train", "mlp_bwd_dx", "mlp_bwd_dw", "composite_train", "front", "warp_fwd", "warp_bwd", "loss", "adam"):
        assert name in table and table[name][1] > 0, name
    assert "composite_bwd" not in table and "composite_fwd" not in table
    assert table["mlp_fwd_train"][0] == 6 and table["resample"][0] == 3 and table["composite_train"][0] == 6          # coarse + fine per step


@pytest.mark.parametrize("cfg", ["cfg2_nerf_inn_llff_hier", "dtu"])
def test_fused_iteration_with_and_without_the_second_stream(cfg):
    """niw_train_desc.overlap: the small independent stages on the library's second stream beside the field-MLP chain -- same kernels,
    same numbers, eager and captured"""
    from neural_invertible_warp_amd import engine
    runs = []
    for overlap, hip_graph in ((False, False), (True, False), (True, True)):
        tr, var0 = _trainer(cfg, True, hip_graph=hip_graph)
        tr.overlap = overlap
        losses = [{k: float(v.detach()) for k, v in tr.train_iteration(type(var0)(var0)).items()} for _ in range(6)]
        torch.cuda.synchronize()
        assert tr.fused.desc.overlap == int(overlap)
        runs.append((losses, [f.clone() for f in tr._flats()]))
    for losses, flats in runs[1:]:
        assert losses == runs[0][0]
        for x, y in zip(flats, runs[0][1]):
            assert torch.equal(x, y)


def test_fused_iteration_captured_graph_equals_eager():
    a, la = _run("cfg3_barf_inn_llff", True, steps=7)
    b, lb = _run("cfg3_barf_inn_llff", True, steps=7, hip_graph=True)
    assert b._captured is not None
    assert la == lb
    for fa, fb in zip(a._flats(), b._flats()):
        assert torch.equal(fa, fb)


def test_captured_iteration_survives_sync_state_under_ray_sharding():
    """Round-4 advisor finding: a captured iteration has the pose table's device address baked in (`d.poses`); sync_state() of a sharded
    job used to REPLACE the table's storage with the gathered rows, after which every replay wrote the registered poses into freed memory
    and validation / checkpoints read a stale table.  Rank 0 of a 2-way sharded job with a live (one-rank, forced) process group: capture,
    sync_state between replays -- the table must stay where it is, and equal the launched (hip_graph=False) trainer's after every sync."""
    import os
    import torch.distributed as dist
    from neural_invertible_warp_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29677")
    dist.init_process_group(backend="gloo", rank=0, world_size=1)
    parallel.FORCE_COLLECTIVES = True
    try:
        tables = {}
        for graph in (False, True):
            tr, var0 = _trainer("cfg3_barf_inn_llff", True, rank=0, world=2, hip_graph=graph)
            table = tr.graph.global_rigid.weight
            ptr = table.data.data_ptr()
            seen = []
            for k in range(8):
                tr.train_iteration(type(var0)(var0))
                if k in (3, 5, 7):
                    tr.sync_state()
                    assert table.data.data_ptr() == ptr, "sync_state moved the pose table"
                    seen.append(table.data.clone())
            torch.cuda.synchronize()
            if graph:
                assert tr._captured is not None
            tables[graph] = seen
        win = parallel.ViewWindow(5, 37, 0, 2)
        for a, b in zip(tables[False], tables[True]):
            assert torch.equal(a, b)
            assert float(a[win.own0:win.own1].abs().sum()) > 0 and float(a[win.own1:].abs().sum()) == 0      # owned rows registered, the others summed as zeros
        assert not torch.equal(tables[True][0], tables[True][-1])            # the replays after a sync_state still refresh the live table
    finally:
        parallel.FORCE_COLLECTIVES = False
        dist.destroy_process_group()


def test_fused_iteration_is_refused_or_bypassed_where_it_does_not_apply():
    from neural_invertible_warp_amd import configs, engine
    from neural_invertible_warp_amd._lib import NiwError
    # (NDC, density noise and an opaque background are inside the call since round 6; draws injected through torch are not)
    opt = configs.cfg3_barf_inn_llff(device=DEV)
    opt.nerf.density_noise_reg, opt.nerf.density_noise_rng = 1.0, "torch"      # a harness that injects torch.randn draws
    tr = engine.INNTrainer(opt, 3, fused_step="auto")
    assert tr.fused is None and "torch.randn" in tr.fused_fallback_reason
    with pytest.raises(NiwError, match="does not cover"):
        engine.INNTrainer(configs.cfg3_barf_inn_llff(device=DEV), 3, ray_sampler="randperm", fused_step=True)


def test_adam_of_all_groups_in_one_launch_equals_the_per_group_launches():
    from neural_invertible_warp_amd import ops
    gen = torch.Generator(device=DEV).manual_seed(1)
    sizes = [530052, 165900, 18 * 128, 7]
    mk = lambda: [torch.randn(n, device=DEV, generator=gen) for n in sizes]
    p, g, m, v = mk(), mk(), mk(), [x.abs() for x in mk()]
    p2, m2, v2 = [x.clone() for x in p], [x.clone() for x in m], [x.clone() for x in v]
    lrs, step = [1e-3, 5e-4, 5e-4, 2e-3], 37
    for k in range(4):
        if k != 2:
            ops.adam_step(p[k], g[k], m[k], v[k], lrs[k], step)
    ops.adam_step_multi([None if k == 2 else (p2[k], g[k], m2[k], v2[k], lrs[k], step) for k in range(4)])
    for k in range(4):
        assert torch.equal(p[k], p2[k]) and torch.equal(m[k], m2[k]) and torch.equal(v[k], v2[k]), k
    # step scalars read from device memory (graph replays): [groups][2]
    hyper = torch.tensor([ops.adam_hyper(lr, step + 1) for lr in lrs], device=DEV, dtype=torch.float32)
    for k in range(4):
        ops.adam_step(p[k], g[k], m[k], v[k], lrs[k], step + 1)
    ops.adam_step_multi([(p2[k], g[k], m2[k], v2[k], 123.0, 1) for k in range(4)], hyper_dev=hyper)
    for k in range(4):
        assert torch.equal(p[k], p2[k]), k


# ------------------------------------------------------------------------------------------------ the vanilla model (BASELINE configs[0])
def _vanilla_trainer(fused, ndc=False, noise=1.0, B=5, S=16):
    """options/nerf_llff_repr.yaml at a small shape: ground-truth cameras (a few degrees / centimetres off identity, so that world and
    camera frames differ), ReLU density with density noise, coarse + fine pass (reference model/nerf.py:251-288)"""
    from neural_invertible_warp_amd import camera, configs, engine
    opt = configs.cfg1_nerf_llff_repr(device=DEV)
    opt.nerf.rand_rays, opt.nerf.sample_intvs, opt.nerf.sample_intvs_fine, opt.max_iter = B * 37, S, S, 40
    opt.camera.ndc, opt.nerf.density_noise_reg = ndc, noise
    var0 = engine.synthetic_scene(opt, B)
    gen = torch.Generator().manual_seed(3)
    var0.pose = camera.lie.se3_to_SE3(torch.randn(B, 6, generator=gen) * 0.03).to(DEV)[:, :3].contiguous()
    tr = engine.NeRFTrainer(opt, B, seed=4, fused_step=fused)
    assert (tr.fused is not None) == bool(fused), tr.fused_fallback_reason
    return tr, var0


@pytest.mark.parametrize("ndc,noise", [(False, 1.0), (True, 1.0), (False, None)])
def test_vanilla_iteration_trains_bit_for_bit_like_the_autograd_mirror(ndc, noise):
    """round 6: niw_train_step with warp_params = NULL -- rays of the given cameras (niw_raygen), NDC re-parametrisation, density noise
    drawn by niw_normal_rng in both passes, no ray-gradient tail of the dX chain -- against Graph.render + compute_loss + backward of the
    mirror (engine.NeRFTrainer(fused_step=False)), which draws its noise from the same keyed streams"""
    runs = []
    for fused in (False, True):
        tr, var0 = _vanilla_trainer(fused, ndc=ndc, noise=noise)
        losses = []
        for _ in range(5):
            loss = tr.train_iteration(type(var0)(var0))
            losses.append({k: float(v.detach()) for k, v in loss.items()})
        torch.cuda.synchronize()
        runs.append((tr, losses))
    (a, la), (b, lb) = runs
    assert [sorted(x) for x in la] == [sorted(x) for x in lb] and sorted(la[0]) == ["all", "render", "render_fine"]
    for x, y in zip(la, lb):
        for k in x:
            if k == "all":
                assert abs(x[k] - y[k]) <= 1e-6 * max(abs(x[k]), 1e-6), (k, x[k], y[k])
            else:
                assert x[k] == y[k], (k, x[k], y[k])
    assert la[0]["render"] != la[1]["render"]
    assert torch.equal(a.bucket.flat, b.bucket.flat), float((a.bucket.flat - b.bucket.flat).abs().max())
    for fa, fb in zip(a._flats() + a.m + a.v, b._flats() + b.m + b.v):
        assert torch.equal(fa, fb)


def test_vanilla_iteration_with_density_noise_vs_oracle():
    """the same iteration against the oracle (reference model/nerf.py:251-288, 416-447): the pixel draw, the stratified draws and the
    two noise tensors of the iteration are read back from the keyed streams (ops.draw_ray_idx / sample_stratified_rng / normal_rng) and
    handed to the oracle as explicit tensors"""
    from neural_invertible_warp_amd import engine, ops
    from oracle import niw_oracle as O
    tr, var0 = _vanilla_trainer(True, noise=0.5)
    opt, B = tr.opt, 5
    R, S, Sf = opt.nerf.rand_rays // B, opt.nerf.sample_intvs, opt.nerf.sample_intvs_fine
    clone = lambda mod: {k: v.detach().cpu().clone().requires_grad_(True) for k, v in mod.state_dict().items() if k != "progress"}
    pc, pf = clone(tr.graph.nerf), clone(tr.graph.nerf_fine)
    loss = tr.train_iteration(type(var0)(var0))
    grads = tr.bucket.flat.detach().cpu().clone()
    # the iteration's draws (iteration 0 -> draw 1)
    ray_idx = ops.draw_ray_idx(opt.H * opt.W, R, 0, 1, DEV).cpu()
    seed_d = (0x5D1F) & (2 ** 64 - 1)
    _, u = ops.sample_stratified_rng(seed_d, 1, B * R, S, opt.nerf.depth.range, opt.nerf.depth.param, DEV, return_u=True)
    n_c = ops.normal_rng(engine.noise_stream_seed(opt, 0, 0), 1, B * R * S, 0.5, DEV).view(B, R, S).cpu()
    n_f = ops.normal_rng(engine.noise_stream_seed(opt, 0, 1), 1, B * R * (S + Sf), 0.5, DEV).view(B, R, S + Sf).cpu()
    # The density is a ReLU of (raw + noise): a sample whose pre-activation sits within fp32 round-off of zero switches its whole gradient
    # path on or off, so ANY fp32 evaluation deviates from the exact gradient by a few such samples.  The yardstick is therefore the
    # oracle in float64, and the bound what torch's own fp32 evaluation of the same function shows against it (round 4's criterion).
    def oracle(dtype):
        cast = lambda d: {k: v.detach().to(dtype).requires_grad_(True) for k, v in d.items()}
        qc, qf = cast(pc), cast(pf)
        center, ray = (x.to(dtype) for x in O.center_and_ray(opt.H, opt.W, var0.pose.cpu(), var0.intr.cpu()))     # (rays are data here: fp32 values)
        out = O.render_rays(qc, center[:, ray_idx], ray[:, ray_idx], u.view(B, R, S, 1).cpu().to(dtype), S, tuple(opt.nerf.depth.range), opt.nerf.depth.param,
                            p_fine=qf, Sf=Sf, density_activ="relu", density_noise=n_c.to(dtype), density_noise_fine=n_f.to(dtype))
        target = O.gather_pixels(var0.image.cpu().to(dtype), ray_idx)
        l_c, l_f = O.mse_loss(out["rgb"], target), O.mse_loss(out["rgb_fine"], target)
        (l_c + l_f).backward()
        return float(l_c.detach()), float(l_f.detach()), qc, qf

    l_c, l_f, pc32, pf32 = oracle(torch.float32)
    l_c64, l_f64, pc64, pf64 = oracle(torch.float64)
    assert abs(float(loss.render.detach()) - l_c) <= 2e-6 and abs(float(loss.render_fine.detach()) - l_f) <= 2e-6, (float(loss.render), l_c, float(loss.render_fine), l_f)
    assert abs(float(loss.render.detach()) - l_c64) <= 2e-6 and abs(float(loss.render_fine.detach()) - l_f64) <= 2e-6
    off, worst, rows = 0, (0.0, 0.0, None), []
    for net, p32, p64 in ((tr.graph.nerf, pc32, pc64), (tr.graph.nerf_fine, pf32, pf64)):
        for k, v in net.state_dict().items():
            if k == "progress":
                continue
            g_hip, g64, g32 = grads[off:off + v.numel()].view(v.shape).double(), p64[k].grad, p32[k].grad.double()
            off += v.numel()
            scale = max(float(g64.abs().max()), 1e-12)
            err_hip, err_t32 = float((g_hip - g64).abs().max()) / scale, float((g32 - g64).abs().max()) / scale
            if err_hip > worst[0]:
                worst = (err_hip, err_t32, k)
            rows.append((k, err_hip, err_t32, scale))
    for k, e, t, sc in rows:
        print(f"   {k:22s} HIP {e:.2e}  torch fp32 {t:.2e}  max |g| {sc:.2e}")
    # one sample whose pre-activation flips a ReLU (the density's, or a hidden unit's) moves a tensor's gradient by ~1e-2 of its maximum at
    # this batch size -- torch's own fp32 shows exactly that against float64 on the fine network (measured 1.8e-2), on tensors of its own
    # choosing.  Per tensor: 3 x the larger of torch's deviation on it and torch's worst deviation anywhere, + 2e-4; and the MEDIAN tensor
    # (no flip) must be as accurate as torch's median to within the same factor.
    import statistics
    t_worst = max(t for _, _, t, _ in rows)
    for k, e, t, sc in rows:
        assert e <= 3 * max(t, t_worst) + 2e-4, (k, e, t, t_worst)
    half = len(rows) // 2                                  # (coarse network's tensors, then the fine network's)
    for part in (rows[:half], rows[half:]):
        assert statistics.median(e for _, e, _, _ in part) <= 3 * statistics.median(t for _, _, t, _ in part) + 2e-4
    print(f"vanilla step with density noise: losses to 2e-6; worst gradient tensor {worst[2]}: HIP {worst[0]:.2e} of max vs float64, torch fp32 {worst[1]:.2e}")


@pytest.mark.parametrize("cfg,world,noise", [("cfg3_barf_inn_llff", 1, None), ("cfg2_nerf_inn_llff_hier", 1, 0.5), ("cfg3_barf_inn_llff", 3, None)])
def test_fused_iteration_with_ndc_behind_the_warp_equals_the_mirror(cfg, world, noise):
    """round 6: camera.ndc with WARPED rays inside niw_train_step -- the rays are re-parametrised behind the warp (niw_convert_ndc) and the
    summed gradient routes go back through niw_convert_ndc_bwd before they reach the warp, the launch the mirror's autograd makes
    (ops._ConvertNDC) -- unsharded, with the fine pass (there also with density noise in both passes, nerf.py:428-429: the INN models take
    it from the same keyed streams as the vanilla one), and as every rank of a three-rank job"""
    for rank in range(world):
        runs = []
        for fused in (False, True):
            tr, var0 = _trainer(cfg, fused, rank=rank, world=world, ndc=True, stratified=world == 1, noise=noise)
            if world > 1:
                with torch.no_grad():
                    g = torch.Generator().manual_seed(11)
                    tr.warp_mlp.flat_params.add_(0.01 * torch.randn(tr.warp_mlp.flat_params.shape, generator=g).to(DEV))
            losses = []
            for _ in range(3):
                loss = tr.train_iteration(type(var0)(var0))
                losses.append({k: float(v.detach()) for k, v in loss.items()})
            torch.cuda.synchronize()
            runs.append((tr, losses))
        (a, la), (b, lb) = runs
        for x, y in zip(la, lb):
            for k in x:
                if k == "all":
                    assert abs(x[k] - y[k]) <= 1e-6 * max(abs(x[k]), 1e-6), (k, x[k], y[k])
                else:
                    assert x[k] == y[k], (rank, k, x[k], y[k])
        assert torch.equal(a.bucket.flat, b.bucket.flat), (rank, float((a.bucket.flat - b.bucket.flat).abs().max()))
        for fa, fb in zip(a._flats() + a.m + a.v, b._flats() + b.m + b.v):
            assert torch.equal(fa, fb)
        # the warp does receive a gradient through the re-parametrisation
        n_nets = len(a.nets)
        assert float(a.bucket.segment(n_nets).abs().max()) > 0


def test_ndc_reverse_pass_vs_autograd_of_the_reference_formulas():
    """niw_convert_ndc_bwd against torch autograd of the oracle's convert_ndc (reference camera.py:523-540) in float64, on rays in front of
    the cameras; and the forward against the same"""
    from neural_invertible_warp_amd import ops
    from oracle import niw_oracle as O
    gen = torch.Generator().manual_seed(5)
    B, R = 3, 257
    center = (torch.randn(B, R, 3, generator=gen) * 0.2)
    ray = torch.randn(B, R, 3, generator=gen) * 0.5
    ray[..., 2] = ray[..., 2].abs() + 0.5                        # +z forward
    intr = torch.tensor([[320.0, 0, 200.0], [0, 300.0, 150.0], [0, 0, 1]]).repeat(B, 1, 1) * torch.linspace(0.9, 1.1, B)[:, None, None]
    intr[:, 2, 2] = 1.0
    gc, gr = torch.randn(B, R, 3, generator=gen), torch.randn(B, R, 3, generator=gen)
    c64, r64 = center.double().requires_grad_(True), ray.double().requires_grad_(True)
    oc64, or64 = O.convert_ndc(c64, r64, intr.double())
    ((oc64 * gc.double()).sum() + (or64 * gr.double()).sum()).backward()
    cg, rg = center.to(DEV).requires_grad_(True), ray.to(DEV).requires_grad_(True)
    oc, orr = ops.convert_ndc(cg, rg, intr.to(DEV))
    ((oc * gc.to(DEV)).sum() + (orr * gr.to(DEV)).sum()).backward()
    rel = lambda x, y: float((x.detach().cpu().double() - y.detach()).abs().max() / y.detach().abs().max())
    assert rel(oc, oc64) < 2e-6 and rel(orr, or64) < 2e-6
    assert rel(cg.grad, c64.grad) < 5e-6 and rel(rg.grad, r64.grad) < 5e-6, (rel(cg.grad, c64.grad), rel(rg.grad, r64.grad))


@pytest.mark.parametrize("cfg", ["cfg3_barf_inn_llff", "cfg2_nerf_inn_llff_hier"])
def test_fused_iteration_with_an_opaque_background_equals_the_mirror(cfg):
    """nerf.setbg_opaque (reference model/nerf.py:470-472: rgb += bgcolor (1 - opacity)) inside niw_train_step: the passes then run as
    niw_composite_fwd / niw_mse_fwd_bwd / niw_composite_bwd with the background, the launches the mirror makes"""
    runs = []
    for fused in (False, True):
        tr, var0 = _trainer(cfg, fused, bg=1.0)
        losses = []
        for _ in range(3):
            loss = tr.train_iteration(type(var0)(var0))
            losses.append({k: float(v.detach()) for k, v in loss.items()})
        torch.cuda.synchronize()
        runs.append((tr, losses))
    (a, la), (b, lb) = runs
    for x, y in zip(la, lb):
        for k in x:
            if k == "all":
                assert abs(x[k] - y[k]) <= 1e-6 * max(abs(x[k]), 1e-6), (k, x[k], y[k])
            else:
                assert x[k] == y[k], (k, x[k], y[k])
    assert torch.equal(a.bucket.flat, b.bucket.flat), float((a.bucket.flat - b.bucket.flat).abs().max())
    for fa, fb in zip(a._flats() + a.m + a.v, b._flats() + b.m + b.v):
        assert torch.equal(fa, fb)
    # (with the reference's last interval of 1e10 and a positive density the opacity is exactly 1 and the background term exactly 0:
    # the check is that both forms run the background's launches and agree, not that the loss moves)


def test_captured_iteration_with_density_noise_draws_a_fresh_stream_every_replay():
    """density noise inside a REPLAYED graph: niw_normal_rng reads the draw number from the step constants (draw_dev), so every replay adds
    the noise of ITS iteration -- the captured run must equal the launched one bit for bit over several replays, and two consecutive
    iterations must not see the same noise"""
    runs = []
    for graph in (False, True):
        tr, var0 = _trainer("cfg2_nerf_inn_llff_hier", True, hip_graph=graph, noise=0.5)
        losses = []
        for _ in range(7):
            loss = tr.train_iteration(type(var0)(var0))
            losses.append(float(loss.render_fine.detach()))
        torch.cuda.synchronize()
        if graph:
            assert tr._captured is not None
        runs.append((tr, losses))
    (a, la), (b, lb) = runs
    assert la == lb, (la, lb)
    assert len(set(la)) == len(la)
    for fa, fb in zip(a._flats() + a.m + a.v, b._flats() + b.m + b.v):
        assert torch.equal(fa, fb)
