"""The operations around the MLP that a training step is made of besides the render kernels: fused alignment loss, sort-free
pixel draw, pixel-range ray generation, single-launch losses, device-resident step constants and the HIP-graph replay of a whole
iteration.  Each against the CPU oracle / a torch restatement on the same inputs.  Needs a GPU."""
import numpy as np
import pytest
import torch

from oracle import niw_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _points(B, N, seed, reflect=False):
    gen = torch.Generator().manual_seed(seed)
    src = torch.randn(B, N, 3, generator=gen)
    A = torch.linalg.qr(torch.randn(B, 3, 3, generator=gen))[0]
    if reflect:
        A[:, :, 0] *= torch.sign(torch.det(A))[:, None] * -1            # improper: forces the det(UV^T) = -1 branch
    tgt = src @ A.transpose(1, 2) + torch.randn(B, 1, 3, generator=gen) + 0.1 * torch.randn(B, N, 3, generator=gen)
    return src, tgt


@pytest.mark.parametrize("B,N,reflect", [(18, 226, False), (56, 72, False), (3, 1364, False), (5, 40, True)])
def test_fused_alignment_loss_vs_oracle_autograd_through_svd(B, N, reflect):
    from neural_invertible_warp_amd import ops
    src, tgt = _points(B, N, seed=B + N, reflect=reflect)
    t64 = tgt.double().requires_grad_(True)
    Rg, tg = O.rigid_registration(t64, src.double())
    ref_pose = torch.cat([Rg, tg[..., None]], -1)
    ref = ((t64 - O.cam2world(src.double(), ref_pose)) ** 2).mean()
    ref.backward()                                                        # the reference's route: autograd THROUGH the SVD
    tg_dev = tgt.to(DEV).requires_grad_(True)
    poses = ops.rigid_registration(tg_dev, src.to(DEV))
    loss = ops.alignment_residual(tg_dev, src.to(DEV), poses)
    (3.0 * loss).backward()
    assert torch.allclose(poses.cpu().double(), ref_pose.detach(), atol=2e-6)
    assert abs(float(loss.detach()) - float(ref.detach())) <= 2e-6 * max(1.0, float(ref.detach()))
    g = tg_dev.grad.cpu().double() / 3.0
    assert (g - t64.grad).abs().max() <= 2e-6 * t64.grad.abs().max() + 1e-12
    # rotations: orthonormal with determinant +1, also when the unconstrained optimum is a reflection
    R = poses[:, :, :3].double().cpu()
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3, dtype=torch.float64).expand(B, 3, 3), atol=1e-6)
    assert (torch.det(R) - 1).abs().max() < 1e-6


def test_pixel_draw_is_a_partitioned_duplicate_free_uniform_subset():
    from neural_invertible_warp_amd import ops
    HW = 300 * 400
    a = ops.draw_ray_idx(HW, 2048, seed=0, draw=1, device=DEV).cpu().numpy()
    assert a.min() >= 0 and a.max() < HW and len(set(a.tolist())) == 2048
    # ranks keep idx[rank::world] of the SAME permutation: disjoint, together the unsharded draw
    parts = [ops.draw_ray_idx(HW, len(range(r, 2048, 8)), seed=0, draw=1, device=DEV, first=r, stride=8).cpu().numpy() for r in range(8)]
    for r in range(8):
        assert np.array_equal(parts[r], a[r::8])
    # a fresh subset per draw and per seed; draw_dev overrides the by-value draw number
    b = ops.draw_ray_idx(HW, 2048, seed=0, draw=2, device=DEV).cpu().numpy()
    c = ops.draw_ray_idx(HW, 2048, seed=1, draw=1, device=DEV).cpu().numpy()
    assert len(set(a.tolist()) & set(b.tolist())) < 100 and len(set(a.tolist()) & set(c.tolist())) < 100
    word = torch.tensor([2], dtype=torch.int64, device=DEV)
    assert np.array_equal(ops.draw_ray_idx(HW, 2048, seed=0, draw=999, device=DEV, draw_dev=word).cpu().numpy(), b)
    # the whole permutation: every pixel exactly once (also for sizes that are not powers of two / four)
    for n in (HW, 12 * 16, 1000, 4097):
        full = ops.draw_ray_idx(n, n, seed=3, draw=7, device=DEV).cpu().numpy()
        assert np.array_equal(np.sort(full), np.arange(n))
    # uniformity of the first 2048 of 120,000 over many draws: mean pixel id and a chi-square over 16 bins
    draws = np.concatenate([ops.draw_ray_idx(HW, 2048, seed=5, draw=d, device=DEV).cpu().numpy() for d in range(200)])
    assert abs(draws.mean() / HW - 0.5) < 0.005
    hist = np.histogram(draws, bins=16, range=(0, HW))[0]
    expected = len(draws) / 16
    assert ((hist - expected) ** 2 / expected).sum() < 50           # 15 dof: 50 is far in the tail


def test_raygen_pixel_range_equals_index_tensor():
    from neural_invertible_warp_amd import ops
    H, W, B = 30, 40, 3
    intr = torch.tensor([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]], device=DEV).repeat(B, 1, 1)
    pose = torch.eye(3, 4, device=DEV).repeat(B, 1, 1)
    pose[:, :, 3] = torch.randn(B, 3, device=DEV)
    for mode in (0, 1):
        a = ops.raygen(intr, pose, torch.arange(100, 777, device=DEV), H, W, mode)
        b = ops.raygen(intr, pose, None, H, W, mode, pixel_range=(100, 677))
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    full = ops.raygen(intr, pose, None, H, W, 1)
    assert torch.equal(full[1][:, 100:777], b[1])
    from neural_invertible_warp_amd._lib import NiwError
    with pytest.raises(NiwError, match="leaves the"):
        ops.raygen(intr, pose, None, H, W, 1, pixel_range=(1000, 400))


def test_image_sweep_is_slice_independent():
    """render_by_slices in pixel ranges of any size gives bit-identical images (reference nerf.py:321-332 semantics)"""
    from neural_invertible_warp_amd import configs
    from neural_invertible_warp_amd.model import nerf
    opt = configs.cfg1_nerf_llff_repr(device=DEV)
    opt.H, opt.W = 24, 32
    opt.nerf.sample_stratified = False
    opt.nerf.sample_intvs, opt.nerf.sample_intvs_fine = 16, 16
    g = nerf.Graph(opt)
    intr = torch.tensor([[0.8 * 32, 0, 16], [0, 0.8 * 32, 12], [0, 0, 1]], device=DEV)[None]
    pose = torch.eye(3, 4, device=DEV)[None]
    with torch.no_grad():
        outs = []
        for rays in (768, 100, 257):
            opt.nerf.rand_rays = rays
            outs.append(g.render_by_slices(opt, pose, intr=intr, mode="eval"))
    for k in outs[0]:
        assert outs[0][k].shape[1] == 24 * 32
        assert torch.equal(outs[0][k], outs[1][k]) and torch.equal(outs[0][k], outs[2][k]), k


def test_mse_single_launch_matches_torch_at_full_image():
    from neural_invertible_warp_amd import ops
    B, H, W = 2, 300, 400
    rgb = torch.rand(B, H * W, 3, device=DEV).requires_grad_(True)
    image = torch.rand(B, 3, H, W, device=DEV)
    loss = ops.mse_gather(rgb, image)
    loss.backward()
    r2 = rgb.detach().clone().requires_grad_(True)
    ref = ((r2 - image.view(B, 3, H * W).permute(0, 2, 1)) ** 2).mean()
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-6
    assert (rgb.grad - r2.grad).abs().max() < 1e-9
    assert torch.equal(ops.mse_gather(rgb.detach(), image), ops.mse_gather(rgb.detach(), image))     # no float atomics


def test_ndc_reparametrisation_keeps_the_gradient_to_the_warp():
    """with camera.ndc the rays of a training step still depend on the warp; the photometric gradient must reach
    warp_mlp / warp_latent (the NDC kernel has no backward, so that case runs as torch algebra) and match the kernel's values."""
    from neural_invertible_warp_amd import camera, configs, ops
    from neural_invertible_warp_amd.model import barf_inn_llff
    from neural_invertible_warp_amd.util import edict
    opt = configs.cfg3_barf_inn_llff(device=DEV, global_alignment=None)
    opt.H, opt.W, opt.camera.ndc = 24, 32, True
    opt.nerf.sample_intvs, opt.nerf.rand_rays = 16, 3 * 20
    opt.nerf.depth.range = [0, 1]
    opt.nerf.depth.param = "metric"
    graph = barf_inn_llff.Graph(opt).attach_warp(opt, 3)
    with torch.no_grad():
        # the reference initialisation zeroes the heads AND the latent columns of the first layers, which makes the latent's
        # gradient vanish identically at step 0: perturb every warp parameter so that each route carries signal
        for p in graph.warp_mlp.parameters():
            p.add_(torch.randn_like(p) * 0.02)
    var = edict(idx=torch.arange(3), image=torch.rand(3, 3, 24, 32, device=DEV),
                intr=torch.tensor([[0.8 * 32, 0, 16], [0, 0.8 * 32, 12], [0, 0, 1]], device=DEV).repeat(3, 1, 1))
    var = graph.forward(opt, var, mode="train", iter=50000)
    graph.compute_loss(opt, var, mode="train").render.backward()
    assert graph.warp_latent.weight.grad is not None and graph.warp_latent.weight.grad.abs().max() > 0
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in graph.warp_mlp.parameters())
    # torch route == kernel route on the same rays
    c = torch.randn(3, 50, 3, device=DEV) * 0.1
    r = torch.randn(3, 50, 3, device=DEV) * 0.2 + torch.tensor([0.0, 0.0, 1.0], device=DEV)
    a = camera.convert_NDC(opt, c, r, var.intr)
    b = camera.convert_NDC(opt, c.clone().requires_grad_(True), r, var.intr)
    for x, y in zip(a, b):
        torch.testing.assert_close(x, y.detach(), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("family", ["llff", "dtu"])
def test_hip_graph_replay_equals_eager_training(family):
    """The captured iteration (engine.INNTrainer(hip_graph=True): constants uploaded per step, one graph replay) must train exactly
    like the eager engine: same pixel draws (Feistel, keyed by the iteration), deterministic mid-point samples, 8 steps; parameters
    agree to the noise of the float atomics of the ray-gradient accumulation."""
    from neural_invertible_warp_amd import configs, engine

    def run(hip_graph):
        if family == "llff":
            opt = configs.cfg3_barf_inn_llff(device=DEV)
            B = 5
            var0, init = engine.synthetic_scene(opt, B), None
        else:
            opt = configs.cfg5_barf_inn_dtu(device=DEV)
            B = 3
            var0, init = engine.synthetic_dtu_scene(opt, B)
        opt.nerf.sample_stratified = False
        opt.nerf.rand_rays, opt.nerf.sample_intvs, opt.max_iter = B * 40, 32, 40      # short schedules: c2f bands, windows and lr all move
        opt.inn.real_nvp.max_pe_iter = 20
        tr = engine.INNTrainer(opt, B, warp_perturb=0.02, seed=11, initial_poses_w2c=init, hip_graph=hip_graph)
        losses = []
        for _ in range(8):
            loss = tr.train_iteration(type(var0)(var0))
            losses.append({k: float(v.detach()) for k, v in loss.items()})
        return tr, losses

    eager, l_e = run(False)
    graph, l_g = run(True)
    assert graph._captured is not None and not getattr(graph, "hip_graph_failed", False), "the iteration was not captured"
    for a, b in zip(l_e, l_g):
        for k in a:
            assert abs(a[k] - b[k]) <= 2e-4 * max(abs(a[k]), 1e-3), (k, a[k], b[k])
    for fa, fb in zip(eager._flats(), graph._flats()):
        assert (fa - fb).abs().max() <= 5e-4 * fa.abs().max()
    assert l_g[-1]["render"] < l_g[0]["render"]
    # the c2f progress Parameter is written on demand under replay
    graph.sync_state()
    assert abs(float(graph.graph.nerf.progress.data) - 8 / 40) < 1e-7


@pytest.mark.parametrize("cfg", ["cfg3", "cfg5"])
def test_hip_graph_replay_equals_eager_at_full_batch_size(cfg):
    """Round 2 regression: at the BASELINE batch sizes the captured iteration went NaN on its second replay (a hipMemsetAsync of the
    warp's 31 MB factor workspace, recorded as a memset node, left stale pad columns; the small-shape test above never saw it).
    Full cfg3 / cfg5 shapes, random stratified draws, 6 steps: every loss term of the replayed engine stays finite and tracks the
    eager engine (the two differ by float-atomic noise that the first Adam steps amplify)."""
    import bench
    runs = {}
    for hip_graph in (False, True):
        tr, var0 = bench.build_workloads(cfg, DEV, 0, 1, "weak", 0, hip_graph=hip_graph)[0][0][:2]
        runs[hip_graph] = [{k: float(v.detach()) for k, v in tr.train_iteration(type(var0)(var0)).items()} for _ in range(6)]
        assert all(bool(torch.isfinite(f).all()) for f in tr._flats())
        assert (tr._captured is not None) == hip_graph
        del tr
    for a, b in zip(runs[False], runs[True]):
        for k in a:
            assert abs(a[k] - b[k]) <= 2e-3 * max(abs(a[k]), 1e-4), (k, a[k], b[k])


def test_hip_graph_replays_without_host_synchronisation_use_their_own_step_constants():
    """Round 2 regression: the per-step scalars (pixel-draw number, c2f bands, warp windows, Adam step sizes) reach the device by an
    asynchronous copy from pinned memory.  With ONE staging buffer the host, running ahead, overwrote it before the device had read
    it, and un-synchronised replays trained with the constants of later steps.  Twelve replays issued back to back (no read-back in
    between) must leave the same parameters as twelve synchronised ones."""
    from neural_invertible_warp_amd import configs, engine

    def run(sync):
        opt = configs.cfg3_barf_inn_llff(device=DEV)
        opt.nerf.sample_stratified = False
        opt.nerf.rand_rays, opt.nerf.sample_intvs, opt.max_iter = 5 * 40, 32, 40
        opt.inn.real_nvp.max_pe_iter = 20
        var0 = engine.synthetic_scene(opt, 5)
        tr = engine.INNTrainer(opt, 5, warp_perturb=0.02, seed=11, hip_graph=True)
        for _ in range(12):
            tr.train_iteration(type(var0)(var0))
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return tr

    a, b = run(True), run(False)
    for fa, fb in zip(a._flats(), b._flats()):
        assert (fa - fb).abs().max() <= 5e-4 * fa.abs().max()


def test_hip_graph_capture_after_a_resume_at_a_late_iteration():
    """A resumed run enters train_iteration for the first time with a large iteration number: the trainer must still run its two
    launch-by-launch warm-up iterations before capturing (the capture used to be attempted at once when `it >= 2`), and then track
    the eager engine started at the same iteration."""
    from neural_invertible_warp_amd import configs, engine

    def run(hip_graph):
        opt = configs.cfg3_barf_inn_llff(device=DEV)
        opt.nerf.sample_stratified = False
        opt.nerf.rand_rays, opt.nerf.sample_intvs, opt.max_iter = 5 * 40, 32, 4000
        var0 = engine.synthetic_scene(opt, 5)
        tr = engine.INNTrainer(opt, 5, warp_perturb=0.02, seed=3, hip_graph=hip_graph)
        tr.it = 1234                                    # what checkpoint.restore_checkpoint leaves behind
        for n in tr.nets:
            n.set_progress(tr.it / opt.max_iter)
        losses = [float(tr.train_iteration(type(var0)(var0)).render.detach()) for _ in range(6)]
        return tr, losses

    eager, l_e = run(False)
    graph, l_g = run(True)
    assert graph._captured is not None and graph.it == 1240
    for a, b in zip(l_e, l_g):
        assert abs(a - b) <= 2e-4 * max(abs(a), 1e-3)
    for fa, fb in zip(eager._flats(), graph._flats()):
        assert (fa - fb).abs().max() <= 5e-4 * fa.abs().max()


def test_eval_render_under_hip_graph_uses_the_host_bands_not_the_step_buffer():
    """Round 2 advisor finding: with hip_graph=True the networks kept pointing at the device-resident step constants, so a render
    outside a train iteration read the c2f bands / annealing windows of the LAST train step (all zeros before the first one).  A
    full-image render and a gradient-free warped-pose render of a graph-mode trainer must equal those of an eager trainer in the
    same state: before the first step, and after 5 steps (3 of them replays)."""
    from neural_invertible_warp_amd import configs, engine, evaluation

    def build(hip_graph):
        opt = configs.cfg3_barf_inn_llff(device=DEV)
        opt.H, opt.W = 12, 16
        opt.nerf.sample_stratified = False
        opt.nerf.rand_rays, opt.nerf.sample_intvs, opt.max_iter = 5 * 16, 32, 20        # progress moves through the c2f window [0.1, 0.5]
        opt.inn.real_nvp.max_pe_iter = 10
        var0 = engine.synthetic_scene(opt, 5)
        return opt, var0, engine.INNTrainer(opt, 5, warp_perturb=0.02, seed=5, hip_graph=hip_graph)

    def renders(opt, var0, tr):
        with torch.no_grad():
            pose = torch.eye(3, 4, device=DEV)[None]
            img = tr.graph.render_by_slices(opt, pose, intr=var0.intr[:1], mode="val")
            ray_idx = torch.arange(0, opt.H * opt.W, 7, device=DEV)
            val = tr.graph.render(opt, pose, intr=var0.intr[:1], ray_idx=ray_idx, mode="val")
            # the learnt camera poses (warp of the pixel grid with the annealing window of the CURRENT iteration, then registration)
            learnt, _ = evaluation.LLFFEvaluator(opt, tr.graph, var0.pose).get_all_training_poses(opt)
        return img.rgb.clone(), val.rgb.clone(), learnt.clone()

    (oe, ve, eager), (og, vg, graph) = build(False), build(True)
    for steps in (0, 5):
        for _ in range(steps):
            eager.train_iteration(type(ve)(ve))
            graph.train_iteration(type(vg)(vg))
        if steps:
            assert graph._captured is not None
            assert 0.1 < graph.it / og.max_iter < 0.5, "the test must sit inside the c2f window"
        for a, b in zip(renders(oe, ve, eager), renders(og, vg, graph)):
            assert bool(torch.isfinite(b).all())
            assert (a - b).abs().max() < (1e-6 if steps == 0 else 2e-3), (steps, float((a - b).abs().max()))
        for n in graph.nets:
            assert n.band_dev is None
        assert graph.warp_mlp.window_dev is None and graph.graph.draw_dev is None


def test_hip_graph_replay_refuses_or_refreshes_inputs_it_was_not_captured_with():
    """Round 2 advisor finding: a replay reads the tensors saved at capture time and used to ignore the `var` it was handed.  Same
    data in new storage is copied into the trainer's PRIVATE copies (training continues on the NEW values, the caller's tensors are never
    written); another shape is an error."""
    from neural_invertible_warp_amd import configs, engine
    from neural_invertible_warp_amd._lib import NiwError
    opt = configs.cfg3_barf_inn_llff(device=DEV)
    opt.H, opt.W = 12, 16
    opt.nerf.sample_stratified = False
    opt.nerf.rand_rays, opt.nerf.sample_intvs, opt.max_iter = 5 * 16, 32, 40
    var0 = engine.synthetic_scene(opt, 5)
    var1 = type(var0)(var0)
    var1.image = torch.full_like(var0.image, 0.25)             # a constant image in NEW storage
    assert var1.image.data_ptr() != var0.image.data_ptr()
    # what the fourth step on the new image, and the fifth back on the first one, must give: the eager engine
    ref = engine.INNTrainer(opt, 5, warp_perturb=0.02, seed=5, hip_graph=False)
    for _ in range(3):
        ref.train_iteration(type(var0)(var0))
    loss_ref = float(ref.train_iteration(type(var1)(var1)).render.detach())
    loss_back = float(ref.train_iteration(type(var0)(var0)).render.detach())
    image0 = var0.image.clone()
    tr = engine.INNTrainer(opt, 5, warp_perturb=0.02, seed=5, hip_graph=True)
    for _ in range(3):
        tr.train_iteration(type(var0)(var0))
    assert tr._captured is not None
    loss_new = float(tr.train_iteration(var1).render.detach())
    assert abs(loss_new - loss_ref) <= 2e-3 * loss_ref, (loss_new, loss_ref)
    # round 3 advisor finding: the refresh wrote batch B INTO the caller's batch-A tensor.  The replay owns private copies now: the
    # caller's first batch is untouched, and alternating back trains on its values again
    assert torch.equal(var0.image, image0), "a replay wrote into the caller's tensor"
    loss_a = float(tr.train_iteration(type(var0)(var0)).render.detach())
    assert abs(loss_a - loss_back) <= 2e-3 * loss_back, (loss_a, loss_back)
    assert torch.equal(var0.image, image0)
    var2 = type(var0)(var0)
    var2.image = var0.image[:, :, :6].contiguous()
    with pytest.raises(NiwError, match="captured iteration"):
        tr.train_iteration(var2)


def _trained_flats(hip_graph, steps=6):
    from neural_invertible_warp_amd import configs, engine
    opt = configs.cfg3_barf_inn_llff(device=DEV)
    opt.nerf.rand_rays, opt.nerf.sample_intvs, opt.max_iter = 5 * 40, 32, 40
    opt.inn.real_nvp.max_pe_iter = 20
    var0 = engine.synthetic_scene(opt, 5)
    tr = engine.INNTrainer(opt, 5, warp_perturb=0.02, seed=11, hip_graph=hip_graph)
    losses = [float(tr.train_iteration(type(var0)(var0)).all.detach()) for _ in range(steps)]
    torch.cuda.synchronize()
    return [f.clone() for f in tr._flats()], losses


def test_training_is_bit_reproducible_and_graph_replay_equals_eager_bit_for_bit():
    """No float atomics are left on the train path (round 3: the ray gradients d_center / d_ray are fixed-order per-ray sums; dW and
    the warp's parameter gradients were already fixed-order reductions; pixel and depth draws are keyed by seed and iteration): two
    runs of the same training give IDENTICAL parameters, and so does the captured-graph engine against the eager one -- rounds 1-2
    could only compare them "to the noise of the float atomics"."""
    a, la = _trained_flats(False)
    b, lb = _trained_flats(False)
    assert la == lb
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    c, lc = _trained_flats(True)
    assert la == lc
    for x, y in zip(a, c):
        assert torch.equal(x, y)


# ---------------------------------------------------------------------------------------------
# round 5: compositing + photometric residual + their backward as ONE launch (niw_composite_mse_train) against the three calls it
# replaces in a train iteration -- same device functions, same arithmetic order: every output bit for bit, the loss included
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S,N,B,R,first", [(16, 37, 3, 20, 5), (64, 45, 5, 9, 0), (128, 254, 18, 113, 508), (192, 101, 4, 60, 17), (200, 33, 3, 11, 0),
                                          (256, 18, 2, 9, 0)])
def test_one_launch_compositing_loss_and_backward_equal_the_three_calls(S, N, B, R, first):
    from neural_invertible_warp_amd import _lib, ops
    P = ops._p
    rng = np.random.default_rng(S + N)
    H, W = 12, 16
    hw = H * W
    g = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    ray, rgb_s = g(rng.standard_normal((N, 3))), g(rng.uniform(0, 1, (N, S, 3)))
    sig = g(rng.uniform(0, 3, (N, S)) * (rng.uniform(0, 1, (N, S)) > 0.3))
    dep = g(np.sort(rng.uniform(0.5, 6, (N, S)), axis=1))
    image = g(rng.uniform(0, 1, (B, 3, H, W)))
    ray_idx = torch.from_numpy(rng.permutation(hw)[:R].astype(np.int64)).to(DEV)
    n_norm, scale = float(3 * B * R), 10.0 ** 0.5
    st = ops._stream()
    new = lambda *shape: torch.full(shape, float("nan"), device=DEV)

    a = dict(rgb=new(N, 3), depth=new(N), opa=new(N), prob=new(N, S), d_rgb=new(N, 3), d_rgb_s=new(N, S, 3), d_sig=new(N, S), d_ray=new(N, 3), loss=new(1))
    _lib.call("niw_composite_fwd", P(ray), P(rgb_s), P(sig), P(dep), N, S, 0, 0.0, P(a["rgb"]), P(a["depth"]), P(a["opa"]), P(a["prob"]), st)
    _lib.call("niw_mse_fwd_bwd", P(a["rgb"]), P(image), P(ray_idx), B, R, hw, first, N, n_norm, scale, P(a["loss"]), P(a["d_rgb"]), st)
    _lib.call("niw_composite_bwd", P(ray), P(rgb_s), P(sig), P(dep), N, S, 0, 0.0, P(a["d_rgb"]), None, None, None, P(a["d_rgb_s"]), P(a["d_sig"]), P(a["d_ray"]), st)

    b = dict(rgb=new(N, 3), depth=new(N), opa=new(N), prob=new(N, S), d_rgb=new(N, 3), d_rgb_s=new(N, S, 3), d_sig=new(N, S), d_ray=new(N, 3), loss=new(1))
    resid = new(N, 3)
    _lib.call("niw_composite_mse_train", P(ray), P(rgb_s), P(sig), P(dep), N, S, P(image), P(ray_idx), B, R, hw, first, n_norm, scale,
              P(b["rgb"]), P(b["depth"]), P(b["opa"]), P(b["prob"]), P(resid), P(b["d_rgb"]), P(b["d_rgb_s"]), P(b["d_sig"]), P(b["d_ray"]), st)
    _lib.call("niw_mse_from_residuals", P(resid), N, n_norm, P(b["loss"]), st)
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), (k, float((a[k] - b[k]).abs().max()))
    assert not torch.isnan(b["d_rgb_s"]).any() and float(b["loss"]) > 0
    # the residuals are rgb - pixel of the ray's view and pixel
    br = first + torch.arange(N, device=DEV)
    pix = image.view(B, 3, hw)[br // R][:, :, ray_idx][torch.arange(N, device=DEV), :, br % R]
    assert torch.equal(resid, b["rgb"] - pix)


def test_one_launch_form_refuses_shapes_outside_the_span_kernels():
    from neural_invertible_warp_amd import _lib, ops
    from neural_invertible_warp_amd._lib import NiwError
    P = ops._p
    N, B, R, hw = 8, 2, 4, 16
    z = lambda *shape: torch.zeros(shape, device=DEV)
    for S in (6, 260):
        out = [z(N, 3), z(N), z(N), z(N, S), z(N, 3), z(N, 3), z(N, S, 3), z(N, S), z(N, 3)]
        with pytest.raises(NiwError, match="one-launch form"):
            _lib.call("niw_composite_mse_train", P(z(N, 3)), P(z(N, S, 3)), P(z(N, S)), P(z(N, S)), N, S, P(z(B, 3, hw)), None, B, R, hw, 0, 24.0, 1.0,
                      *[P(o) for o in out], ops._stream())


@pytest.mark.parametrize("case", ["cfg3", "cfg2", "dtu", "cfg3_rank1of3", "cfg2_ndc_noise", "vanilla", "cfg3_mirror"])
def test_train_step_never_reads_memory_it_has_not_written(case):
    """round 6: the one-call iteration (and the mirror) with EVERY uninitialised buffer hostile -- torch.empty filled with NaN from the
    first allocation on (torch.utils.deterministic.fill_uninitialized_memory), and the persistent workspace re-filled with NaN, then with
    all-ones bits (NaN as float, -1 as integer), before every iteration -- must train bit for bit like the plain run: nothing reads a
    workspace piece, a pad column, a partial tile or a loss slot before this iteration has written it.  (Fresh device memory comes zeroed
    and recycled memory holds last iteration's benign values, so an uninitialised read would otherwise stay invisible until several
    processes share a device.)"""
    from tests.test_gpu_fused_step import _trainer, _vanilla_trainer
    make = {"cfg3": lambda: _trainer("cfg3_barf_inn_llff", True), "cfg2": lambda: _trainer("cfg2_nerf_inn_llff_hier", True),
            "dtu": lambda: _trainer("dtu", True), "cfg3_rank1of3": lambda: _trainer("cfg3_barf_inn_llff", True, rank=1, world=3, stratified=False),
            "cfg2_ndc_noise": lambda: _trainer("cfg2_nerf_inn_llff_hier", True, ndc=True, noise=0.5), "vanilla": lambda: _vanilla_trainer(True),
            "cfg3_mirror": lambda: _trainer("cfg3_barf_inn_llff", False)}[case]

    def run(poison):
        tr, var0 = make()
        losses = []
        for i in range(3):
            ws = getattr(tr.fused, "ws", None) if tr.fused is not None else None
            if poison and ws is not None:
                ws.fill_(float("nan")) if i % 2 == 0 else ws.view(torch.int32).fill_(-1)
            loss = tr.train_iteration(type(var0)(var0))
            losses.append({k: float(v.detach()) for k, v in loss.items()})
        torch.cuda.synchronize()
        return tr, losses

    a, la = run(False)
    det, fill = torch.are_deterministic_algorithms_enabled(), torch.utils.deterministic.fill_uninitialized_memory
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
    try:
        b, lb = run(True)
    finally:
        torch.use_deterministic_algorithms(det)
        torch.utils.deterministic.fill_uninitialized_memory = fill
    assert la == lb, (la, lb)
    assert torch.equal(a.bucket.flat, b.bucket.flat)
    for x, y in zip(a._flats() + a.m + a.v, b._flats() + b.m + b.v):
        assert torch.equal(x, y)


@pytest.mark.parametrize("case", ["cfg3", "cfg2", "vanilla", "cfg3_mirror"])
def test_training_on_a_non_default_stream_equals_the_default_stream(case):
    """every launch of an iteration -- the library's, the wrappers' and torch's own -- must follow the CURRENT stream: the same iterations
    issued inside `torch.cuda.stream(s)`, with the default stream kept busy by an unrelated long kernel train, train bit for bit alike
    (a launch that slipped onto the default stream would run late or early against its neighbours)"""
    from tests.test_gpu_fused_step import _trainer, _vanilla_trainer
    make = {"cfg3": lambda: _trainer("cfg3_barf_inn_llff", True), "cfg2": lambda: _trainer("cfg2_nerf_inn_llff_hier", True),
            "vanilla": lambda: _vanilla_trainer(True), "cfg3_mirror": lambda: _trainer("cfg3_barf_inn_llff", False)}[case]

    def run(stream):
        tr, var0 = make()
        busy = torch.rand(1 << 24, device=DEV)
        losses = []
        for i in range(4):
            if stream is None:
                loss = tr.train_iteration(type(var0)(var0))
            else:
                for _ in range(4):
                    busy.mul_(1.0000001)                       # the default stream is never idle
                with torch.cuda.stream(stream):
                    loss = tr.train_iteration(type(var0)(var0))
            losses.append(loss)
        torch.cuda.synchronize()
        return tr, [{k: float(v.detach()) for k, v in l.items()} for l in losses]

    a, la = run(None)
    s = torch.cuda.Stream()
    b, lb = run(s)
    assert la == lb, (la, lb)
    assert torch.equal(a.bucket.flat, b.bucket.flat)
    for x, y in zip(a._flats() + a.m + a.v, b._flats() + b.m + b.v):
        assert torch.equal(x, y)
