"""PSNR-trajectory parity run (TEST INFRASTRUCTURE ONLY, like everything under oracle/): N identical optimisation steps of
the INN train iteration on the HIP path (engine.INNTrainer: fused kernels + niw_adam_step) and on the CPU oracle (autograd +
torch.optim.Adam), from identical weights, images, pixel draws and stratified draws.  Used by tests/test_gpu_psnr_parity.py and
by the `psnr_parity` field of bench.py, where the oracle is the checker, never the thing measured."""
import math

import numpy as np
import torch

from . import niw_oracle as O


def psnr(mse):
    return -10.0 * math.log10(mse)


def psnr_trajectories(dev, steps=25, B=3, H=12, W=16, R=16, S=32, seed=7, precision="fp32"):
    """-> (psnr_hip[steps], psnr_oracle[steps]) of the photometric loss of a barf_inn_llff run (c2f encoding, annealed warp
    embedding, both Adam groups under ExponentialLR; no alignment term: its Kabsch solver is parity-unpinned)."""
    from neural_invertible_warp_amd import configs, engine
    from neural_invertible_warp_amd.util import edict
    opt = configs.cfg3_barf_inn_llff(device=dev, global_alignment=None)
    opt.H, opt.W = H, W
    opt.arch.precision = precision                 # arithmetic of the HIP engine's field MLP; the oracle is always the reference's fp32
    opt.nerf.sample_intvs, opt.nerf.rand_rays = S, R * B
    tr = engine.INNTrainer(opt, B, warp_perturb=0.0, ray_sampler="randperm")     # the harness injects the pixel draw through torch.randperm
    pc, wp, lat = O.make_nerf_params(71), O.make_warp_params(72, 0.02), O.make_latent(73, B)
    with torch.no_grad():
        for mod, prm in ((tr.graph.nerf, pc), (tr.graph.warp_mlp, wp)):
            sd = mod.state_dict()
            for k, v in prm.items():
                sd[k].copy_(v)
        tr.graph.warp_latent.weight.copy_(lat)
    req = lambda d: {k: v.clone().requires_grad_(True) for k, v in d.items()}
    pc, wp, lat = req(pc), req(wp), lat.clone().requires_grad_(True)
    o = opt.optim
    opt_nerf = torch.optim.Adam(list(pc.values()), lr=o.lr)
    opt_pose = torch.optim.Adam(list(wp.values()) + [lat], lr=o.lr_pose)
    g_nerf = (o.lr_end / o.lr) ** (1.0 / opt.max_iter)
    g_pose = (o.lr_pose_end / o.lr_pose) ** (1.0 / opt.max_iter)

    rng = np.random.default_rng(seed)
    image = torch.from_numpy(rng.uniform(0, 1, (B, 3, H, W)).astype(np.float32))
    intr = torch.tensor([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]], dtype=torch.float32).repeat(B, 1, 1)
    var0 = edict(idx=torch.arange(B), image=image.to(dev), intr=intr.to(dev))
    rand, perm = torch.rand, torch.randperm
    psnr_gpu, psnr_cpu = [], []
    try:
        for it in range(steps):                    # `it` = the reference's self.it during the step (0-based)
            u = torch.from_numpy(rng.uniform(0, 1, (B, R, S, 1)).astype(np.float32))
            ray_idx = torch.from_numpy(rng.permutation(H * W)[:R].astype(np.int64))
            torch.rand, torch.randperm = (lambda *a, **k: u.to(dev)), (lambda *a, **k: ray_idx.to(dev))
            loss = tr.train_iteration(edict(var0))
            torch.rand, torch.randperm = rand, perm
            psnr_gpu.append(psnr(float(loss.render.detach())))
            # the same step on the oracle: progress was set to it / max_iter after the previous step (barf_inn_llff.py:117)
            prog = it / opt.max_iter
            w3, wv = O.c2f_weights(prog, opt.barf_c2f, 10), O.c2f_weights(prog, opt.barf_c2f, 4)
            for prm in list(pc.values()) + list(wp.values()) + [lat]:
                prm.grad = None
            out = O.inn_train_step(pc, wp, lat, image, intr, ray_idx, u, H, W, S, (1, 0), "inverse", it / opt.inn.real_nvp.max_pe_iter,
                                   w3d=w3, wview=wv)
            out["loss"].backward()
            for grp, lr0, gam in ((opt_nerf, o.lr, g_nerf), (opt_pose, o.lr_pose, g_pose)):
                grp.param_groups[0]["lr"] = lr0 * gam ** it
                grp.step()
            psnr_cpu.append(psnr(float(out["loss_render"].detach())))
    finally:
        torch.rand, torch.randperm = rand, perm
    return psnr_gpu, psnr_cpu


def long_trajectories(dev, steps=1000, views=8, size=(48, 64), R=256, S=64, ga=4, seed=0, draw_seed=0, oracle=True, log_every=50, precision="fp32", Sf=0):
    """Long-horizon parity on the demo scene (tools/teacher_student_demo.py: analytic density blobs rendered from perturbed poses,
    training starts from identity poses): `steps` chained iterations of barf_inn_llff WITH the alignment term on
      * the HIP engine (engine.INNTrainer: fused kernels, niw_adam_step), and
      * the oracle (autograd + torch.optim.Adam) run on `dev` through torch's own kernels,
    from identical weights, with identical pixel draws and stratified draws (numpy stream `draw_seed`).  fp32 trajectories of a
    non-convex optimisation separate chaotically, so the yardstick is the spread between HIP runs that differ ONLY in `draw_seed`.
    -> dict(it=[...], hip=dict(psnr=[...], rel_rot=[...]), oracle=dict(...))   (rel_rot: pairwise relative rotation error in degrees of
    the poses registered from the step's warped points, against the scene's ground truth)
    Sf > 0 (round 5): the cfg2 shape instead -- hierarchical resampling and a FINE network (S coarse + Sf resampled positions, both losses,
    one Adam over both networks: reference model/nerf.py:34-38, 293-319), no alignment term; `psnr` is then the FINE image's, psnr_coarse
    the coarse one's."""
    from neural_invertible_warp_amd import camera, configs, engine
    from neural_invertible_warp_amd.util import edict
    from tools.teacher_student_demo import render_teacher
    H, W = size
    B = views
    if Sf:
        opt, ga = configs.cfg2_nerf_inn_llff_hier(device=dev), None
        opt.nerf.sample_intvs_fine = Sf
    else:
        opt = configs.cfg3_barf_inn_llff(device=dev, global_alignment=ga)
    opt.H, opt.W, opt.data.image_size = H, W, [H, W]
    opt.max_iter = steps
    opt.nerf.sample_intvs, opt.nerf.rand_rays = S, R * B
    opt.inn.real_nvp.max_pe_iter = steps // 2
    opt.arch.precision = precision                   # the HIP engine's field-MLP arithmetic (include/niw.h niw_precision); the oracle stays fp32
    gen = torch.Generator().manual_seed(seed)
    pose_GT = camera.lie.se3_to_SE3(torch.randn(B, 6, generator=gen) * torch.tensor([0.06, 0.06, 0.03, 0.15, 0.15, 0.05])).to(dev)
    intr = torch.tensor([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]], dtype=torch.float32).repeat(B, 1, 1).to(dev)
    image = render_teacher(opt, pose_GT, intr)
    tr = engine.INNTrainer(opt, B, warp_perturb=0.0, ray_sampler="randperm", seed=seed)
    # identical initial weights on both sides: the oracle starts from a copy of the engine's (reference initialisation)
    clone = lambda mod: {k: v.detach().clone().requires_grad_(True) for k, v in mod.state_dict().items() if k != "progress"}
    pc, wp = clone(tr.graph.nerf), clone(tr.graph.warp_mlp)
    pf = clone(tr.graph.nerf_fine) if Sf else None
    lat = tr.graph.warp_latent.weight.detach().clone().requires_grad_(True)
    o = opt.optim
    opt_nerf = torch.optim.Adam(list(pc.values()) + (list(pf.values()) if Sf else []), lr=o.lr)
    opt_pose = torch.optim.Adam(list(wp.values()) + [lat], lr=o.lr_pose)
    g_nerf = (o.lr_end / o.lr) ** (1.0 / opt.max_iter)
    g_pose = (o.lr_pose_end / o.lr_pose) ** (1.0 / opt.max_iter)
    Rg = pose_GT[:, :, :3]
    rel_gt = Rg[:, None] @ Rg[None].transpose(-1, -2)

    def rel_rot(grid_cam, center_cam, grid_3D, center_3D):
        with torch.no_grad():
            _, poses = O.global_alignment_loss(grid_cam.float(), center_cam.float(), grid_3D.float(), center_3D.float())
            Rp = poses[:, :, :3]
            rel = camera.rotation_distance(Rp[:, None] @ Rp[None].transpose(-1, -2), rel_gt)
            return float(rel.sum() / (B * B - B)) * 57.29577951308232

    rng = np.random.default_rng(1000 + draw_seed)
    var0 = edict(idx=torch.arange(B), image=image, intr=intr)
    out = dict(it=[], hip=dict(psnr=[], rel_rot=[], psnr_coarse=[]), oracle=dict(psnr=[], rel_rot=[], psnr_coarse=[]))
    rand, perm = torch.rand, torch.randperm
    try:
        for it in range(steps):
            u = torch.from_numpy(rng.uniform(0, 1, (B, R, S, 1)).astype(np.float32)).to(dev)
            ray_idx = torch.from_numpy(rng.permutation(H * W)[:R].astype(np.int64)).to(dev)
            log = (it % log_every == 0) or it == steps - 1
            torch.rand, torch.randperm = (lambda *a, **k: u), (lambda *a, **k: ray_idx)
            var = edict(var0)
            loss = tr.train_iteration(var)
            torch.rand, torch.randperm = rand, perm
            if log:
                out["it"].append(it)
                out["hip"]["psnr"].append(psnr(float((loss.render_fine if Sf else loss.render).detach())))
                out["hip"]["psnr_coarse"].append(psnr(float(loss.render.detach())))
                out["hip"]["rel_rot"].append(rel_rot(var.grid_cam, var.center_cam, var.grid_3D, var.center))
            if not oracle:
                continue
            prog = it / opt.max_iter
            w3, wv = O.c2f_weights(prog, opt.barf_c2f, 10), O.c2f_weights(prog, opt.barf_c2f, 4)
            for prm in list(pc.values()) + list(wp.values()) + [lat] + (list(pf.values()) if Sf else []):
                prm.grad = None
            ref = O.inn_train_step(pc, wp, lat, image, intr, ray_idx, u, H, W, S, (1, 0), "inverse", min(max(it / opt.inn.real_nvp.max_pe_iter, 0), 1),
                                   nerf_fine_p=pf, Sf=Sf, ga_weight=ga, w3d=w3, wview=wv)
            ref["loss"].backward()
            for grp, lr0, gam in ((opt_nerf, o.lr, g_nerf), (opt_pose, o.lr_pose, g_pose)):
                grp.param_groups[0]["lr"] = lr0 * gam ** it
                grp.step()
            if log:
                cc, gc = O.unwarped_center_and_grid(H, W, intr, ray_idx)
                out["oracle"]["psnr"].append(psnr(float(ref["loss_render_fine" if Sf else "loss_render"].detach())))
                out["oracle"]["psnr_coarse"].append(psnr(float(ref["loss_render"].detach())))
                out["oracle"]["rel_rot"].append(rel_rot(gc, cc, ref["grid_3D"].detach(), ref["center"].detach()))
    finally:
        torch.rand, torch.randperm = rand, perm
    return out


def trajectory_summary(dev, steps=120, views=8, size=(48, 64), R=256, S=64, ga=4, spread_runs=2, log_every=10, precision="fp32"):
    """The "+ PSNR parity" half of the metric as bench.py reports it (round 6; the 12 x 16 / 10-step sample of rounds 1-5 never left the
    untrained regime): the teacher-student scene of `long_trajectories` -- 8 views of 48 x 64, 256 rays x 64 samples, alignment term ON,
    `steps` >= 120 chained iterations -- on the HIP engine and on the oracle from identical weights with identical pixel and stratified
    draws, plus `spread_runs` HIP runs that differ ONLY in their draws (the yardstick for "the same trajectory" of a chaotic fp32
    optimisation).  -> dict, or None when the HIP run did not train (PSNR rise < 3 dB): a parity figure of an untrained network is not
    evidence."""
    import time
    t0 = time.perf_counter()
    kw = dict(steps=steps, views=views, size=size, R=R, S=S, ga=ga, log_every=log_every, precision=precision)
    pair = long_trajectories(dev, draw_seed=0, oracle=True, **kw)
    others = [long_trajectories(dev, draw_seed=s, oracle=False, **kw)["hip"] for s in range(1, 1 + spread_runs)]
    hip, ora, n = pair["hip"], pair["oracle"], len(pair["it"])
    rise = hip["psnr"][-1] - hip["psnr"][0]
    if not (rise >= 3.0 and math.isfinite(hip["psnr"][-1]) and math.isfinite(ora["psnr"][-1])):
        return None
    half = n // 2
    gap = [abs(a - b) for a, b in zip(hip["psnr"], ora["psnr"])]
    spread = [max(x) - min(x) for x in zip(hip["psnr"], *[o["psnr"] for o in others])]
    rot_gap = [abs(a - b) for a, b in zip(hip["rel_rot"], ora["rel_rot"])]
    r = lambda x, k=4: round(float(x), k)
    return dict(steps=steps, logged_points=n, initial_psnr_hip=r(hip["psnr"][0]), final_psnr_hip=r(hip["psnr"][-1]), final_psnr_oracle=r(ora["psnr"][-1]),
                psnr_rise_db=r(rise), max_abs_diff_db=r(max(gap)), second_half_mean_abs_diff_db=r(sum(gap[half:]) / (n - half)),
                second_half_mean_draw_spread_db=r(sum(spread[half:]) / (n - half)), draw_spread_runs=spread_runs,
                rel_rot_deg_hip=r(hip["rel_rot"][-1]), rel_rot_deg_oracle=r(ora["rel_rot"][-1]), initial_rel_rot_deg=r(hip["rel_rot"][0]),
                max_rel_rot_gap_deg=r(max(rot_gap)), seconds=r(time.perf_counter() - t0, 1),
                sample=f"barf_inn_llff teacher-student scene (analytic density blobs rendered from perturbed poses, training starts at identity poses): "
                       f"{views} views of {size[0]}x{size[1]}, {R} rays x {S} samples, c2f encoding, alignment term 10^{ga}, both Adam groups under "
                       f"ExponentialLR, {steps} chained iterations; HIP engine vs the oracle (autograd + torch.optim.Adam through torch's kernels on "
                       f"this GPU) from identical weights and identical pixel / stratified draws; draw spread = range over the HIP run and "
                       f"{spread_runs} HIP runs that differ only in their draws; PSNR of the step's photometric loss at every {log_every}th iteration")
