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


def psnr_trajectories(dev, steps=25, B=3, H=12, W=16, R=16, S=32, seed=7):
    """-> (psnr_hip[steps], psnr_oracle[steps]) of the photometric loss of a barf_inn_llff run (c2f encoding, annealed warp
    embedding, both Adam groups under ExponentialLR; no alignment term: its Kabsch solver is parity-unpinned)."""
    from neural_invertible_warp_amd import configs, engine
    from neural_invertible_warp_amd.util import edict
    opt = configs.cfg3_barf_inn_llff(device=dev, global_alignment=None)
    opt.H, opt.W = H, W
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
