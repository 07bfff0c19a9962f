"""PSNR parity (the second half of BASELINE.json's metric): N identical optimisation steps of the
INN train iteration on the HIP path (engine.INNTrainer: fused kernels + niw_adam_step) and on the CPU
oracle (autograd + torch.optim.Adam), from identical weights, images, pixel draws and stratified
draws.  The photometric PSNR trajectories must agree to 0.005 dB over the first five steps and to
0.15 dB over all 25 (the two fp32 trajectories separate chaotically later), and the PSNR must rise.
Per-channel annealing (`reference_exact=False`) is not needed here: both sides see the same batch, so
the reference's index-dependent embedder quirk is reproduced identically."""
import math

import numpy as np
import pytest
import torch

from oracle import niw_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def psnr(mse):
    return -10.0 * math.log10(mse)


def test_psnr_trajectory_matches_cpu_oracle():
    from neural_invertible_warp_amd import configs, engine
    from neural_invertible_warp_amd.util import edict
    B, H, W, R, S, steps = 3, 12, 16, 16, 32, 25
    opt = configs.cfg3_barf_inn_llff(device=DEV, global_alignment=None)
    opt.H, opt.W = H, W
    opt.nerf.sample_intvs, opt.nerf.rand_rays = S, R * B
    tr = engine.INNTrainer(opt, B, warp_perturb=0.0)
    # identical initial weights on both sides
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

    rng = np.random.default_rng(7)
    image = torch.from_numpy(rng.uniform(0, 1, (B, 3, H, W)).astype(np.float32))
    intr = torch.tensor([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]], dtype=torch.float32).repeat(B, 1, 1)
    var0 = edict(idx=torch.arange(B), image=image.to(DEV), intr=intr.to(DEV))
    rand, perm = torch.rand, torch.randperm
    psnr_gpu, psnr_cpu = [], []
    try:
        for it in range(1, steps + 1):
            u = torch.from_numpy(rng.uniform(0, 1, (B, R, S, 1)).astype(np.float32))
            ray_idx = torch.from_numpy(rng.permutation(H * W)[:R].astype(np.int64))
            torch.rand, torch.randperm = (lambda *a, **k: u.to(DEV)), (lambda *a, **k: ray_idx.to(DEV))
            loss = tr.train_iteration(edict(var0))
            torch.rand, torch.randperm = rand, perm
            psnr_gpu.append(psnr(float(loss.render.detach())))
            # the same step on the oracle: progress of step `it` is (it-1)/max_iter (set after the previous step)
            prog = (it - 1) / opt.max_iter
            w3, wv = O.c2f_weights(prog, opt.barf_c2f, 10), O.c2f_weights(prog, opt.barf_c2f, 4)
            for prm in list(pc.values()) + list(wp.values()) + [lat]:
                prm.grad = None
            out = O.inn_train_step(pc, wp, lat, image, intr, ray_idx, u, H, W, S, (1, 0), "inverse", it / opt.inn.real_nvp.max_pe_iter,
                                   w3d=w3, wview=wv)
            out["loss"].backward()
            for grp, lr0, gam in ((opt_nerf, o.lr, g_nerf), (opt_pose, o.lr_pose, g_pose)):
                grp.param_groups[0]["lr"] = lr0 * gam ** (it - 1)
                grp.step()
            psnr_cpu.append(psnr(float(out["loss_render"].detach())))
    finally:
        torch.rand, torch.randperm = rand, perm
    diff = max(abs(a - b) for a, b in zip(psnr_gpu, psnr_cpu))
    print("PSNR gpu", [round(x, 3) for x in psnr_gpu[::6]], "cpu", [round(x, 3) for x in psnr_cpu[::6]], "max |diff| dB", diff)
    early = max(abs(a - b) for a, b in zip(psnr_gpu[:5], psnr_cpu[:5]))
    # identical to ~1e-3 dB while roundoff is still roundoff; afterwards the two fp32 trajectories separate
    # chaotically (48-ray batches, ReLU flips, Adam's 1/sqrt(v) normalisation): measured 0.08 dB after 25 steps
    assert early < 0.005, f"first steps differ by {early:.4f} dB"
    assert diff < 0.15, f"PSNR trajectories diverge by {diff:.3f} dB"
    assert sum(psnr_gpu[-5:]) / 5 > sum(psnr_gpu[:3]) / 3            # and the model actually learns
