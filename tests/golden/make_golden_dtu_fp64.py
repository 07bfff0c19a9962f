#!/usr/bin/env python3
"""float64 companions of the two DTU INN-step fixtures (build container only; imports the real reference like make_golden.py):

    python tests/golden/make_golden_dtu_fp64.py        # writes tests/golden/inn_step_cfg5_fp64.npz

Why.  The DTU pose network warps WORLD points (3-4 units from the origin); its 2^5 pi band turns fp32 roundoff into percent-level
differences between any two fp32 evaluations of its gradients -- the reference's own included -- so a fixture of the reference's fp32
gradients can only bound an implementation to the noise of that one evaluation (the 15 % / 60 % tolerances of rounds 1-2).  Here the
reference's step (INNPoseParams.get_warped_rays_in_world, model/pose_models/inn.py:63-93; nerf_inn_dtu.Graph.forward / render_local /
compute_loss, model/nerf_inn_dtu.py:371-456; c2f encoding of model/barf_inn_llff.py:421-442, see make_golden_dtu.py for why that
class stands in) is run on the stored inputs of `inn_step_cfg5.npz` (no mask) and `inn_step_cfg5_c2f.npz` (--barf_c2f=[0.1,0.5])
twice: in float32 -- which must reproduce the stored fixture -- and in float64 (modules .double(), inputs cast; the fp32 band tables
2^k * fp32(pi) are kept, so it is the SAME function evaluated with 29 more bits).  Stored per parameter tensor:

    <tag>.grad64.<name>.{norm,amax,sample,stride}   the float64 gradient (L2 norm + strided sample, stride 11, values rounded to float32 for storage)
    <tag>.cond.<name>                          max |g32_reference - g64| / max |g64|: what ONE fp32 evaluation of this gradient is
                                               worth at these inputs -- the conditioning bound the tests hold HIP and oracle to
    <tag>.rgb64 / loss64                       forward values in float64
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import niw_oracle as O  # noqa: E402

STRIDE = 11


def run_reference(fx, c2f, dtype, ref_barf, ref_dtu, INNPoseParams, edict):
    H, W, S, Rr, it = (int(fx[k]) for k in ("H", "W", "S", "R", "it"))
    B = 3
    over = {"data.image_size": [H, W]}
    if c2f:
        over["barf_c2f"] = [0.1, 0.5]
    o5 = G.load_opt("barf_inn_dtu", "barf_inn_dtu", **over)
    o5.nerf.sample_intvs, o5.nerf.rand_rays = S, Rr * B
    o5.loss_weight.global_alignment = None
    cast = lambda a: torch.from_numpy(np.asarray(a)).to(dtype)
    pose_init = cast(fx["pose_init"])

    class _DtuGraph(ref_dtu.Graph):
        def __init__(self, opt, pose_net):
            super().__init__(opt)
            self.pose_net = pose_net
            if c2f:
                self.nerf = ref_barf.NeRF(opt)

        def get_pose(self, opt, var, mode=None, iter=None):      # barf_inn_dtu.py:535-545
            return self.pose_net.get_warped_rays_in_world(var, mode, iter)

    pn = INNPoseParams(o5, num_poses=B, initial_poses_w2c=pose_init, device="cpu")
    G.set_params(pn.pose_embedding, O.make_warp_params(seed=int(fx["seed_warp"]), perturb=float(fx["warp_perturb"])))
    with torch.no_grad():
        pn.pose_latent.weight.copy_(O.make_latent(int(fx["seed_latent"]), B))
    g = _DtuGraph(o5, pn)
    G.set_params(g.nerf, O.make_nerf_params(seed=int(fx["seed_coarse"])))
    if c2f:
        g.nerf.progress.data.fill_(float(fx["progress"]))
    g.to(dtype)
    pn.to(dtype)
    for name, buf in list(pn.named_buffers()) + [(k, v) for k, v in vars(pn).items() if isinstance(v, torch.Tensor)]:
        if buf.is_floating_point() and buf.dtype != dtype:
            setattr(pn, name.split(".")[-1], buf.to(dtype))
    u, ray_idx = cast(fx["u"]), torch.from_numpy(fx["ray_idx"])
    var = edict(idx=torch.arange(B), image=cast(fx["image"]), intr=cast(fx["intr"]), pose=pose_init, depth_range=cast(fx["depth_range"]))
    rand, perm = torch.rand, torch.randperm
    torch.rand, torch.randperm = (lambda *a, **k: u.clone()), (lambda *a, **k: ray_idx.clone())
    # The un-warped ray points carry no gradient (they are detached before the warp, inn.py:67) and the reference builds their pixel
    # grid in float32 (camera.py:371-372), which its float64 matmul rejects: they are formed in float32, exactly as in the fp32 step,
    # and cast -- the float64 run then differentiates the SAME function of the parameters at the SAME input points.
    import camera as ref_camera
    unwarped = ref_camera.get_unwarped_center_and_ray

    def unwarped_in_fp32(opt, intr=None, ray_idx=None, pose_init=None):
        c, gr = unwarped(opt, intr=intr.float(), ray_idx=ray_idx, pose_init=None if pose_init is None else pose_init.float())
        return c.to(dtype), gr.to(dtype)

    ref_camera.get_unwarped_center_and_ray = unwarped_in_fp32
    try:
        var = g.forward(o5, var, mode="train", iter=it)
    finally:
        torch.rand, torch.randperm = rand, perm
        ref_camera.get_unwarped_center_and_ray = unwarped
    assert var.rgb.dtype == dtype, (var.rgb.dtype, dtype)
    loss = g.compute_loss(o5, var, mode="train")
    loss.render.backward()
    grads = {}
    for k, prm in list(g.nerf.named_parameters()) + [("pose_embedding." + n, p) for n, p in pn.pose_embedding.named_parameters()] + \
            [("pose_latent.weight", pn.pose_latent.weight)]:
        if prm.grad is not None:
            grads[k] = prm.grad.detach().double().clone()
    return dict(rgb=var.rgb.detach().double(), loss=float(loss.render.detach()), grads=grads)


def main():
    G.install_stubs()
    sys.path.insert(0, G.REF)
    import roma
    roma.rigid_points_registration = lambda x, y, *a, **k: O.rigid_registration(x, y)     # only feeds the detached pose_global
    import model.barf_inn_llff as ref_barf
    import model.nerf_inn_dtu as ref_dtu
    from easydict import EasyDict as edict
    from model.pose_models.inn import INNPoseParams

    out = {}
    for tag, c2f in (("cfg5", False), ("cfg5_c2f", True)):
        fx = dict(np.load(os.path.join(HERE, f"inn_step_{tag}.npz")))
        r32 = run_reference(fx, c2f, torch.float32, ref_barf, ref_dtu, INNPoseParams, edict)
        r64 = run_reference(fx, c2f, torch.float64, ref_barf, ref_dtu, INNPoseParams, edict)
        # the float32 run IS the stored fixture (same inputs, same code): a drift here means the two fixtures no longer describe one step
        assert abs(r32["loss"] - float(fx["loss_render"])) < 1e-7, (r32["loss"], float(fx["loss_render"]))
        assert np.abs(r32["rgb"].float().numpy() - fx["rgb"]).max() < 1e-6
        for k, g32 in r32["grads"].items():
            stored = fx[f"grad.{k}.sample"]
            assert np.abs(g32.reshape(-1)[::int(fx[f"grad.{k}.stride"])].float().numpy() - stored).max() <= 1e-5 * max(np.abs(stored).max(), 1e-12) + 1e-9, k
        out[f"{tag}.loss64"] = np.array(r64["loss"])
        out[f"{tag}.rgb64"] = r64["rgb"].float().numpy()
        print(f"{tag}: loss fp32 {r32['loss']:.9f} fp64 {r64['loss']:.9f}; max |rgb32 - rgb64| {float((r32['rgb'] - r64['rgb']).abs().max()):.2e}")
        for k, g64 in r64["grads"].items():
            f = g64.reshape(-1)
            out[f"{tag}.grad64.{k}.norm"] = np.array(float(f.norm()))
            out[f"{tag}.grad64.{k}.sample"] = f[::STRIDE].float().numpy()      # (the float64 value rounded once: 6e-8 relative)
            out[f"{tag}.grad64.{k}.stride"] = np.array(STRIDE)
            out[f"{tag}.grad64.{k}.amax"] = np.array(float(f.abs().max()))
            cond = float((r32["grads"][k] - g64).abs().max() / g64.abs().max().clamp_min(1e-300))
            out[f"{tag}.cond.{k}"] = np.array(cond)
            if cond > 1e-3:
                print(f"   {k:40s} reference fp32 vs fp64: {cond:.2e} of max")
    path = os.path.join(HERE, "inn_step_cfg5_fp64.npz")
    np.savez_compressed(path, **out)
    print(f"inn_step_cfg5_fp64.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(out)} arrays)")


if __name__ == "__main__":
    main()
