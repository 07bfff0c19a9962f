#!/usr/bin/env python3
"""Headline benchmark: ray-samples/sec through the full warp -> MLP -> composite path.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): nerf_inn_llff.yaml hyper-parameters on a 300x400 LLFF-fern
shaped scene, 18 views x 227 rays (4096 // 18) per GPU, 64 coarse + 128 fine hierarchical
samples (the fine network evaluates 192), rays produced by the NVP warp (barf_inn_llff get_pose).
One step = the reference's train iteration on one batch: ray generation, warp, sampling, both
MLPs, both composites, inverse-CDF resampling, photometric loss, full backward (NeRF, fine NeRF,
warp, latents), gradient all-reduce (N > 1) and the Adam updates.  Synthetic images, reference
initialisation (+ N(0, 0.02) on the warp's zero-initialised layers so the warp is non-trivial).
Exact fp32 MFMA arithmetic (v_mfma_f32_32x32x2_f32); nothing is skipped or cached.

ray-samples = MLP evaluations per step = rays x (64 + 192), coarse and fine both counted
(SURVEY section 8d).  Weak scaling: per-GPU work is fixed, rank r renders pixels idx[r::N] of a
global draw N times larger.  Prints ONE JSON line on rank 0.

Besides the contract fields the line carries `roofline` (dominant single MLP kernel: algorithmic FLOPs / mean launch time
from device events on the launch stream vs the fp32-MFMA peak; `traffic` from the PMC passes in profiles/), `kernels` (per-kernel
device-event averages of the timed steps), at N = 1 `forward_only` (one full 300x400 image through the eval path of the same
graph, timed outside the training region; skip with --no-forward-only) and `cpu_baseline` (the CPU oracle's identical step on
the host cores, bounded sample; skip with --no-cpu-baseline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_FWD = 2 * 527872           # algorithmic GEMM FLOPs per MLP evaluation
PEAK_FP32_MFMA = 157.3          # TFLOP/s, MI355X_MICROARCH.md chip table
PEAK_HBM = 8000.0               # GB/s spec


def cpu_baseline(B, S, Sf, H, W):
    """The CPU oracle (a restatement of the reference's PyTorch path, pinned to golden vectors) timed on
    this box's host cores on a bounded sample of the same workload: same views / resolution / samples per
    ray, fewer rays per view."""
    import torch
    from oracle import niw_oracle as O
    # cores this process may actually run on (a cgroup-limited box reports every host core in cpu_count)
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    threads = max(1, min(threads, 64))
    torch.set_num_threads(threads)
    R = 16                                                  # rays per view in the sample
    req = lambda d: {k: v.requires_grad_(True) for k, v in d.items()}
    pc, pf, wp = req(O.make_nerf_params(1)), req(O.make_nerf_params(2)), req(O.make_warp_params(3, 0.02))
    lat = O.make_latent(4, B).requires_grad_(True)
    gen = torch.Generator().manual_seed(0)
    image = torch.rand(B, 3, H, W, generator=gen)
    intr = torch.tensor([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]], dtype=torch.float32).repeat(B, 1, 1)
    w3, wv = O.c2f_weights(0.3, (0.1, 0.5), 10), O.c2f_weights(0.3, (0.1, 0.5), 4)
    times = []
    for i in range(3):
        ray_idx = torch.randperm(H * W, generator=gen)[:R]
        u = torch.rand(B, R, S, 1, generator=gen)
        t0 = time.perf_counter()
        out = O.inn_train_step(pc, wp, lat, image, intr, ray_idx, u, H, W, S, (1, 0), "inverse", 0.3, nerf_fine_p=pf, Sf=Sf, w3d=w3, wview=wv)
        out["loss"].backward()
        times.append(time.perf_counter() - t0)
    evals = B * R * (S + S + Sf)
    best = min(times[1:])
    return dict(value=evals / best, unit="ray-samples/s", cores=threads, kind="port",
                sample=f"{B} views x {R} rays x ({S}+{S + Sf}) samples = {evals} MLP evals per step, fwd+bwd, best of 2 after 1 warm-up, torch CPU {threads} threads")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-forward-only", action="store_true", help="skip the full-image eval render (e.g. when profiling the train step)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from neural_invertible_warp_amd import configs, engine, ops, parallel

    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # NIW_DIST_BACKEND=gloo + fewer GPUs than ranks is a logic test of the N>1 path on a 1-GPU box
    backend = os.environ.get("NIW_DIST_BACKEND")
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    rank, world, _ = parallel.init_from_env(backend=backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = f"cuda:{local}"

    B, rays_per_gpu = 18, 4096
    opt = configs.cfg2_nerf_inn_llff_hier(device=dev)
    opt.nerf.rand_rays = rays_per_gpu * world                # global draw; each rank keeps idx[rank::world]
    S, Sf = opt.nerf.sample_intvs, opt.nerf.sample_intvs_fine
    trainer = engine.INNTrainer(opt, B, rank=rank, world=world, warp_perturb=0.02)
    var0 = engine.synthetic_scene(opt, B)
    R = (opt.nerf.rand_rays // B + world - 1 - rank) // world   # rays per view on this rank
    evals_local = B * R * (S + S + Sf)

    def step():
        var = type(var0)(var0)
        return trainer.train_iteration(var)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ops.TIMING.enabled = True
    ops.TIMING.reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    ops.TIMING.enabled = False
    kern = ops.TIMING.summary()

    tt = torch.tensor([dt, float(evals_local)], device=dev, dtype=torch.float64)
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, evals_total = float(tmax[0]), float(tsum[1])
    else:
        evals_total = float(evals_local)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # per-kernel rates from the device events recorded during the timed steps
    kernels = {}
    for name, (n, ms, units) in kern.items():
        entry = dict(launches=n, avg_ms=round(ms, 4), samples_per_launch=units)
        if name.startswith("mlp_fwd"):
            entry["tflops"] = units * FLOP_FWD / (ms * 1e-3) / 1e12
        elif name == "mlp_bwd_dx":
            entry["tflops"] = units * FLOP_FWD / (ms * 1e-3) / 1e12          # dX chain: same MAC count as the forward
        elif name == "mlp_bwd_dw":
            entry["tflops"] = units * FLOP_FWD / (ms * 1e-3) / 1e12          # dW: same MAC count (incl. reduce kernels)
        elif name == "composite_fwd":
            entry["gbps"] = units * 24.2 / (ms * 1e-3) / 1e9                  # 20 B/sample in + 4 B/sample prob + 20 B/ray out
        elif name == "composite_bwd":
            entry["gbps"] = units * 40.2 / (ms * 1e-3) / 1e9                  # re-reads 20 + 4 (d_prob) B, writes 16 B per sample
        kernels[name] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in entry.items()}
    # roofline: the dominant SINGLE kernel (mlp_bwd_dw is a group of 12 GEMM + 12 reduce launches and is listed
    # under `kernels` only), so that its average can be checked against one row of the rocprofv3 summary
    rocprof_name = {"mlp_fwd_train": "mlp_fwd_kernel<true>", "mlp_fwd": "mlp_fwd_kernel<false>", "mlp_bwd_dx": "mlp_bwd_dx_kernel"}
    mlp = {k: v for k, v in kernels.items() if k in rocprof_name}
    dom = max(mlp, key=lambda k: mlp[k]["avg_ms"] * mlp[k]["launches"]) if mlp else None
    roofline = None
    if dom:
        a = kernels[dom]["tflops"]
        # HBM bytes per launch of that kernel: rocprofv3 PMC passes recorded in profiles/r1_traffic.json
        # (FETCH_SIZE x2 + WRITE_SIZE, bytes per sample) times the samples one launch processes
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r1_traffic.json")) as f:
                t = json.load(f)["bytes_per_sample"].get(dom)
            if t:
                traffic = dict(value=round((t["fetch_corrected"] + t["write"]) * kernels[dom]["samples_per_launch"] / 1e9, 3), unit="GB",
                               source="profiles/r1_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes)")
        except (OSError, KeyError, ValueError):
            pass
        roofline = dict(bound="mfma", kernel=rocprof_name[dom], achieved=a, peak=PEAK_FP32_MFMA, unit="TFLOP/s", frac=round(a / PEAK_FP32_MFMA, 4), traffic=traffic)

    out = dict(metric="ray-samples/sec (warp+MLP+composite) on LLFF-fern, 1/2/4/8 GPUs + PSNR parity",
               value=evals_total * args.steps / dt, unit="ray-samples/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32",
               data="synthetic",
               config=dict(workload="cfg2: nerf_inn_llff.yaml fern 300x400, 18 views x 227 rays x (64 coarse + 192 fine) per GPU, "
                                    "NVP-warped rays, fwd+bwd+Adam", rays_per_gpu=B * R, samples_per_ray="64+192",
                           mlp_evals_per_step_per_gpu=evals_local, parallelism=f"ray-shard dp{world}", precision="exact fp32 MFMA"),
               loss=float(loss.all.detach()), roofline=roofline, kernels=kernels)
    if world == 1 and not args.no_forward_only:
        # forward-only figure of SURVEY section 8d: one full 300x400 image through the eval path of the same graph
        # (render_by_slices: slices of rand_rays rays, coarse + fine networks), outside the timed training region
        with torch.no_grad():
            g = trainer.graph
            pose1, intr1 = torch.eye(3, 4, device=dev)[None], var0.intr[:1]
            g.render_by_slices(opt, pose1, intr=intr1, mode="eval")
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                g.render_by_slices(opt, pose1, intr=intr1, mode="eval")
            torch.cuda.synchronize()
            t_img = (time.perf_counter() - t1) / 2
            # the same image in the largest slices one launch takes (nerf.eval_slice_rays; results are slice-independent)
            opt.nerf.eval_slice_rays = opt.H * opt.W
            g.render_by_slices(opt, pose1, intr=intr1, mode="eval")
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                g.render_by_slices(opt, pose1, intr=intr1, mode="eval")
            torch.cuda.synchronize()
            t_big = (time.perf_counter() - t1) / 2
            opt.nerf.eval_slice_rays = None
        n_eval = opt.H * opt.W * (S + S + Sf)
        out["forward_only"] = dict(value=n_eval / t_img, unit="ray-samples/s", ms_per_image=round(t_img * 1e3, 2),
                                   workload=f"one {opt.H}x{opt.W} image, {S} coarse + {S + Sf} fine samples per ray, "
                                            f"{-(-opt.H * opt.W // opt.nerf.rand_rays)} slices of {opt.nerf.rand_rays} rays",
                                   frac_of_fwd_roofline=round(n_eval / t_img * FLOP_FWD / 1e12 / PEAK_FP32_MFMA, 4),
                                   largest_slices=dict(value=n_eval / t_big, ms_per_image=round(t_big * 1e3, 2),
                                                       frac_of_fwd_roofline=round(n_eval / t_big * FLOP_FWD / 1e12 / PEAK_FP32_MFMA, 4)))
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(B, S, Sf, opt.H, opt.W)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
