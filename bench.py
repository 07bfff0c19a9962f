#!/usr/bin/env python3
"""Headline benchmark: ray-samples/sec through the full warp -> MLP -> composite path.

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg3|cfg4|cfg4-<scene>|cfg5] [--scaling weak|strong] [--shard-of K]
    (N > 1: launched by torch.distributed.run)

Default workload = BASELINE.json configs[1] ("cfg2"): nerf_inn_llff.yaml hyper-parameters on a 300x400 LLFF-fern shaped scene,
18 views x 227 rays (4096 // 18) per GPU, 64 coarse + 128 fine hierarchical samples (the fine network evaluates 192), rays
produced by the NVP warp (barf_inn_llff get_pose).  The other BASELINE configs are selectable:
    cfg3          scripts/train_llff.sh:1   barf_inn_llff fern: 18 views x 113 rays x 128 samples, c2f encoding, Kabsch alignment loss x 1e4
    cfg4-<scene>  scripts/train_llff.sh:1-8 the same for one of the 8 LLFF scenes (18 ... 56 train views, 2048 // B rays per view)
    cfg4          all 8 scenes, one train iteration of each per step (8 resident trainers)
    cfg5          scripts/train_dtu.sh:6    barf_inn_dtu: 3 views x 682 rays x 128 samples, metric depth [1.2, 5.2], noisy initial poses,
                                            alignment loss x 1e3
One step = the reference's train iteration on one batch: ray generation, warp, sampling, MLP(s), composite(s), inverse-CDF
resampling (cfg2), losses, full backward (NeRF, fine NeRF, warp, latents), gradient all-reduce (N > 1) and the Adam updates.
Synthetic images, reference initialisation (+ N(0, 0.02) on the warp's zero-initialised layers so the warp is non-trivial).
Exact fp32 MFMA arithmetic (v_mfma_f32_32x32x2_f32); nothing is skipped or cached.

ray-samples = MLP evaluations per step (coarse and fine both counted).  Scaling: `weak` (default) keeps the
per-GPU ray count fixed (the global draw is N times larger; every rank warps all of it and renders a contiguous 1/N share of the
B x R rays); `strong` keeps the reference's GLOBAL batch (4096 / 2048 rays) and splits it over the ranks.  `--shard-of K` (N = 1 only) runs rank 0's 1/K shard of the global batch
on one GPU: a proxy of what one rank of a K-GPU strong-scaled job executes (no collective).  Prints ONE JSON line on rank 0.

At N = 1 the timed iterations replay ONE captured HIP graph each (engine.INNTrainer(hip_graph=True): forward, backward, gradient
gather and the Adam updates; the step's scalars -- c2f bands, warp windows, Adam bias corrections, pixel-draw number -- travel in a
256-byte device buffer refreshed before every replay).  Under N > 1 the default is launch-by-launch (--hip-graph on: two graphs with
the RCCL all-reduce issued eagerly between them); the step is GPU-bound at every BASELINE size, replay and eager launch time alike.

Besides the contract fields the line carries `roofline` (dominant single MLP kernel: algorithmic FLOPs / mean launch time from
device events on the launch stream vs the fp32-MFMA peak; `traffic` from the PMC passes in profiles/), `kernels` (per-kernel
device-event averages of the timed steps), and at N = 1: `composite_scan` (the compositing kernels alone at full-image size: achieved
HBM GB/s), `forward_only` (one full 300x400 image through the eval path), `psnr_parity` (HIP path vs CPU oracle, bounded run) and
`cpu_baseline` (the CPU oracle's identical step on the host cores, bounded sample).  Each can be skipped with --no-<name>.
"""
import argparse
import math
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_FWD = 2 * 527872           # algorithmic GEMM FLOPs per MLP evaluation
PEAK_FP32_MFMA = 157.3          # TFLOP/s, MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA = 2500.0         # TFLOP/s dense (the guide's ~2.5 PF; never the 2:1-sparsity figure)
PEAK_HBM = 8000.0               # GB/s spec (6290 GB/s measured streaming copy)


def cpu_baseline(B, S, Sf, H, W, ga_weight=None):
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
    R = max(2, 288 // B)                                    # rays per view in the sample (16 at 18 views)
    req = lambda d: {k: v.requires_grad_(True) for k, v in d.items()}
    pc, wp = req(O.make_nerf_params(1)), req(O.make_warp_params(3, 0.02))
    pf = req(O.make_nerf_params(2)) if Sf else None
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
        out = O.inn_train_step(pc, wp, lat, image, intr, ray_idx, u, H, W, S, (1, 0), "inverse", 0.3, nerf_fine_p=pf, Sf=Sf,
                               ga_weight=ga_weight, w3d=w3, wview=wv)
        out["loss"].backward()
        times.append(time.perf_counter() - t0)
    evals = B * R * (S + (S + Sf if Sf else 0))
    best = min(times[1:])
    out = dict(value=evals / best, unit="ray-samples/s", cores=threads, kind="port",
               sample=f"{B} views x {R} rays x ({S}" + (f"+{S + Sf}" if Sf else "") + f") samples = {evals} MLP evals per step, fwd+bwd, "
                      f"best of 2 after 1 warm-up, torch CPU {threads} threads")
    # the port must time within +-10 % of the imported reference; measured in the build container it takes 0.895 x
    # the reference's time (it forms the un-warped ray grid once per step where the reference forms it twice), i.e. it flatters the
    # CPU by 10.5 %: the figure the reference itself would reach on these cores is reported beside it
    try:
        with open(os.path.join(ROOT, "profiles", "r2_oracle_calibration.json")) as f:
            ratio = float(json.load(f)["oracle_over_reference_time"])
        out["calibration"] = dict(oracle_over_reference_time=ratio, source="profiles/r2_oracle_calibration.json (tools/calibrate_oracle.py, build container)",
                                  reference_equivalent_value=round(out["value"] * ratio, 1))
    except (OSError, KeyError, ValueError):
        pass
    return out


def rocprof_row(config, kernel):
    """The committed rocprofv3 --kernel-trace --stats summary of this same command (profiles/r3_kernel_stats_<config>.csv, else
    round 2's): the row of `kernel`, so that the device-event average of this run stands next to the profiler's."""
    import csv
    for rnd in ("r3", "r2"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats_{config}.csv")
        try:
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    name = row.get("Name", "")
                    if kernel in name:
                        return dict(file=f"profiles/{rnd}_kernel_stats_{config}.csv", name=name, calls=int(row["Calls"]),
                                    avg_ms=round(float(row["AverageNs"]) / 1e6, 4))
        except (OSError, KeyError, ValueError):
            continue
    return None


def torch_rocm_baseline(dev, B, R, S, Sf, H, W, ga_weight=None, depth_range=(1, 0), param="inverse"):
    """The oracle's identical train step (PyTorch autograd, fp32) at the FULL shapes of the workload on this same MI355X, through
    torch's own ROCm kernels (hipBLASLt / rocBLAS GEMMs, ATen elementwise): what the reference's algorithm costs on this node
    when simply run under PyTorch-ROCm.  Reported beside cpu_baseline; outside the timed region; the oracle is the thing timed
    here, never the product."""
    import torch
    from oracle import niw_oracle as O
    to = lambda d: {k: v.to(dev).requires_grad_(True) for k, v in d.items()}
    pc, wp = to(O.make_nerf_params(1)), to(O.make_warp_params(3, 0.02))
    pf = to(O.make_nerf_params(2)) if Sf else None
    lat = O.make_latent(4, B).to(dev).requires_grad_(True)
    gen = torch.Generator(device=dev).manual_seed(0)
    image = torch.rand(B, 3, H, W, device=dev, generator=gen)
    intr = torch.tensor([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]], dtype=torch.float32).repeat(B, 1, 1).to(dev)
    w3, wv = O.c2f_weights(0.3, (0.1, 0.5), 10), O.c2f_weights(0.3, (0.1, 0.5), 4)
    times = []
    for i in range(4):
        ray_idx = torch.randperm(H * W, device=dev, generator=gen)[:R]
        u = torch.rand(B, R, S, 1, device=dev, generator=gen)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = O.inn_train_step(pc, wp, lat, image, intr, ray_idx, u, H, W, S, depth_range, param, 0.3, nerf_fine_p=pf, Sf=Sf,
                               ga_weight=ga_weight, w3d=w3, wview=wv)
        out["loss"].backward()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        for prm in list(pc.values()) + list(wp.values()) + (list(pf.values()) if pf else []) + [lat]:
            prm.grad = None
    del out
    torch.cuda.empty_cache()
    evals = B * R * (S + (S + Sf if Sf else 0))
    best = min(times[1:])
    return dict(value=evals / best, unit="ray-samples/s", ms_per_step=round(best * 1e3, 3), kind="port",
                device=torch.cuda.get_device_name(dev), torch=torch.__version__,
                sample=f"full shapes: {B} views x {R} rays x ({S}" + (f"+{S + Sf}" if Sf else "") + f") samples = {evals} MLP evals, fwd+bwd (no optimizer), "
                       "best of 3 after 1 warm-up, fp32, PyTorch-ROCm eager")


def composite_scan(dev, iters=20):
    """The compositing kernels alone at the size where they reach HBM (one 300x400 image: 120,000 rays x 192 samples, 0.55 GB forward):
    achieved ALGORITHMIC bytes per second.  The launches go straight through the C ABI into pre-allocated buffers, `iters` of them
    back to back between two device events on the launch stream, so the average is kernel time (plus the ~1.5 us kernel boundary), not
    allocator or Python time.  tools/composite_bench.py is the stand-alone version the rocprofv3 PMC passes of
    profiles/r2_composite_traffic.json run."""
    import torch
    from neural_invertible_warp_amd import _lib, ops
    N, S = 120000, 192
    gen = torch.Generator(device=dev).manual_seed(5)
    ray = torch.randn(N, 3, device=dev, generator=gen)
    rgb_s = torch.rand(N, S, 3, device=dev, generator=gen)
    sig = torch.rand(N, S, device=dev, generator=gen) * 2
    dep = (torch.rand(N, S, device=dev, generator=gen) * 0.9 / S + torch.arange(S, device=dev) / S + 1.0).contiguous()
    g_rgb = torch.randn(N, 3, device=dev, generator=gen)
    rgb, depth, opa, prob = (torch.empty(N, 3, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, S, device=dev))
    d_rgb_s, d_sig, d_ray = torch.empty_like(rgb_s), torch.empty_like(sig), torch.empty_like(ray)
    P, st = ops._p, ops._stream()

    def fwd():
        _lib.call("niw_composite_fwd", P(ray), P(rgb_s), P(sig), P(dep), N, S, 0, 0.0, P(rgb), P(depth), P(opa), P(prob), st)

    def bwd():
        _lib.call("niw_composite_bwd", P(ray), P(rgb_s), P(sig), P(dep), N, S, 0, 0.0, P(g_rgb), None, None, None, P(d_rgb_s), P(d_sig), P(d_ray), st)

    def timed(fn):
        fn(); fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters                     # ms per launch

    def isolated(fn, reps=5):
        """one launch between two events on an otherwise idle, drained device: the kernel's own duration (what a profiler's
        begin / end timestamps show), without the write-back of the previous launch's dirty lines that a back-to-back train pays"""
        ms = []
        for _ in range(reps):
            torch.cuda.synchronize()
            time.sleep(0.002)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        return sorted(ms)[len(ms) // 2]

    tf, tb = timed(fwd), timed(bwd)
    tf1, tb1 = isolated(fwd), isolated(bwd)
    bf, bb = N * S * 24 + N * 32, N * S * 36 + N * 36
    return dict(workload=f"{N} rays x {S} samples (one 300x400 image, fine pass)", bound="hbm", unit="GB/s", peak=PEAK_HBM, achievable=6290.0,
                timing="us / achieved: 20 launches back to back (steady state: every launch also pays the write-back of its predecessor's dirty "
                       "lines -- the backward leaves 370 MB); us_isolated: a single launch on a drained device = the kernel's own duration, the "
                       "figure a profiler's dispatch timestamps give (profiles/r3_composite_traffic.json)",
                fwd=dict(kernel=f"composite_fwd_kernel<64>", bytes=bf, us=round(tf * 1e3, 1), us_isolated=round(tf1 * 1e3, 1), achieved=round(bf / tf / 1e6, 1), frac_of_peak=round(bf / tf / 1e6 / PEAK_HBM, 4),
                         frac_of_achievable=round(bf / tf / 1e6 / 6290.0, 4)),
                bwd=dict(kernel=f"composite_bwd_kernel<64>", bytes=bb, us=round(tb * 1e3, 1), us_isolated=round(tb1 * 1e3, 1), achieved=round(bb / tb / 1e6, 1), frac_of_peak=round(bb / tb / 1e6 / PEAK_HBM, 4),
                         frac_of_achievable=round(bb / tb / 1e6 / 6290.0, 4)),
                bytes_per_sample=dict(fwd="12 rgb + 4 sigma + 4 depth in, 4 prob out", bwd="20 in, 12 d_rgb + 4 d_sigma out"),
                traffic_source="profiles/r3_composite_traffic.json (rocprofv3 FETCH_SIZE x2 / WRITE_SIZE of the same launches)")


def build_workloads(name, dev, rank, world, scaling, shard_of, hip_graph=True, precision="fp32"):
    """-> (list of (trainer, var0, B, R_local, S, Sf), description, rays of the global batch per scene)"""
    from neural_invertible_warp_amd import configs, engine
    eff_world, eff_rank = (shard_of, 0) if shard_of else (world, rank)
    out, desc = [], None

    def mk(opt, B, rays, warp_perturb=0.02, dtu=False):
        opt.arch.precision = precision                                              # arithmetic of the field MLP (include/niw.h enum niw_precision)
        opt.nerf.rand_rays = rays * (eff_world if scaling == "weak" else 1)        # global draw; each rank keeps idx[rank::world]
        if dtu:
            var0, init = engine.synthetic_dtu_scene(opt, B)
            tr = engine.INNTrainer(opt, B, rank=eff_rank, world=eff_world, warp_perturb=warp_perturb, initial_poses_w2c=init, hip_graph=hip_graph)
        else:
            var0 = engine.synthetic_scene(opt, B)
            tr = engine.INNTrainer(opt, B, rank=eff_rank, world=eff_world, warp_perturb=warp_perturb, hip_graph=hip_graph)
        from neural_invertible_warp_amd import parallel
        lo, hi = parallel.flat_share(B * (opt.nerf.rand_rays // B), eff_rank, eff_world)   # this rank's contiguous share of the B x R rays
        S = opt.nerf.sample_intvs
        Sf = opt.nerf.sample_intvs_fine if opt.nerf.fine_sampling else 0
        out.append((tr, var0, B, (hi - lo) / B, S, Sf))                              # (rays per view: fractional for a share)
        return opt

    if name == "cfg2":
        mk(configs.cfg2_nerf_inn_llff_hier(device=dev), 18, 4096)
        desc = "cfg2: nerf_inn_llff.yaml fern 300x400, 18 views x 227 rays x (64 coarse + 192 fine), NVP-warped rays, fwd+bwd+Adam"
    elif name == "cfg3":
        mk(configs.cfg3_barf_inn_llff(device=dev), 18, 2048)
        desc = "cfg3: barf_inn_llff.yaml fern 300x400 (scripts/train_llff.sh:1), 18 views x 113 rays x 128, c2f PE, Kabsch alignment 1e4, fwd+bwd+Adam"
    elif name.startswith("cfg4"):
        scenes = list(configs.LLFF_TRAIN_VIEWS) if name == "cfg4" else [name.split("-", 1)[1]]
        for sc in scenes:
            if sc not in configs.LLFF_TRAIN_VIEWS:
                raise SystemExit(f"unknown LLFF scene {sc!r}; choose from {list(configs.LLFF_TRAIN_VIEWS)}")
            mk(configs.cfg3_barf_inn_llff(device=dev), configs.LLFF_TRAIN_VIEWS[sc], 2048)
        desc = ("cfg4: barf_inn_llff.yaml, LLFF scenes " + ",".join(f"{s}({configs.LLFF_TRAIN_VIEWS[s]} views)" for s in scenes) +
                ", 2048 // views rays per view x 128, one train iteration of every scene per step")
    elif name == "cfg5":
        mk(configs.cfg5_barf_inn_dtu(device=dev), 3, 2048, dtu=True)
        desc = "cfg5: barf_inn_dtu.yaml scan65-shaped 300x400 (scripts/train_dtu.sh:6), 3 views x 682 rays x 128, metric depth [1.2,5.2], noisy_gt poses, alignment 1e3"
    else:
        raise SystemExit(f"unknown --config {name}")
    return out, desc


def needs_launcher(gpus, env):
    """True when this process was started as a plain `python bench.py --gpus N` with N > 1: nobody has set up the ranks
    (torch.distributed.run exports WORLD_SIZE / RANK / LOCAL_RANK for its children)."""
    return gpus > 1 and "WORLD_SIZE" not in env and "RANK" not in env


def launcher_command(gpus, argv, port=None):
    """The child command line: one torch.distributed.run agent that starts `gpus` ranks of this same script with the same arguments
    (rendezvous on 127.0.0.1: the container host name may not resolve)."""
    port = port or (29500 + os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks_if_needed(gpus, argv):
    """`python bench.py --gpus N` (N > 1) with no rank environment: start the N ranks as a CHILD process and exit with its return code.
    The parent has not imported torch or touched HIP at this point and never does (a GPU-initialised process must not exec or fork
    rank processes); a failing rank makes torch.distributed.run -- and therefore this process -- exit non-zero."""
    if not needs_launcher(gpus, os.environ):
        return False
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this pool (RCCL needs it across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // gpus)))
    rc = subprocess.call(launcher_command(gpus, argv), env=env)
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None, help="default: weak (strong with --shard-of)")
    ap.add_argument("--shard-of", type=int, default=0, help="N=1 only: run rank 0's 1/K share of the global batch on this GPU: the work of one rank of a "
                                                            "K-GPU job (strong scaling: the reference's batch split K ways; with --scaling weak: a K times larger batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-forward-only", action="store_true", help="skip the full-image eval render (e.g. when profiling the train step)")
    ap.add_argument("--no-composite-scan", action="store_true")
    ap.add_argument("--no-psnr-parity", action="store_true")
    ap.add_argument("--no-torch-baseline", action="store_true", help="skip the oracle's step under PyTorch-ROCm on this GPU (torch_rocm_baseline)")
    ap.add_argument("--lean", action="store_true", help="train step only: all four --no-* switches")
    ap.add_argument("--hip-graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the captured HIP graph of the iteration (on) or launch its kernels one by one (off); auto = on for one GPU, off under "
                         "torch.distributed (measured: no throughput difference at any BASELINE size -- the step is GPU-bound -- so the multi-rank "
                         "default avoids capturing next to a live RCCL communicator)")
    ap.add_argument("--no-hip-graph", action="store_true", help="same as --hip-graph off")
    ap.add_argument("--kernel-steps", type=int, default=3, help="extra eager steps after the timed region for the per-kernel device-event table (0: skip)")
    ap.add_argument("--precision", choices=["fp32", "bf16x3", "bf16"], default="fp32",
                    help="arithmetic of the field MLP.  fp32 (default, the headline): exact fp32 MFMA.  bf16x3 / bf16: the opt-in fast modes "
                         "(split-bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulation) -- a SEPARATE line with its own parity row "
                         "(tests/test_gpu_fast_precision.py), never comparable with the reference's fp32 tolerance")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the torch.distributed process group even for ONE rank, so that the gradient all-reduce really goes through RCCL "
                         "(hardware evidence of the N > 1 code path on a 1-GPU box)")
    args = ap.parse_args()
    if args.lean:
        args.no_cpu_baseline = args.no_forward_only = args.no_composite_scan = args.no_psnr_parity = args.no_torch_baseline = True

    if launch_ranks_if_needed(args.gpus, sys.argv[1:]):
        return                                              # (never reached: the launcher exits with the children's code)

    import torch
    import torch.distributed as dist
    from neural_invertible_warp_amd import ops, parallel

    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # NIW_DIST_BACKEND=gloo + fewer GPUs than ranks is a logic test of the N>1 path on a 1-GPU box
    backend = os.environ.get("NIW_DIST_BACKEND")
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    rank, world, _ = parallel.init_from_env(backend=backend, force=args.force_dist)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # every rank contributes a one: the sum is the number of ranks the collective back end really connected
    ranks_seen, dist_backend = 1, None
    if dist.is_available() and dist.is_initialized():
        ones = torch.ones(1, device=f"cuda:{local}", dtype=torch.float32)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen, dist_backend = int(round(float(ones))), dist.get_backend()
        if ranks_seen != world:
            raise SystemExit(f"bench.py: the {dist_backend} all-reduce saw {ranks_seen} ranks, expected {world}")
    assert not (args.shard_of and world > 1), "--shard-of is a single-GPU proxy"
    dev = f"cuda:{local}"
    scaling = args.scaling or ("strong" if args.shard_of else "weak")

    use_graph = False if args.no_hip_graph else ((world == 1 and not args.force_dist) if args.hip_graph == "auto" else args.hip_graph == "on")
    loads, desc = build_workloads(args.config, dev, rank, world, scaling, args.shard_of, hip_graph=use_graph, precision=args.precision)
    exact = args.precision == "fp32"
    peak_mfma = PEAK_FP32_MFMA if exact else PEAK_BF16_MFMA
    evals_local = sum(int(round(B * R)) * (S + (S + Sf if Sf else 0)) for _, _, B, R, S, Sf in loads)

    # the batch tensors stay resident at fixed addresses (the captured graph reads them in place)
    def step(replay=True):
        loss = None
        for tr, var0, *_ in loads:
            loss = tr.train_iteration(type(var0)(var0), replay=replay)
        return loss

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 3)):              # >= 3: two eager steps, then the capture
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    loss_value = float(loss.all.detach())
    if not math.isfinite(loss_value):
        raise SystemExit(f"bench.py: the training loss is {loss_value} after {max(args.warmup, 3) + args.steps} iterations -- a timing of a diverged "
                         "run is not a measurement")
    graphed = all(tr._captured is not None for tr, *_ in loads)
    # per-kernel device events: a few more iterations launched one by one (events cannot be recorded inside a replayed graph);
    # outside the timed region, same kernels, same shapes.  Two untimed launch-by-launch iterations come first: the eager path draws
    # its 19 GB of transient buffers from the caching allocator's general pool, not from the graph's private one, and the first
    # iterations there pay for fresh blocks (round 2: a table measured without them summed to MORE than the step it decomposes).
    # The table is checked before it is believed: its launches must sum to no more than the eager iteration they were recorded in,
    # and that iteration must take what a timed (replayed) one takes.
    kern, eager_ms, kernel_check = {}, None, None
    if args.kernel_steps > 0:
        for attempt in range(2):
            for _ in range(2):
                step(replay=False)
            fence()
            ops.TIMING.enabled = True
            ops.TIMING.reset()
            t1 = time.perf_counter()
            for _ in range(args.kernel_steps):
                step(replay=False)
            fence()
            eager_ms = (time.perf_counter() - t1) / args.kernel_steps * 1e3
            ops.TIMING.enabled = False
            kern = ops.TIMING.summary()
            kernel_sum_ms = sum(n * ms for n, ms, _ in kern.values()) / args.kernel_steps
            ms_timed = dt / args.steps * 1e3
            # (launch by launch the host can be the limit -- ~50 launches per iteration against a 1.2 ms shard step -- so the eager
            # iteration may take longer than a replayed one; what must hold is that the kernels fit inside both)
            kernel_check = dict(kernel_sum_ms=round(kernel_sum_ms, 4), eager_ms_per_step=round(eager_ms, 4), timed_ms_per_step=round(ms_timed, 4),
                                eager_host_bound=bool(eager_ms > 1.05 * ms_timed),
                                consistent=bool(kernel_sum_ms <= eager_ms and kernel_sum_ms <= 1.01 * ms_timed))
            if kernel_check["consistent"]:
                break
        if not kernel_check["consistent"]:
            # a table that does not add up is not evidence: keep the record of the failed check, drop the table
            print(f"bench.py: per-kernel table rejected {kernel_check}", file=sys.stderr, flush=True)
            kern = {}

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
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    # per-kernel rates from the device events recorded during the timed steps
    kernels = {}
    for name, (n, ms, units) in kern.items():
        entry = dict(launches=n, avg_ms=round(ms, 4), samples_per_launch=units)
        if name.startswith("mlp_fwd") or name in ("mlp_bwd_dx", "mlp_bwd_dw"):
            entry["tflops"] = units * FLOP_FWD / (ms * 1e-3) / 1e12          # dX chain and dW (incl. its reduce kernels): the forward's MAC count each
        elif name == "composite_fwd":
            entry["gbps"] = units * 24.2 / (ms * 1e-3) / 1e9                  # 20 B/sample in + 4 B/sample prob + 32 B/ray
        elif name == "composite_bwd":
            entry["gbps"] = units * 36.2 / (ms * 1e-3) / 1e9                  # 20 B in, 16 B out per sample
        kernels[name] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in entry.items()}
    # roofline: the dominant SINGLE kernel (mlp_bwd_dw is a group of GEMM + reduce launches and is listed
    # under `kernels` only), so that its average can be checked against one row of the rocprofv3 summary
    rocprof_name = {"mlp_fwd_train": "mlp_fwd_kernel<true>", "mlp_fwd": "mlp_fwd_kernel<false>", "mlp_bwd_dx": "mlp_bwd_dx_kernel"} if exact else \
        {"mlp_fwd_train": "mlp_fwd_fast_kernel", "mlp_fwd": "mlp_fwd_fast_kernel", "mlp_bwd_dx": "mlp_bwd_dx_fast_kernel"}
    mlp = {k: v for k, v in kernels.items() if k in rocprof_name}
    dom = max(mlp, key=lambda k: mlp[k]["avg_ms"] * mlp[k]["launches"]) if mlp else None
    roofline = None
    if dom:
        a = kernels[dom]["tflops"]
        # HBM bytes per launch of that kernel: rocprofv3 PMC passes recorded in profiles/ (FETCH_SIZE x2 + WRITE_SIZE, bytes per
        # sample) times the samples one launch processes; a pointer to the committed measurement, not measured in this run
        traffic = None
        for fname in ("r3_traffic.json", "r2_traffic.json", "r1_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", fname)) as f:
                    t = json.load(f)["bytes_per_sample"].get(dom)
                if t:
                    traffic = dict(value=round((t["fetch_corrected"] + t["write"]) * kernels[dom]["samples_per_launch"] / 1e9, 3), unit="GB",
                                   source=f"profiles/{fname} (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes)")
                    break
            except (OSError, KeyError, ValueError):
                pass
        roofline = dict(bound="mfma", kernel=rocprof_name[dom], achieved=a, peak=peak_mfma, unit="TFLOP/s", frac=round(a / peak_mfma, 4), traffic=traffic,
                        avg_ms=kernels[dom]["avg_ms"], samples_per_launch=kernels[dom]["samples_per_launch"], flop_per_sample=FLOP_FWD,
                        rocprof=rocprof_row(args.config, rocprof_name[dom]))
    elif kernel_check is not None:
        # no trustworthy per-kernel table: the whole step against the train roofline (3 x forward FLOPs per sample) is all that can be claimed
        a = evals_total / world * args.steps / dt * 3 * FLOP_FWD / 1e12
        roofline = dict(bound="mfma", kernel="whole train step (per-kernel table rejected)", achieved=round(a, 4), peak=peak_mfma, unit="TFLOP/s",
                        frac=round(a / peak_mfma, 4), traffic=None)

    ms_step = dt / args.steps * 1e3
    value = evals_total * args.steps / dt
    par = f"ray-shard dp{world}" + (f" ({scaling} scaling)" if world > 1 else "") + (f"; 1/{args.shard_of} shard of the global batch" if args.shard_of else "")
    out = dict(metric="ray-samples/sec (warp+MLP+composite) on LLFF-fern, 1/2/4/8 GPUs + PSNR parity",
               value=value, unit="ray-samples/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=ms_step, higher_is_better=True, scaling=scaling, vs_baseline=None, dtype="f32" if exact else args.precision, data="synthetic",
               config=dict(workload=desc, name=args.config, rays_per_gpu=sum(int(round(B * R)) for _, _, B, R, _, _ in loads),
                           samples_per_ray="+".join(str(x) for x in ((loads[0][4], loads[0][4] + loads[0][5]) if loads[0][5] else (loads[0][4],))),
                           mlp_evals_per_step_per_gpu=evals_local, parallelism=par,
                           precision="exact fp32 MFMA" if exact else
                           f"{args.precision}: OPT-IN fast mode, split-bf16 operands on v_mfma_f32_32x32x16_bf16 with fp32 accumulation ("
                           + ("hi*hi + hi*mid + mid*hi, 16 significand bits per operand" if args.precision == "bf16x3" else "leading plane only, 8 bits") +
                           "); not the headline, own parity row in tests/test_gpu_fast_precision.py"),
               frac_of_train_roofline=round(value / world * 3 * FLOP_FWD / 1e12 / peak_mfma, 4),
               loss=loss_value, hip_graph=graphed, ranks_seen=ranks_seen, backend=dist_backend, roofline=roofline, kernel_check=kernel_check, kernels=kernels)
    g, opt, var0 = loads[0][0].graph, loads[0][0].opt, loads[0][1]
    S, Sf = loads[0][4], loads[0][5]
    if world == 1 and not args.no_composite_scan:
        out["composite_scan"] = composite_scan(dev)
    if world == 1 and not args.no_forward_only and not args.shard_of:
        # forward-only figure of SURVEY section 8d: one full image through the eval path of the same graph, outside the timed
        # training region.  Under no_grad render_by_slices is ONE niw_render_fwd call; the stage-by-stage sweep in the reference's
        # slices of rand_rays rays (bit-identical image, tests/test_gpu_render_call.py) is timed beside it.
        with torch.no_grad():
            pose1, intr1 = torch.eye(3, 4, device=dev)[None], var0.intr[:1]
            kw = dict(depth_range=[1.2, 5.2]) if args.config == "cfg5" else {}

            def time_image(render, reps=3):
                render()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    render()
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / reps

            t_img = time_image(lambda: g.render_by_slices(opt, pose1, intr=intr1, mode="eval", **kw))
            t_sweep = time_image(lambda: g._sweep_image(opt, lambda first, count: g._render_pixels(opt, pose1, intr1, "eval", pixel_range=(first, count), **kw)))
        n_eval = opt.H * opt.W * (S + (S + Sf if Sf else 0))
        frac = lambda t: round(n_eval / t * FLOP_FWD / 1e12 / peak_mfma, 4)
        out["forward_only"] = dict(value=n_eval / t_img, unit="ray-samples/s", ms_per_image=round(t_img * 1e3, 2),
                                   workload=f"one {opt.H}x{opt.W} image, {S} coarse" + (f" + {S + Sf} fine" if Sf else "") + " samples per ray, "
                                            "one niw_render_fwd call",
                                   frac_of_fwd_roofline=frac(t_img),
                                   stage_by_stage=dict(value=n_eval / t_sweep, ms_per_image=round(t_sweep * 1e3, 2), frac_of_fwd_roofline=frac(t_sweep),
                                                       slices=f"{-(-opt.H * opt.W // opt.nerf.rand_rays)} of {opt.nerf.rand_rays} rays"))
    if world == 1 and not args.no_psnr_parity:
        # the "+ PSNR parity" half of the metric, bounded: 10 identical optimisation steps on the HIP path and on the CPU oracle
        from oracle import parity
        pg, pc = parity.psnr_trajectories(dev, steps=10, precision=args.precision)
        out["psnr_parity"] = dict(steps=len(pg), max_abs_diff_db=round(max(abs(a - b) for a, b in zip(pg, pc)), 5),
                                  final_psnr_hip=round(pg[-1], 4), final_psnr_oracle=round(pc[-1], 4),
                                  sample="barf_inn_llff, 3 views x 16 rays x 32 samples on 12x16 images, identical weights / pixel draws / stratified draws, "
                                         "photometric PSNR of every step, HIP engine vs CPU oracle (autograd + torch.optim.Adam)")
    ga = {"cfg3": 4, "cfg5": 3}.get(args.config, 4 if args.config.startswith("cfg4") else None)
    if world == 1 and not args.no_torch_baseline and not args.shard_of:
        rng_kw = dict(depth_range=(1.2, 5.2), param="metric") if args.config == "cfg5" else {}
        try:
            out["torch_rocm_baseline"] = torch_rocm_baseline(dev, loads[0][2], int(round(loads[0][3])), S, Sf, opt.H, opt.W, ga_weight=ga, **rng_kw)
            out["torch_rocm_baseline"]["speedup_of_this_build"] = round(value / out["torch_rocm_baseline"]["value"], 2)
        except torch.cuda.OutOfMemoryError as e:          # a reported side figure must not cost the line
            out["torch_rocm_baseline"] = dict(error=f"out of memory: {e}"[:200])
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(loads[0][2], S, Sf, opt.H, opt.W, ga_weight=ga)
    print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
