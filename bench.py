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

An iteration is ONE library call (niw_train_step: forward, losses, backward of every stage, 25 launches) + the gradient all-reduce
(N > 1) + ONE Adam launch.  At N = 1 the timed iterations replay a captured HIP graph of it (the step's scalars -- c2f bands, warp
windows, Adam bias corrections, pixel-draw number -- travel in a 256-byte device buffer refreshed before every replay); under N > 1
the default is launch by launch, which since round 4 is the FASTER form at every size (the host needs ~0.3 ms per iteration, a rank's
1/8 share of cfg3 takes 1.0 ms: measured 1.005 ms launched, 1.023 ms replayed).  `--hip-graph on` under N > 1 captures two graphs around
the eagerly issued RCCL all-reduce; should the capture fail, the rank says so in a marker file and exits, and the launcher parent --
which never touched the GPU -- starts a fresh set of ranks with `--hip-graph off` (a failed capture is not recoverable in-process).

N > 1 prints ONE line holding both scaling modes: the weak-scaled step is the headline (`value`, `ms_per_step`: 4096 / 2048 rays per
GPU), and `strong` = the reference's own batch (model/nerf_inn_llff.py:510: one pixel set of rand_rays // B per view) split N ways,
timed right after it in the same processes; `comm_ms` = device events around the flat gradient all-reduce (max over ranks).  The run
exits non-zero when a loss is not finite or the ranks' parameters disagree after the last step.

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
XGMI_LINK_GBPS = 153.0          # GB/s per xGMI link (7 links per GPU, point to point): a ring all-reduce is bound by ONE link per neighbour


def wire_ms(n_bytes, world):
    """Algorithmic wire time of a ring all-reduce of n_bytes over `world` GPUs: every rank sends (and receives) 2 (N - 1) / N x the
    buffer over one xGMI link per neighbour.  A measured comm_ms far above this is latency (2 (N - 1) hops), not bandwidth."""
    if world < 2 or not n_bytes:
        return 0.0
    return 2.0 * (world - 1) / world * n_bytes / (XGMI_LINK_GBPS * 1e9) * 1e3


def host_flags_agree(tag, ok, rank, world):
    """HOST-ONLY exchange of one boolean per rank through the process group's key-value store (no device work, no collective): used
    where a rank's HIP state may have become unusable (a failed graph capture is sticky in-process) and its peers must learn of it BEFORE
    they enter a collective that rank will never join.  -> list of the ranks that reported False."""
    import torch.distributed as dist
    store = dist.distributed_c10d._get_default_store()
    store.set(f"niw_flag/{tag}/{rank}", "1" if ok else "0")
    return [r for r in range(world) if store.get(f"niw_flag/{tag}/{r}") != b"1"]


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(B, S, Sf, H, W, ga_weight=None, vanilla=False, budget_s=60.0, timed_steps=3):
    """The CPU oracle (a restatement of the reference's PyTorch path, pinned to golden vectors) timed on this box's host cores on a
    bounded sample of the same workload: same views / resolution / samples per ray, and since round 5 HALF the rays of the batch (113
    per view at 18 views; the whole 227 when a step fits the budget): at 16 rays per view (rounds 1-4) the fixed per-step work -- the
    full-resolution ray grid, the warp's parameter preparation -- was a fifth of a step that is 2 % of one at the real size.
    Threads: torch's CPU path does not speed up monotonically with threads on a box of unknown topology / cgroup quota, so a short
    step (16 rays per view) is timed at 8 / 16 / 32 / 64 threads (no more than the cores the process may run on) and the best count is used.
    vanilla: BASELINE configs[0] (ground-truth poses, ReLU density, metric depth [0,1], no warp: model/nerf.py:251-288).
    The oracle makes the reference's discarded first ray-grid call too (`reference_cost`), so that it costs what the reference costs
    (profiles/r5_oracle_calibration.json)."""
    import torch
    from oracle import niw_oracle as O
    # cores this process may actually run on (a cgroup-limited box reports every host core in cpu_count)
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    req = lambda d: {k: v.requires_grad_(True) for k, v in d.items()}
    pc, wp = req(O.make_nerf_params(1)), req(O.make_warp_params(3, 0.02))
    pf = req(O.make_nerf_params(2)) if Sf else None
    lat = O.make_latent(4, B).requires_grad_(True)
    gen = torch.Generator().manual_seed(0)
    image = torch.rand(B, 3, H, W, generator=gen)
    intr = torch.tensor([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]], dtype=torch.float32).repeat(B, 1, 1)
    pose = torch.eye(3, 4).repeat(B, 1, 1)
    w3, wv = O.c2f_weights(0.3, (0.1, 0.5), 10), O.c2f_weights(0.3, (0.1, 0.5), 4)
    params = list(pc.values()) + list(wp.values()) + (list(pf.values()) if pf else []) + [lat]

    def one_step(R):
        ray_idx = torch.randperm(H * W, generator=gen)[:R]
        u = torch.rand(B, R, S, 1, generator=gen)
        for prm in params:
            prm.grad = None
        t0 = time.perf_counter()
        if vanilla:
            center, ray = O.center_and_ray(H, W, pose, intr)
            out = O.render_rays(pc, center[:, ray_idx], ray[:, ray_idx], u, S, (0, 1), "metric", p_fine=pf, Sf=Sf, density_activ="relu")
            target = O.gather_pixels(image, ray_idx)
            loss = O.mse_loss(out["rgb"], target) + (O.mse_loss(out["rgb_fine"], target) if Sf else 0.0)
        else:
            out = O.inn_train_step(pc, wp, lat, image, intr, ray_idx, u, H, W, S, (1, 0), "inverse", 0.3, nerf_fine_p=pf, Sf=Sf,
                                   ga_weight=ga_weight, w3d=w3, wview=wv, reference_cost=True)
            loss = out["loss"]
        loss.backward()
        return time.perf_counter() - t0

    per_ray = S + (S + Sf if Sf else 0)
    t_begin = time.perf_counter()
    # 1. thread count: a short step at every candidate, best of two after a warm-up
    R_small = min(max(2, 288 // B), H * W)
    sweep = {}
    for n in sorted({min(n, affinity) for n in (8, 16, 32, 64)}):          # (all 256 hardware threads of a GPU box: 80 s per step, 90 x the best)
        torch.set_num_threads(n)
        warm = one_step(R_small)
        if sweep and warm > 4 * min(sweep.values()):                        # hopeless count: its warm-up step is the record
            sweep[n] = warm
            continue
        sweep[n] = min(one_step(R_small), one_step(R_small))
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    # 2. the figure (1 warm-up + >= 3 timed steps): the full batch if the budget allows (estimated from the short
    # step; cfg2's 14 s step does: ~57 s), else half the batch's rays per view, halved again while the estimate is twice the budget
    R_full = None
    est = sweep[threads] / (B * R_small * per_ray)                    # seconds per evaluation at the short step (an over-estimate)
    R_half = max(R_small, 2048 // B)
    rays_of_batch = min(H * W, 4096 // B if Sf and not vanilla else 2048 // B if not vanilla else 1024 // B)
    n_steps = 1 + max(3, int(timed_steps))
    R = rays_of_batch if est * B * rays_of_batch * per_ray * n_steps < budget_s else min(R_half, rays_of_batch)
    while R > R_small and est * B * R * per_ray * n_steps > 2 * budget_s:
        R = max(R_small, R // 2)
    times = [one_step(R) for _ in range(n_steps)]
    evals = B * R * per_ray
    timed = sorted(times[1:])
    best, median = timed[0], timed[len(timed) // 2]
    fwd_only = None
    if not vanilla:
        # forward only (BASELINE.md section 4 asks for both figures): the same step without the backward pass, one timed call
        with torch.no_grad():
            ray_idx = torch.randperm(H * W, generator=gen)[:R]
            u = torch.rand(B, R, S, 1, generator=gen)
            t0 = time.perf_counter()
            O.inn_train_step(pc, wp, lat, image, intr, ray_idx, u, H, W, S, (1, 0), "inverse", 0.3, nerf_fine_p=pf, Sf=Sf, ga_weight=ga_weight,
                             w3d=w3, wview=wv, reference_cost=True)
            fwd_only = evals / (time.perf_counter() - t0)
    out = dict(value=evals / best, value_median=evals / median, forward_only_value=fwd_only, cpu_model=cpu_model_name(),
               unit="ray-samples/s", cores=threads, kind="port",
               sample=f"{B} views x {R} rays x ({S}" + (f"+{S + Sf}" if Sf else "") + f") samples = {evals} MLP evals per step (the batch has {rays_of_batch} "
                      f"rays per view), fwd+bwd, {len(timed)} timed steps after 1 warm-up (value = best, value_median = median), torch CPU {threads} threads",
               seconds_per_step=[round(t, 3) for t in times[1:]], affinity_cores=affinity, host_cores=os.cpu_count(),
               thread_sweep={str(n): dict(seconds_per_step=round(t, 4), value=round(B * R_small * per_ray / t, 1)) for n, t in sweep.items()},
               thread_sweep_sample=f"{B} views x {R_small} rays, best of 2 after 1 warm-up at each count; the count with the shortest step times the figure",
               wall_s=round(time.perf_counter() - t_begin, 1))
    # the port must time within +-10 % of the imported reference -- measured in the build container at the cfg3 batch
    # itself (tools/calibrate_oracle.py); the figure the reference would reach on these cores is reported beside the port's
    if not vanilla:
        for fname in ("r5_oracle_calibration.json", "r2_oracle_calibration.json"):
            try:
                with open(os.path.join(ROOT, "profiles", fname)) as f:
                    ratio = float(json.load(f)["oracle_over_reference_time"])
                out["calibration"] = dict(oracle_over_reference_time=ratio, source=f"profiles/{fname} (tools/calibrate_oracle.py, build container)",
                                          reference_equivalent_value=round(out["value"] * ratio, 1))
                break
            except (OSError, KeyError, ValueError):
                pass
    return out


def rocprof_row(config, kernel):
    """The committed rocprofv3 --kernel-trace --stats summary of this same command (profiles/r4_kernel_stats_<config>.csv, else an
    earlier round's): the row of `kernel`, so that the device-event average of this run stands next to the profiler's."""
    import csv
    for rnd in ("r6", "r5", "r4", "r3", "r2"):
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
                       "figure a profiler's dispatch timestamps give (profiles/r5_composite_traffic.json)",
                fwd=dict(kernel="composite_fwd_span_kernel<3, 4, nt>", bytes=bf, us=round(tf * 1e3, 1), us_isolated=round(tf1 * 1e3, 1), achieved=round(bf / tf / 1e6, 1), frac_of_peak=round(bf / tf / 1e6 / PEAK_HBM, 4),
                         frac_of_achievable=round(bf / tf / 1e6 / 6290.0, 4)),
                bwd=dict(kernel="composite_bwd_span_kernel<3, 4, nt>", bytes=bb, us=round(tb * 1e3, 1), us_isolated=round(tb1 * 1e3, 1), achieved=round(bb / tb / 1e6, 1), frac_of_peak=round(bb / tb / 1e6 / PEAK_HBM, 4),
                         frac_of_achievable=round(bb / tb / 1e6 / 6290.0, 4)),
                bytes_per_sample=dict(fwd="12 rgb + 4 sigma + 4 depth in, 4 prob out", bwd="20 in, 12 d_rgb + 4 d_sigma out"),
                traffic_source="profiles/r5_composite_traffic.json (rocprofv3 FETCH_SIZE x2 / WRITE_SIZE of the same launches)")


def scene_from_disk(opt, root, scene, dev):
    """--llff-root: the resident batch of a REAL LLFF scene, read by the parser that is pinned to the reference's (data/llff.py <->
    /root/reference data/llff.py:28-72): train split, images resized to opt.H x opt.W, intrinsics and poses from poses_bounds.npy.
    -> (var0, number of train views)"""
    from neural_invertible_warp_amd.data import llff
    from neural_invertible_warp_amd.util import edict
    opt.data.root, opt.data.scene, opt.data.dataset = root, scene, "llff"
    batch = llff.Dataset(opt, split="train").prefetch_all_data(opt)
    return edict({k: batch[k].to(dev) for k in ("idx", "image", "intr", "pose")}), int(batch.image.shape[0])


def scenes_of_rank(scenes, rank, world):
    """replica placement: scene i trains whole on rank i mod N (N = 8: one scene per GPU, scripts/train_llff.sh:1-8 side by side)"""
    return [sc for i, sc in enumerate(scenes) if i % world == rank]


def build_workloads(name, dev, rank, world, scaling, shard_of, hip_graph=True, precision="fp32", overlap=True, placement="shard", llff_root=None,
                    split_exchange="auto", fused_step="auto"):
    """-> (list of (trainer, var0, B, R_local, S, Sf), description).  placement (cfg4, all scenes): "shard" = every rank trains every scene on
    its share of the rays (gradient all-reduce per scene and step); "replicas" = scene i trains WHOLE on rank i mod N, no exchange at all
    (/root/reference scripts/train_llff.sh:1-8: eight independent runs)."""
    from neural_invertible_warp_amd import configs, engine
    eff_world, eff_rank = (shard_of, 0) if shard_of else (world, rank)
    replicas = placement == "replicas"
    tr_world, tr_rank = (1, 0) if replicas else (eff_world, eff_rank)
    out, desc = [], None
    data = []

    def mk(opt, B, rays, warp_perturb=0.02, dtu=False, scene=None):
        opt.arch.precision = precision                                              # arithmetic of the field MLP (include/niw.h enum niw_precision)
        opt.nerf.rand_rays = rays * (tr_world if scaling == "weak" else 1)         # global draw; each rank renders a contiguous 1/N of it
        var0 = None
        if llff_root and scene and not dtu and os.path.isdir(os.path.join(llff_root, scene)):
            var0, B = scene_from_disk(opt, llff_root, scene, dev)
            data.append(f"llff:{scene}")
        else:
            data.append("synthetic")
        kw = dict(rank=tr_rank, world=tr_world, warp_perturb=warp_perturb, hip_graph=hip_graph, overlap=overlap, collectives=not replicas,
                  split_exchange=split_exchange, fused_step=fused_step)
        if dtu:
            var0, init = engine.synthetic_dtu_scene(opt, B)
            tr = engine.INNTrainer(opt, B, initial_poses_w2c=init, **kw)
        else:
            var0 = var0 if var0 is not None else engine.synthetic_scene(opt, B)
            tr = engine.INNTrainer(opt, B, **kw)
        tr.scene_name = scene or name
        from neural_invertible_warp_amd import parallel
        lo, hi = parallel.flat_share(B * (opt.nerf.rand_rays // B), tr_rank, tr_world)   # this rank's contiguous share of the B x R rays
        S = opt.nerf.sample_intvs
        Sf = opt.nerf.sample_intvs_fine if opt.nerf.fine_sampling else 0
        out.append((tr, var0, B, (hi - lo) / B, S, Sf))                              # (rays per view: fractional for a share)
        return opt

    if name == "cfg1":
        # BASELINE configs[0]: the vanilla model on ground-truth poses (no warp, no ray gradients); its engine is NeRFTrainer
        opt = configs.cfg1_nerf_llff_repr(device=dev)
        opt.arch.precision = precision
        if eff_world > 1:
            raise SystemExit("cfg1 (vanilla NeRF, configs[0]) is a single-GPU line")
        B = 18
        var0 = engine.synthetic_scene(opt, B)
        tr = engine.NeRFTrainer(opt, B, fused_step=fused_step)
        out.append((tr, var0, B, opt.nerf.rand_rays // B, opt.nerf.sample_intvs, opt.nerf.sample_intvs_fine if opt.nerf.fine_sampling else 0))
        desc = ("cfg1: nerf_llff_repr.yaml 300x400 (configs[0]), 18 views x 56 rays x (64 coarse + 192 fine), ReLU density + noise, metric depth [0,1], "
                "ground-truth poses (no warp, no ray gradients), fwd+bwd+Adam; one niw_train_step call (warp_params = NULL) unless --fused-step off")
    elif name == "cfg2":
        mk(configs.cfg2_nerf_inn_llff_hier(device=dev), 18, 4096, scene="fern")
        desc = "cfg2: nerf_inn_llff.yaml fern 300x400, 18 views x 227 rays x (64 coarse + 192 fine), NVP-warped rays, fwd+bwd+Adam"
    elif name == "cfg3":
        mk(configs.cfg3_barf_inn_llff(device=dev), 18, 2048, scene="fern")
        desc = "cfg3: barf_inn_llff.yaml fern 300x400 (scripts/train_llff.sh:1), 18 views x 113 rays x 128, c2f PE, Kabsch alignment 1e4, fwd+bwd+Adam"
    elif name.startswith("cfg4"):
        scenes = list(configs.LLFF_TRAIN_VIEWS) if name == "cfg4" else [name.split("-", 1)[1]]
        for sc in scenes:
            if sc not in configs.LLFF_TRAIN_VIEWS:
                raise SystemExit(f"unknown LLFF scene {sc!r}; choose from {list(configs.LLFF_TRAIN_VIEWS)}")
        mine = scenes_of_rank(scenes, eff_rank, eff_world) if replicas else scenes
        for sc in mine:
            mk(configs.cfg3_barf_inn_llff(device=dev), configs.LLFF_TRAIN_VIEWS[sc], 2048, scene=sc)
        desc = ("cfg4: barf_inn_llff.yaml, LLFF scenes " + ",".join(f"{s}({configs.LLFF_TRAIN_VIEWS[s]} views)" for s in scenes) +
                ", 2048 // views rays per view x 128, one train iteration of every scene per step" +
                (f"; placement replicas: scene i whole on rank i mod {eff_world}, no gradient exchange (this rank: {','.join(mine) or 'none'})" if replicas else ""))
    elif name == "cfg5":
        mk(configs.cfg5_barf_inn_dtu(device=dev), 3, 2048, dtu=True)
        desc = "cfg5: barf_inn_dtu.yaml scan65-shaped 300x400 (scripts/train_dtu.sh:6), 3 views x 682 rays x 128, metric depth [1.2,5.2], noisy_gt poses, alignment 1e3"
    else:
        raise SystemExit(f"unknown --config {name}")
    build_workloads.data = sorted(set(data)) or ["synthetic"]
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


CAPTURE_MARKER_ENV = "NIW_CAPTURE_FAILED_FILE"      # a rank whose HIP-graph capture failed touches this file before it exits


def launch_ranks_if_needed(gpus, argv):
    """`python bench.py --gpus N` (N > 1) with no rank environment: start the N ranks as a CHILD process and exit with its return code.
    The parent has not imported torch or touched HIP at this point and never does (a GPU-initialised process must not exec or fork
    rank processes); a failing rank makes torch.distributed.run -- and therefore this process -- exit non-zero.  One recovery is the
    parent's to make: a rank whose graph capture failed (HIP's capture error is sticky in-process) leaves a marker file, and the
    parent then starts a FRESH set of ranks with `--hip-graph off` appended."""
    if not needs_launcher(gpus, os.environ):
        return False
    import shutil
    import subprocess
    import tempfile
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this pool (RCCL needs it across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // gpus)))
    # the marker lives in a directory of this parent's own (mkdtemp: mode 0700, unguessable name): a stale or foreign file under a
    # predictable /tmp name could otherwise turn any rank failure into a "capture failure" and mask the original error behind a retry
    marker_dir = tempfile.mkdtemp(prefix="niw_bench_")
    marker = os.path.join(marker_dir, "capture_failed")
    env[CAPTURE_MARKER_ENV] = marker
    try:
        rc = subprocess.call(launcher_command(gpus, argv), env=env)
        if rc != 0 and os.path.exists(marker):
            os.remove(marker)
            print("bench.py: HIP-graph capture failed on a rank (exit code 75 there); starting fresh ranks with --hip-graph off -- the line will say "
                  "capture_fallback: true", file=sys.stderr, flush=True)
            env["NIW_CAPTURE_FALLBACK"] = "1"
            rc = subprocess.call(launcher_command(gpus, list(argv) + ["--hip-graph", "off"], port=29500 + (os.getpid() + 977) % 2000), env=env)
    finally:
        shutil.rmtree(marker_dir, ignore_errors=True)
    sys.exit(rc)


def report_capture_failure_and_exit(err):
    """called by a rank: leave the marker for the launcher parent (if there is one) and exit with a distinct code"""
    marker = os.environ.get(CAPTURE_MARKER_ENV)
    if marker:
        try:
            open(marker, "w").write(str(err)[:500])
        except OSError:
            pass
    print(f"bench.py: {err}", file=sys.stderr, flush=True)
    os._exit(75)                                             # no destructors: the process's HIP state is unusable after a failed capture


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="default: N = 1 weak (strong with --shard-of); N > 1 BOTH -- the weak-scaled step is the headline, `strong` holds the other")
    ap.add_argument("--shard-of", type=int, default=0, help="N=1 only: run rank 0's 1/K share of the global batch on this GPU: the work of one rank of a "
                                                            "K-GPU job (strong scaling: the reference's batch split K ways; with --scaling weak: a K times larger batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-forward-only", action="store_true", help="skip the full-image eval render (e.g. when profiling the train step)")
    ap.add_argument("--no-composite-scan", action="store_true")
    ap.add_argument("--no-psnr-parity", action="store_true")
    ap.add_argument("--no-torch-baseline", action="store_true", help="skip the oracle's step under PyTorch-ROCm on this GPU (torch_rocm_baseline)")
    ap.add_argument("--lean", action="store_true", help="train step only: all four --no-* switches")
    ap.add_argument("--hip-graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the captured HIP graph of the iteration (on) or launch it call by call (off); auto = on for one GPU, off under "
                         "torch.distributed (round 4: launched is the faster form at every size, also for a 1/8 share -- 1.005 vs 1.023 ms).  With "
                         "`on` under N > 1 a failed capture makes the launcher parent start fresh ranks with `off`")
    ap.add_argument("--no-hip-graph", action="store_true", help="same as --hip-graph off")
    ap.add_argument("--kernel-steps", type=int, default=3, help="extra eager steps after the timed region for the per-kernel device-event table (0: skip)")
    ap.add_argument("--precision", choices=["fp32", "bf16x3", "bf16"], default="fp32",
                    help="arithmetic of the field MLP.  fp32 (default, the headline): exact fp32 MFMA.  bf16x3 / bf16: the opt-in fast modes "
                         "(split-bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulation) -- a SEPARATE line with its own parity row "
                         "(tests/test_gpu_fast_precision.py), never comparable with the reference's fp32 tolerance")
    ap.add_argument("--overlap", choices=["on", "off"], default="on",
                    help="niw_train_desc.overlap: the small independent stages of an iteration on the library's second stream beside the field-MLP kernels (default) "
                         "or everything on one stream in stage order")
    ap.add_argument("--placement", choices=["both", "shard", "replicas"], default="both",
                    help="cfg4 (all 8 LLFF scenes) on N GPUs: `shard` = every scene ray-sharded N ways (one gradient all-reduce per scene and step), "
                         "`replicas` = scene i whole on rank i mod N with no exchange at all (the reference's own way: scripts/train_llff.sh:1-8 are "
                         "eight independent runs); `both` (default) times the shard placement as the line's headline and the replicas beside it "
                         "(`replicas` object).  With --shard-of K at N = 1: the scenes / shares of rank 0 of K")
    ap.add_argument("--llff-root", default=os.environ.get("NIW_LLFF_ROOT"),
                    help="directory holding real LLFF scenes (<root>/<scene>/images, poses_bounds.npy; also NIW_LLFF_ROOT): a scene found there feeds "
                         "the trainer instead of the synthetic one (data/llff.py, pinned to /root/reference data/llff.py:28-72) and `data` says so")
    ap.add_argument("--split-exchange", choices=["auto", "on", "off"], default="auto",
                    help="N > 1 with a fine network: the gradient exchange as two all-reduces, the fine network's segment on a communication stream "
                         "while the rest of the backward runs (auto: where the iteration is one launched call); off: one flat all-reduce")
    ap.add_argument("--ab", choices=["auto", "on", "off"], default="auto",
                    help="N > 1 (auto) or any live process group (on): after the headline, time the weak workload a few more steps with the split "
                         "gradient exchange forced on / off and launched vs replayed as a HIP graph -- `split_exchange: {on_ms, off_ms}`, "
                         "`hip_graph: {launched_ms, replayed_ms}` -- so that ONE hardware run answers both open questions of DESIGN section 5")
    ap.add_argument("--fused-step", choices=["auto", "off"], default="auto",
                    help="off: the iteration through the autograd mirror over the per-stage entry points instead of the one niw_train_step call")
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

    from neural_invertible_warp_amd import engine
    use_graph = False if args.no_hip_graph else ((world == 1 and not args.force_dist) if args.hip_graph == "auto" else args.hip_graph == "on")
    exact = args.precision == "fp32"
    peak_mfma = PEAK_FP32_MFMA if exact else PEAK_BF16_MFMA
    n_evals = lambda loads_: sum(int(round(B * R)) * (S + (S + Sf if Sf else 0)) for _, _, B, R, S, Sf in loads_)

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # untimed iterations in front of the timed ones: --warmup, at least 3 (two launch-by-launch iterations, then the capture), and at
    # least 16 when the iterations are LAUNCHED rather than replayed: the cross-stream events of the launched form (the second stream
    # of niw_train_step, the collective's stream) make the HIP runtime grow an internal pool ONCE -- a 50-100 ms host stall on about the
    # seventh iteration after start-up (tools/rccl_probe.py, round 4) that would otherwise sit inside the timed region
    n_warm = max(args.warmup, 3 if use_graph else 16)

    def timed_run(loads_, exit_on_capture_error=True):
        """warm-up + `--steps` timed iterations of one workload -> (seconds, final loss, replayed?, mean all-reduce ms | None, step fn)"""
        # the batch tensors stay resident at fixed addresses (a captured graph reads private copies of them, in place)
        def step(replay=True):
            loss_ = None
            for tr, var0, *_ in loads_:
                loss_ = tr.train_iteration(type(var0)(var0), replay=replay)
            return loss_
        try:
            for _ in range(n_warm):                       # >= 3: two launch-by-launch iterations, then the capture
                step()
        except engine.CaptureError as e:
            if not exit_on_capture_error:
                raise
            report_capture_failure_and_exit(e)
        for tr, *_ in loads_:
            if hasattr(tr, "comm_events"):
                tr.comm_events = []
        fence()
        trace = [] if os.environ.get("NIW_BENCH_TRACE") else None
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss_ = step()
            if trace is not None:
                trace.append(time.perf_counter())
        if trace is not None:
            torch.cuda.synchronize()
            trace.append(time.perf_counter())
        fence()
        dt_ = time.perf_counter() - t0
        if trace is not None:
            print("bench.py trace (ms): host per step " + " ".join(f"{1e3 * (b - a):.2f}" for a, b in zip([t0] + trace[:-2], trace[:-1])) +
                  f" | drain {1e3 * (trace[-1] - trace[-2]):.2f} | fence {1e3 * (t0 + dt_ - trace[-1]):.2f}", file=sys.stderr, flush=True)
        # device events of the gradient exchange (engine.INNTrainer._all_reduce): per iteration the collectives' own durations (summed: a split
        # exchange has two) and the EXPOSED part -- end of the backward on the main stream to the start of the optimizer
        entries = [e for tr, *_ in loads_ for e in (getattr(tr, "comm_events", None) or [])]
        comm = [sum(a.elapsed_time(b) for a, b in pairs) for pairs, _ in entries]
        exposed = [x.elapsed_time(y) for _, (x, y) in entries]
        for tr, *_ in loads_:
            if hasattr(tr, "comm_events"):
                tr.comm_events = None
        value_ = float(loss_.all.detach())
        if not math.isfinite(value_):
            # say what diverged and where: which rank, which trainer, every loss term, and whether parameters / gradients are finite
            terms = {k: float(v.detach()) for k, v in loss_.items()}
            state = [dict(scene=getattr(tr, "scene_name", None), it=tr.it, hip_graph=bool(getattr(tr, "hip_graph", False)),
                          split_exchange=str(getattr(tr, "split_exchange", None)),
                          finite_params=[bool(torch.isfinite(f).all()) for f in tr._flats()],
                          finite_grads=bool(torch.isfinite(tr.bucket.flat).all())) for tr, *_ in loads_]
            print(f"bench.py: rank {rank}: non-finite loss {terms}; trainers {state}", file=sys.stderr, flush=True)
            raise SystemExit(f"bench.py: the training loss is {value_} after {n_warm + args.steps} iterations -- a timing of a diverged "
                             "run is not a measurement")
        mean = lambda v: sum(v) / len(v) if v else None
        return dt_, value_, all(getattr(tr, "_captured", None) is not None for tr, *_ in loads_), (mean(comm), mean(exposed)), step

    class CaptureAbort(Exception):
        """a rank's graph capture failed and every rank has learnt of it (host-side) before any of them entered a replay's collective"""

    def guard_captures(loads_, tag):
        """wrap every trainer's capture so that its outcome is agreed on by all ranks through the store BEFORE the first replay: a rank
        whose capture failed can issue no further device work, and a peer that went on to replay would wait for it in the all-reduce"""
        for j, (tr, *_) in enumerate(loads_):
            def guarded(var, it, _orig=tr._capture, _tag=f"{tag}/{j}"):
                err = None
                try:
                    if os.environ.get("NIW_TEST_FAIL_CAPTURE_RANK") == str(rank):      # (tests: the abort protocol without a broken device)
                        raise engine.CaptureError("injected by NIW_TEST_FAIL_CAPTURE_RANK")
                    ok = _orig(var, it)
                except engine.CaptureError as e:
                    ok, err = False, e
                bad = host_flags_agree(_tag, bool(ok), rank, world)
                if bad:
                    raise CaptureAbort(f"capture failed on rank(s) {bad}" + (f": {err}" if err else ""))
                return ok
            tr._capture = guarded

    def over_ranks(dt_, evals_, comm_, loads_, check_params=True):
        """max / min over ranks of the timed seconds, sum of the evaluations, max of the all-reduce time; and the ranks' parameters must
        agree after the last step (same reduced gradients, same Adam): a checksum of every flat parameter buffer, min == max"""
        if world == 1:
            return dt_, dt_, float(evals_), comm_
        check = sum(float(f.double().sum()) for tr, *_ in loads_ for f in tr._flats()) if check_params else 0.0
        t = torch.tensor([dt_, -dt_, comm_[0] or 0.0, check, -check, comm_[1] or 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e = torch.tensor([float(evals_)], device=dev, dtype=torch.float64)
        dist.all_reduce(e, op=dist.ReduceOp.SUM)
        if float(t[3]) != -float(t[4]):
            raise SystemExit(f"bench.py: the ranks' parameters disagree after the timed steps (checksum max {float(t[3])!r}, min {-float(t[4])!r})")
        return float(t[0]), -float(t[1]), float(e[0]), ((float(t[2]) if comm_[0] is not None else None), (float(t[5]) if comm_[1] is not None else None))

    # under N > 1 (and no explicit --scaling) BOTH modes are timed, weak first: its numbers are the headline
    modes = [args.scaling or ("strong" if args.shard_of else "weak")]
    if world > 1 and args.scaling is None:
        modes = ["weak", "strong"]
    scaling = modes[0]
    placement = "replicas" if args.placement == "replicas" and args.config == "cfg4" else "shard"
    if placement == "replicas":
        modes = ["strong"]                                   # the total work is the eight scenes, whatever N is
        scaling = "strong"
    split = {"auto": "auto", "on": True, "off": False}[args.split_exchange]
    wl = dict(hip_graph=use_graph, precision=args.precision, overlap=args.overlap == "on", llff_root=args.llff_root, split_exchange=split,
              fused_step="auto" if args.fused_step == "auto" else False)
    loads, desc = build_workloads(args.config, dev, rank, world, scaling, args.shard_of, placement=placement, **wl)
    data_label = "+".join(build_workloads.data)
    evals_local = n_evals(loads)
    dt, loss_value, graphed, comm_ms, step = timed_run(loads) if loads else (0.0, 0.0, False, (None, None), None)
    ms_rank = dt / args.steps * 1e3
    # per-kernel device events: a few more iterations launched STAGE BY STAGE (niw_train_step(stage, stage + 1): events cannot be recorded
    # inside a replayed graph, nor between the stages of one call), on one stream, outside the timed region; same kernels, same shapes.
    # Two untimed iterations of that kind come first.  The table is checked before it is believed: its launches must sum to no more than
    # the serial iteration they were recorded in, and the field-MLP kernels -- which never overlap one another -- to no more than a timed
    # iteration (the small stages may: the timed iteration runs them on a second stream beside the MLP kernels, niw_train_desc.overlap).
    kern, eager_ms, kernel_check = {}, None, None
    if args.kernel_steps > 0:
        for attempt in range(2):
            ops.TIMING.enabled = True
            for _ in range(2):
                step(replay=False)
            fence()
            ops.TIMING.reset()
            t1 = time.perf_counter()
            for _ in range(args.kernel_steps):
                step(replay=False)
            fence()
            eager_ms = (time.perf_counter() - t1) / args.kernel_steps * 1e3
            ops.TIMING.enabled = False
            kern = ops.TIMING.summary()
            kernel_sum_ms = sum(n * ms for n, ms, _ in kern.values()) / args.kernel_steps
            mlp_sum_ms = sum(n * ms for k, (n, ms, _) in kern.items() if k.startswith("mlp_")) / args.kernel_steps
            kernel_check = dict(kernel_sum_ms=round(kernel_sum_ms, 4), mlp_kernel_sum_ms=round(mlp_sum_ms, 4), serial_ms_per_step=round(eager_ms, 4),
                                timed_ms_per_step=round(ms_rank, 4), hidden_by_overlap_ms=round(max(0.0, kernel_sum_ms - ms_rank), 4),
                                consistent=bool(kernel_sum_ms <= eager_ms and mlp_sum_ms <= ms_rank))
            # the retry is a COLLECTIVE decision under N > 1: every untimed iteration all-reduces the gradients, so a rank that retried alone
            # would leave its peers in the next collective (seen with two ranks time-slicing one GPU: one rank's table failed its check,
            # the other's did not)
            again = not kernel_check["consistent"]
            if world > 1:
                flag = torch.tensor([1.0 if again else 0.0], device=dev, dtype=torch.float64)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                again = bool(flag.item() > 0)
            if not again:
                break
        if not kernel_check["consistent"]:
            # a table that does not add up is not evidence: keep the record of the failed check, drop the table
            print(f"bench.py: per-kernel table rejected {kernel_check}", file=sys.stderr, flush=True)
            kern = {}
    dt, dt_min, evals_total, comm_ms = over_ranks(dt, evals_local, comm_ms, loads, check_params=placement != "replicas")
    comm_ms, comm_exposed_ms = comm_ms
    rays_per_gpu = sum(int(round(B * R)) for _, _, B, R, _, _ in loads)
    S0, Sf0 = loads[0][4], loads[0][5]

    # the other scaling mode, same processes, right after (N > 1): the reference's own batch split over the ranks
    strong = None
    if len(modes) > 1:
        step = None                                   # (releases the closure over the first workload's trainers)
        for tr, *_ in loads:
            if getattr(tr, "fused", None) is not None:
                tr.fused.ws = None
        torch.cuda.empty_cache()        # (back to the device, not only to this process's allocator: ranks that time-slice ONE GPU share its memory)
        loads2, desc2 = build_workloads(args.config, dev, rank, world, modes[1], 0, **wl)
        e2 = n_evals(loads2)
        dt2, loss2, graphed2, comm2, _ = timed_run(loads2)
        dt2, dt2_min, e2_total, comm2 = over_ranks(dt2, e2, comm2, loads2)
        comm2, comm2_exposed = comm2
        v2 = e2_total * args.steps / dt2
        strong = dict(scaling=modes[1], ms_per_step=dt2 / args.steps * 1e3, ms_per_step_fastest_rank=dt2_min / args.steps * 1e3, value=v2, unit="ray-samples/s",
                      mlp_evals_per_step_per_gpu=e2, rays_per_gpu=sum(int(round(B * R)) for _, _, B, R, _, _ in loads2),
                      frac_of_train_roofline=round(v2 / world * 3 * FLOP_FWD / 1e12 / peak_mfma, 4), comm_ms=None if comm2 is None else round(comm2, 4),
                      comm_exposed_ms=None if comm2_exposed is None else round(comm2_exposed, 4), loss=loss2, hip_graph=graphed2, workload=desc2 + f"; the reference's batch split over {world} ranks")
        for tr, *_ in loads2:
            if getattr(tr, "fused", None) is not None:
                tr.fused.ws = None
        del loads2
        torch.cuda.empty_cache()

    # cfg4: the other placement of the eight scenes, same processes, right after -- scene i WHOLE on rank i mod N, no gradient exchange
    # ("report both").  Also the line's own figures when --placement replicas was asked for.
    def scene_table(loads_):
        """per-scene milliseconds of this rank's scenes (a few fenced iterations of each, launched), gathered over the ranks"""
        rows = []
        for tr, var0, B, R, S, Sf in loads_:
            for _ in range(3):
                tr.train_iteration(type(var0)(var0))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(max(5, args.steps)):
                tr.train_iteration(type(var0)(var0))
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t1) / max(5, args.steps) * 1e3
            ev = int(round(B * R)) * (S + (S + Sf if Sf else 0))
            rows.append(dict(scene=getattr(tr, "scene_name", None), rank=rank, views=B, rays=int(round(B * R)), mlp_evals=ev, ms_per_step=round(ms, 4),
                             frac_of_train_roofline=round(ev / (ms * 1e-3) * 3 * FLOP_FWD / 1e12 / peak_mfma, 4)))
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, rows)
            rows = [r for part in gathered for r in part]
        return rows

    replicas = None
    eff_world = args.shard_of or world
    if args.config == "cfg4" and (placement == "replicas" or (args.placement == "both" and eff_world > 1)):
        if placement == "replicas":
            loads_r, desc_r, timing_r = loads, desc, (dt, dt_min, evals_total, loss_value, graphed)
        else:
            step = None
            for tr, *_ in loads:
                if getattr(tr, "fused", None) is not None:
                    tr.fused.ws = None
            torch.cuda.empty_cache()
            loads_r, desc_r = build_workloads(args.config, dev, rank, world, "strong", args.shard_of, placement="replicas", **wl)
            e_r = n_evals(loads_r)
            dt_r, loss_r, graphed_r, comm_r, _ = timed_run(loads_r) if loads_r else (0.0, 0.0, False, (None, None), None)
            dt_r, dt_r_min, e_r_total, _ = over_ranks(dt_r, e_r, comm_r, loads_r, check_params=False)
            timing_r = (dt_r, dt_r_min, e_r_total, loss_r, graphed_r)
        dt_r, dt_r_min, e_r_total, loss_r, graphed_r = timing_r
        v_r = e_r_total * args.steps / dt_r if dt_r > 0 else 0.0
        replicas = dict(placement="replicas", ms_per_step=dt_r / args.steps * 1e3, ms_per_step_fastest_rank=dt_r_min / args.steps * 1e3, value=v_r,
                        unit="ray-samples/s", comm_ms=0, comm_exposed_ms=0, gradient_exchange="none: every scene trains whole on one GPU",
                        frac_of_train_roofline=round(v_r / eff_world * 3 * FLOP_FWD / 1e12 / peak_mfma, 4) if not args.shard_of else
                        round(v_r * 3 * FLOP_FWD / 1e12 / peak_mfma, 4),
                        scenes=scene_table(loads_r), hip_graph=graphed_r, workload=desc_r,
                        note="a step = one train iteration of every scene this rank holds; ranks holding more scenes (N < 8: scene i on rank i mod N) set the pace")
        if placement != "replicas":
            for tr, *_ in loads_r:
                if getattr(tr, "fused", None) is not None:
                    tr.fused.ws = None
            del loads_r
            torch.cuda.empty_cache()

    # bucket bytes one iteration exchanges (all trainers of the step) and their algorithmic wire time over one xGMI link per neighbour
    bucket_bytes = sum(4 * tr.bucket.flat.numel() for tr, *_ in loads if getattr(tr, "collectives", False))
    auto_split = bool(loads and hasattr(loads[0][0], "_overlapped_exchange") and loads[0][0]._overlapped_exchange())

    # A/B legs (N > 1): the headline workload a few more timed steps (i) with the split gradient exchange forced off / on, launched,
    # (ii) flat exchange, launched vs replayed as HIP graphs -- in the same processes, so that one hardware run answers whether the
    # overlap pays over xGMI and whether the replayed form catches up when every rank's host also drives a communicator.  The replayed
    # leg comes LAST and its captures are agreed on host-side (guard_captures): a failed capture costs this field, never the line.
    split_ab, graph_ab, capture_abort = None, None, None
    want_ab = args.ab == "on" or (args.ab == "auto" and world > 1)
    if want_ab and dist.is_initialized() and placement == "shard" and loads and getattr(loads[0][0], "fused", None) is not None:
        def release(loads_):
            for tr, *_ in loads_:
                if getattr(tr, "fused", None) is not None:
                    tr.fused.ws = None
            torch.cuda.empty_cache()

        def leg(loads_):
            dt_, _, graphed_, comm_, _ = timed_run(loads_, exit_on_capture_error=False)
            mx, _, _, comm_ = over_ranks(dt_, 0, comm_, loads_, check_params=False)
            r4 = lambda x: None if x is None else round(x, 4)
            return dict(ms=r4(mx / args.steps * 1e3), comm_ms=r4(comm_[0]), comm_exposed_ms=r4(comm_[1]), hip_graph=graphed_)

        step = None
        release(loads)
        wl_ab = dict(wl, hip_graph=False, split_exchange=False)
        loads_ab, _ = build_workloads(args.config, dev, rank, world, scaling, 0, placement="shard", **wl_ab)
        off = leg(loads_ab)
        on = None
        if all(tr.bucket.n_head > 0 for tr, *_ in loads_ab):
            for tr, *_ in loads_ab:
                tr.split_exchange = True
            on = leg(loads_ab)
        split_ab = dict(off_ms=off["ms"], on_ms=None if on is None else on["ms"], off=off, on=on, auto_resolves_to="on" if auto_split else "off",
                        steps=args.steps, workload=f"{scaling}-scaled {args.config}, launched; off = one flat all-reduce behind the backward, on = the fine "
                        "network's segment all-reduced on a communication stream while the coarse / warp backward runs" +
                        ("" if on is not None else " (no fine network in this config: nothing to split)"))
        release(loads_ab)
        del loads_ab
        try:
            dist.distributed_c10d._get_default_store()          # (the host-side agreement on the captures needs the group's store)
            store_ok = True
        except Exception:  # noqa: BLE001
            store_ok = False
        if not store_ok:
            graph_ab = dict(launched_ms=off["ms"], replayed_ms=None, skipped="this torch build does not expose the process group's store: the "
                            "captures cannot be agreed on host-side, so the replayed leg is not attempted")
            loads_g = []
        else:
            loads_g, _ = build_workloads(args.config, dev, rank, world, scaling, 0, placement="shard", **dict(wl_ab, hip_graph=True))
            guard_captures(loads_g, "ab_graph")
        try:
            if not store_ok:
                raise StopIteration
            rep = leg(loads_g)
            graph_ab = dict(launched_ms=off["ms"], replayed_ms=rep["ms"] if rep["hip_graph"] else None, replayed=rep, steps=args.steps,
                            workload=f"{scaling}-scaled {args.config}, flat all-reduce; replayed = two captured graphs (forward + backward | Adam) around the "
                                     "eagerly issued all-reduce")
        except StopIteration:
            pass
        except (CaptureAbort, engine.CaptureError) as e:
            capture_abort = str(e)[:300]
            graph_ab = dict(launched_ms=off["ms"], replayed_ms=None, capture_failed=capture_abort)
        if capture_abort is None:
            release(loads_g)
        else:
            # (a one-rank --force-dist run: the side figures below would need the device)
            args.no_composite_scan = args.no_forward_only = args.no_psnr_parity = args.no_torch_baseline = True
        del loads_g

    def finish(code=0):
        """leave the process: normally through destroy_process_group; after an aborted capture some rank's HIP state is unusable, so every
        rank leaves without touching the device or the communicator again"""
        sys.stdout.flush()
        sys.stderr.flush()
        if capture_abort is not None:
            os._exit(code)
        if dist.is_initialized():
            dist.destroy_process_group()

    # a fingerprint of the trained state after the headline's timed steps: two runs of the same command line must print the same one (the
    # kernels and the exchange are deterministic) -- a cheap detector of anything that corrupts a step
    try:
        param_checksum = repr(sum(float(f.double().sum()) for tr, *_ in loads for f in tr._flats())) if capture_abort is None else None
    except RuntimeError:
        param_checksum = None
    # peak device memory of the run's trainers (torch's allocator: workspaces, batch, parameters), max over ranks
    peak_mem = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    if world > 1 and capture_abort is None:
        pm = torch.tensor([peak_mem], device=dev, dtype=torch.float64)
        dist.all_reduce(pm, op=dist.ReduceOp.MAX)
        peak_mem = float(pm[0])
    if rank != 0:
        finish()
        return

    # per-kernel rates from the device events recorded during the timed steps
    kernels = {}
    for name, (n, ms, units) in kern.items():
        entry = dict(launches=n, avg_ms=round(ms, 4), samples_per_launch=units)
        if name.startswith("mlp_fwd") or name in ("mlp_bwd_dx", "mlp_bwd_dw"):
            entry["tflops"] = units * FLOP_FWD / (ms * 1e-3) / 1e12          # dX chain and dW (incl. its reduce kernels): the forward's MAC count each
        elif name == "composite_fwd":
            entry["gbps"] = units * 24.2 / (ms * 1e-3) / 1e9                  # 20 B/sample in + 4 B/sample prob + 32 B/ray
        elif name == "composite_train":
            entry["gbps"] = units * 40.3 / (ms * 1e-3) / 1e9                  # one launch: 20 B in, 4 B prob + 16 B gradients out per sample, 68 B per ray
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
        for fname in ("r6_traffic.json", "r5_traffic.json", "r4_traffic.json", "r3_traffic.json", "r2_traffic.json", "r1_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", fname)) as f:
                    t = json.load(f)["bytes_per_sample"].get(dom)
                if t:
                    traffic = dict(value=round((t["fetch_corrected"] + t["write"]) * kernels[dom]["samples_per_launch"] / 1e9, 3), unit="GB",
                                   source=f"profiles/{fname} (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes)")
                    break
            except (OSError, KeyError, ValueError):
                pass
        # `traffic` = HBM bytes per launch (a number, or null); where it comes from sits beside it
        roofline = dict(bound="mfma", kernel=rocprof_name[dom], achieved=a, peak=peak_mfma, unit="TFLOP/s", frac=round(a / peak_mfma, 4),
                        traffic=None if traffic is None else round(traffic["value"] * 1e9), traffic_unit="bytes per launch (HBM, PMC counters)",
                        traffic_source=None if traffic is None else traffic["source"],
                        avg_ms=kernels[dom]["avg_ms"], samples_per_launch=kernels[dom]["samples_per_launch"], flop_per_sample=FLOP_FWD,
                        rocprof=rocprof_row(args.config, rocprof_name[dom]))
        if not exact:
            # The opt-in modes run 3 or 1 bf16 MFMA terms per algorithmic MAC at 16x the fp32 matrix rate, while their kernels still
            # stream the saved activations / gradients: BOTH ceilings are priced and the binding one is named (round-3 review: an MFMA
            # roofline alone named the wrong bound).  Algorithmic bytes per sample: tools/mlp_traffic.py (fp32 workspaces in bf16x3,
            # bf16 quad rows in bf16); counter traffic of these kernels: profiles/r4_fast_<precision>_traffic.json.
            terms = 3 if args.precision == "bf16x3" else 1
            algo = {"fp32w": {"mlp_fwd_train": 16.0 + 9384.0, "mlp_fwd": 32.0, "mlp_bwd_dx": 1092.0 + 9168.0},
                    "bf16w": {"mlp_fwd_train": 16.0 + 4840.0, "mlp_fwd": 32.0, "mlp_bwd_dx": 900.0 + 4792.0}}["fp32w" if terms == 3 else "bf16w"][dom]
            ms, n = kernels[dom]["avg_ms"], kernels[dom]["samples_per_launch"]
            gbps = algo * n / (ms * 1e-3) / 1e9
            issued_tf = a * terms                                     # bf16 MFMA work actually issued
            t_hbm, t_mfma = algo * n / (PEAK_HBM * 1e9), terms * FLOP_FWD * n / (PEAK_BF16_MFMA * 1e12)
            counter = None
            try:
                fast = next(p_ for p_ in (os.path.join(ROOT, "profiles", f"{r_}_fast_{args.precision}_traffic.json") for r_ in ("r6", "r5", "r4")) if os.path.exists(p_))
                with open(fast) as f:
                    t = json.load(f)["bytes_per_sample"].get(dom)
                if t:
                    counter = dict(value=round((t["fetch_corrected"] + t["write"]) * n / 1e9, 3), unit="GB",
                                   source=f"profiles/{os.path.basename(fast)} (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes)")
            except (OSError, KeyError, ValueError, StopIteration):
                pass
            roofline = dict(bound="hbm" if t_hbm >= t_mfma else "mfma", kernel=rocprof_name[dom],
                            achieved=round(gbps, 1) if t_hbm >= t_mfma else round(issued_tf, 2), peak=PEAK_HBM if t_hbm >= t_mfma else PEAK_BF16_MFMA,
                            unit="GB/s" if t_hbm >= t_mfma else "TFLOP/s",
                            frac=round(max(t_hbm, t_mfma) / (ms * 1e-3), 4), traffic=None if counter is None else round(counter["value"] * 1e9),
                            traffic_unit="bytes per launch (HBM, PMC counters)", traffic_source=None if counter is None else counter["source"],
                            avg_ms=ms, samples_per_launch=n,
                            ceilings=dict(hbm=dict(algorithmic_bytes_per_sample=algo, achieved_gbps=round(gbps, 1), peak_gbps=PEAK_HBM, frac=round(gbps / PEAK_HBM, 4),
                                                   floor_ms=round(t_hbm * 1e3, 4)),
                                          mfma=dict(terms_per_mac=terms, issued_tflops=round(issued_tf, 2), peak_tflops=PEAK_BF16_MFMA,
                                                    frac=round(issued_tf / PEAK_BF16_MFMA, 4), floor_ms=round(t_mfma * 1e3, 4))),
                            note="frac = the higher of the two floors / measured time; neither ceiling is close: the kernels are bound by instruction issue "
                                 "(~5 non-MFMA instructions per 32-cycle MFMA of an in-order wave) and the chip clocks 1.8-2.0 GHz under bf16 MFMA",
                            rocprof=rocprof_row(args.config + "_" + args.precision, rocprof_name[dom]))
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
               ms_per_step=ms_step, higher_is_better=True, scaling=scaling, vs_baseline=None, dtype="f32" if exact else args.precision, data=data_label,
               config=dict(workload=desc, name=args.config, rays_per_gpu=rays_per_gpu,
                           samples_per_ray="+".join(str(x) for x in ((S0, S0 + Sf0) if Sf0 else (S0,))),
                           mlp_evals_per_step_per_gpu=evals_local, parallelism=par,
                           precision="exact fp32 MFMA" if exact else
                           f"{args.precision}: OPT-IN fast mode, split-bf16 operands on v_mfma_f32_32x32x16_bf16 with fp32 accumulation ("
                           + ("hi*hi + hi*mid + mid*hi, 16 significand bits per operand" if args.precision == "bf16x3" else "leading plane only, 8 bits") +
                           "); not the headline, own parity row in tests/test_gpu_fast_precision.py"),
               frac_of_train_roofline=round(value / world * 3 * FLOP_FWD / 1e12 / peak_mfma, 4),
               loss=loss_value, hip_graph=graphed, ranks_seen=ranks_seen, backend=dist_backend, warmup_run=n_warm,
               ms_per_step_fastest_rank=dt_min / args.steps * 1e3, comm_ms=None if comm_ms is None else round(comm_ms, 4),
               comm_exposed_ms=None if comm_exposed_ms is None else round(comm_exposed_ms, 4), strong=strong, placement=placement if args.config == "cfg4" else None,
               replicas=replicas, capture_fallback=bool(os.environ.get("NIW_CAPTURE_FALLBACK")),
               comm_bucket_bytes=bucket_bytes if dist_backend else None,
               comm_wire_ms=round(wire_ms(bucket_bytes, world), 4) if dist_backend else None,
               comm_wire_model=f"ring all-reduce: 2 (N - 1) / N x bucket bytes over one xGMI link per neighbour at {XGMI_LINK_GBPS:.0f} GB/s; comm_ms well above it "
                               "= a latency-bound exchange (2 (N - 1) hops), not a bandwidth-bound one" if dist_backend else None,
               split_exchange=split_ab, hip_graph_ab=graph_ab, peak_memory_gb=round(peak_mem, 2), param_checksum=param_checksum,
               launches_per_step=("one niw_train_step call (vanilla model: front, noise draws, ray generation, two field passes with compositing + loss "
                                  "+ backward, closing kernel) + one Adam launch" if args.config == "cfg1" else
                                  "one niw_train_step call (22 kernel launches for a single-pass config, 29 with the fine pass) + gradient exchange + one Adam launch")
               if getattr(loads[0][0], "fused", None) is not None else "autograd mirror over the per-stage entry points",
               roofline=roofline, kernel_check=kernel_check, kernels=kernels)
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
    if world == 1 and not args.no_psnr_parity and args.config != "cfg1":
        # the "+ PSNR parity" half of the metric, bounded (~20 s): the teacher-student scene the trajectory tests use, alignment term on, 120
        # chained iterations on the HIP engine and on the oracle, with the spread between draws as the yardstick; null when the run did not train
        from oracle import parity
        out["psnr_parity"] = parity.trajectory_summary(dev, steps=120, precision=args.precision)
    ga = {"cfg3": 4, "cfg5": 3}.get(args.config, 4 if args.config.startswith("cfg4") else None)
    if world == 1 and not args.no_torch_baseline and not args.shard_of and args.config != "cfg1":
        rng_kw = dict(depth_range=(1.2, 5.2), param="metric") if args.config == "cfg5" else {}
        try:
            out["torch_rocm_baseline"] = torch_rocm_baseline(dev, loads[0][2], int(round(loads[0][3])), S, Sf, opt.H, opt.W, ga_weight=ga, **rng_kw)
            out["torch_rocm_baseline"]["speedup_of_this_build"] = round(value / out["torch_rocm_baseline"]["value"], 2)
        except torch.cuda.OutOfMemoryError as e:          # a reported side figure must not cost the line
            out["torch_rocm_baseline"] = dict(error=f"out of memory: {e}"[:200])
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(loads[0][2], S, Sf, opt.H, opt.W, ga_weight=ga, vanilla=args.config == "cfg1")
    print(json.dumps(out), flush=True)
    finish()


if __name__ == "__main__":
    main()
