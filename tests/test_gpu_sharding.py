"""Ray sharding on the HIP path (parallel.py: replicated warp + alignment term, contiguous shares of the B x R rays for everything per
sample): the ranks' losses and gradients must SUM to those of the unsharded step.  The ranks are run one after the other in this
process (no process group: the all-reduce is the sum formed here), each as the engine builds it for (rank, world).  Needs a GPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _step(cfg, rank, world, B, rays, S):
    from neural_invertible_warp_amd import configs, engine
    if cfg == "dtu":
        opt = configs.cfg5_barf_inn_dtu(device=DEV)
        var0, init = engine.synthetic_dtu_scene(opt, B)
    else:
        opt = getattr(configs, cfg)(device=DEV)
        var0, init = engine.synthetic_scene(opt, B), None
    opt.nerf.sample_stratified = False                  # mid-point samples: the shares of a sharded run see the unsharded run's depths
    opt.nerf.rand_rays, opt.nerf.sample_intvs = rays, S
    if opt.nerf.fine_sampling:
        opt.nerf.sample_intvs_fine = S
    tr = engine.INNTrainer(opt, B, rank=rank, world=world, warp_perturb=0.02, seed=4, initial_poses_w2c=init)
    tr.it = 30000                                        # inside the embedder's annealing window: the reference's index quirk is live
    for n in tr.nets:
        n.set_progress(0.3)
    loss = tr._forward_backward(type(var0)(var0), tr.it)
    return {k: float(v.detach()) for k, v in loss.items()}, tr.bucket.flat.clone(), tr


@pytest.mark.parametrize("cfg,B,rays,S,world", [("cfg3_barf_inn_llff", 18, 2048, 16, 8), ("cfg2_nerf_inn_llff_hier", 5, 5 * 37, 16, 3), ("dtu", 3, 3 * 41, 16, 4)])
def test_sharded_ranks_sum_to_the_unsharded_step(cfg, B, rays, S, world):
    ref_loss, ref_grad, ref = _step(cfg, 0, 1, B, rays, S)
    total_loss, total_grad, sizes = None, None, []
    for r in range(world):
        loss, grad, tr = _step(cfg, r, world, B, rays, S)
        total_loss = loss if total_loss is None else {k: total_loss[k] + v for k, v in loss.items()}
        total_grad = grad if total_grad is None else total_grad + grad
        lo, hi = tr.opt.ray_shard and __import__("neural_invertible_warp_amd.parallel", fromlist=["x"]).flat_share(B * (rays // B), r, world)
        sizes.append(hi - lo)
    assert sum(sizes) == B * (rays // B) and max(sizes) - min(sizes) <= 1
    for k in ref_loss:
        assert abs(total_loss[k] - ref_loss[k]) <= 1e-5 * max(abs(ref_loss[k]), 1e-6), (k, total_loss[k], ref_loss[k])
    # every optimizer group: NeRF(s), warp network, latent table.  The embedder's index window is live (alpha = 0.3) and the sums still
    # agree: the warp runs on the whole batch on every rank, so its points keep the indices the reference gives them.
    for i in range(len(ref.bucket.groups)):
        a, b = total_grad[ref.bucket.starts[i]:ref.bucket.starts[i] + ref.bucket.sizes[i]], ref.bucket.segment(i)
        assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()), (cfg, i, float((a - b).abs().max() / b.abs().max()))


def test_photometric_share_kernel_equals_the_slice_of_the_whole_batch():
    """niw_mse_fwd_bwd on the rays [lo, hi) of the flattened view-major list == the corresponding slice of the whole-batch call"""
    from neural_invertible_warp_amd import ops
    B, R, H, W = 5, 37, 12, 16
    gen = torch.Generator(device=DEV).manual_seed(0)
    image = torch.rand(B, 3, H, W, device=DEV, generator=gen)
    rgb = torch.rand(B, R, 3, device=DEV, generator=gen).requires_grad_(True)
    idx = torch.randperm(H * W, device=DEV, generator=gen)[:R]
    n = 3 * B * R
    whole = ops.mse_gather(rgb, image, idx, n)
    whole.backward()
    parts, lo = 0.0, 0
    for hi in (61, 62, 150, B * R):
        piece = rgb.detach().reshape(1, B * R, 3)[:, lo:hi].clone().requires_grad_(True)
        l = ops.mse_gather(piece, image, idx, n, share=(lo, hi))
        l.backward()
        assert torch.allclose(piece.grad[0], rgb.grad.reshape(B * R, 3)[lo:hi], rtol=0, atol=1e-9)
        parts, lo = parts + float(l.detach()), hi
    assert abs(parts - float(whole.detach())) < 1e-6


def test_bench_line_through_a_live_rccl_group_eager_and_captured():
    """bench.py --force-dist: a ONE-rank RCCL process group (everything a one-GPU box can exercise of the N > 1 path on hardware): the
    rank check (an all-reduce of ones through RCCL), the flat gradient all-reduce, and -- with --hip-graph on -- capture and replay of
    the iteration while the communicator is live.  Both runs must print a line with ranks_seen 1, backend nccl and the same loss
    (same number of untimed iterations in front: a launched run takes at least 16, bench.py n_warm)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    losses = {}
    for mode in ("off", "on"):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29631 + (mode == "on")))
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--config", "cfg3", "--lean", "--force-dist", "--hip-graph", mode,
               "--steps", "4", "--warmup", "16", "--kernel-steps", "0"]
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        if mode == "on" and r.returncode == 75:
            # exit code 75 = the capture itself failed (round 4: seen once in ~10 runs beside a live communicator --
            # hipErrorStreamCaptureInvalidated -- which is why launched is bench.py's default under N > 1 and why its launcher parent
            # restarts fresh ranks without the graph).  One fresh process more, as that parent would start it; then the graph must hold.
            env["MASTER_PORT"] = str(29641)
            r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
            if r.returncode == 75:
                pytest.skip("HIP-graph capture beside a live RCCL communicator failed twice in fresh processes (exit code 75)")
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["ranks_seen"] == 1 and line["backend"] == "nccl" and line["hip_graph"] == (mode == "on")
        losses[mode] = line["loss"]
    assert abs(losses["on"] - losses["off"]) <= 1e-6 * abs(losses["off"])


def test_two_ranks_train_validate_and_checkpoint_through_the_command_line(tmp_path):
    """Round-3 advisor finding, on hardware: `torchrun train.py` with TWO ranks (gloo, both on this GPU) through the reference's Model call
    sequence -- validate at iteration 0, sharded train iterations (one-call form, gradient all-reduce), validate and save_checkpoint at
    iteration 2 and 4 -- must finish on both ranks (a collective behind a rank gate would hang it) and leave rank 0's checkpoints, whose
    per-view pose table holds a row from every view's OWNING rank."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--model=barf_inn_llff", "--yaml=barf_inn_llff", "--barf_c2f=[0.1,0.5]", "--loss_weight.global_alignment=2", "--data.dataset=synthetic",
            "--data.image_size=[24,32]", "--nerf.rand_rays=288", "--nerf.sample_intvs=32", "--freq.val=2", "--freq.ckpt=2", "--freq.scalar=1",
            f"--output_root={tmp_path}", "--name=two", "--max_iter=4"]
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", NIW_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, "-m", "neural_invertible_warp_amd.train"] + args, cwd=root, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
        assert "[val it 0]" in out and "[val it 2]" in out and "[val it 4]" in out and "[train it 4]" in out
    # identical validation numbers on both ranks: same parameters after the all-reduced steps, same collected pose table
    val = [[l for l in out.splitlines() if l.startswith("[val it 4]")] for out, _ in outs]
    assert val[0] == val[1], val
    path = [d for d, _, f in os.walk(tmp_path) if "model.ckpt" in f][0]
    assert os.path.exists(os.path.join(path, "model", "2.ckpt")) and os.path.exists(os.path.join(path, "model", "4.ckpt"))
    ck = torch.load(os.path.join(path, "model.ckpt"), weights_only=False)
    table = ck["graph"]["global_rigid.weight"]
    eye = torch.eye(3, 4).reshape(1, 12)
    assert ck["iter"] == 4 and bool(((table.cpu() - eye).abs().amax(dim=1) > 0).all()), "a view's row was never refreshed by its owning rank"


def test_bench_line_under_torchrun_with_two_ranks():
    """The driver's N > 1 invocation (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) with two gloo ranks on this
    one GPU: weak + strong step, the per-kernel table (whose retry must be a collective decision: round 4 found a rank retrying alone and
    its peer in the next collective), parameter checksums equal across ranks, one JSON line from rank 0."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NIW_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29683",
           os.path.join(root, "bench.py"), "--gpus", "2", "--config", "cfg3", "--lean", "--steps", "2", "--warmup", "1", "--kernel-steps", "1"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["backend"] == "gloo" and line["scaling"] == "weak"
    assert line["strong"] and line["strong"]["scaling"] == "strong" and line["comm_ms"] is not None
    assert line["kernel_check"] is not None


def _bench_line(args, env_extra=None, port=29700, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(env_extra or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_split_gradient_exchange_through_rccl_and_between_two_ranks():
    """round 5: the fine network's gradients are all-reduced on a communication stream while the coarse backward still runs (an event
    recorded inside niw_train_step releases them).  (i) one rank through RCCL (--force-dist), launched: the line reports the collectives' own
    time and the exposed part, and trains exactly like the flat exchange; (ii) two gloo ranks on this GPU: same parameters on both ranks
    (bench.py exits non-zero otherwise) and the same loss as the flat exchange (two ranks: a + b either way)."""
    common = ["--config", "cfg2", "--lean", "--steps", "3", "--warmup", "16", "--kernel-steps", "0", "--hip-graph", "off"]
    a = _bench_line(common + ["--force-dist", "--split-exchange", "auto"], port=29711)
    b = _bench_line(common + ["--force-dist", "--split-exchange", "off"], port=29712)
    assert a["backend"] == "nccl" and a["ranks_seen"] == 1 and a["comm_ms"] is not None and a["comm_exposed_ms"] is not None
    assert a["loss"] == b["loss"], (a["loss"], b["loss"])
    small = ["--config", "cfg2", "--lean", "--steps", "2", "--warmup", "16", "--kernel-steps", "0", "--gpus", "2", "--scaling", "strong"]
    c = _bench_line(small + ["--split-exchange", "auto"], env_extra={"NIW_DIST_BACKEND": "gloo"}, port=29713)
    d = _bench_line(small + ["--split-exchange", "off"], env_extra={"NIW_DIST_BACKEND": "gloo"}, port=29714)
    assert c["ranks_seen"] == 2 and c["backend"] == "gloo" and c["comm_exposed_ms"] is not None
    assert c["loss"] == d["loss"], (c["loss"], d["loss"])


def test_cfg4_line_carries_both_placements_with_two_ranks():
    """bench.py --gpus 2 --config cfg4: the ray-sharded placement (weak + strong, gradient all-reduce per scene) and, beside it, the
    scene-replica placement -- scene i whole on rank i mod 2, no exchange, comm_ms 0, every one of the eight scenes listed with its rank
    (SURVEY section 8(e)(3)).  Two gloo ranks on this one GPU; --placement replicas alone gives the replicas as the headline."""
    from neural_invertible_warp_amd import configs
    line = _bench_line(["--gpus", "2", "--config", "cfg4", "--lean", "--steps", "2", "--kernel-steps", "0"], env_extra={"NIW_DIST_BACKEND": "gloo"}, port=29721,
                       timeout=1500)
    assert line["placement"] == "shard" and line["strong"] is not None and line["comm_ms"] is not None
    rep = line["replicas"]
    assert rep["comm_ms"] == 0 and rep["value"] > 0
    assert sorted(r["scene"] for r in rep["scenes"]) == sorted(configs.LLFF_TRAIN_VIEWS)
    assert {r["scene"]: r["rank"] for r in rep["scenes"]} == {sc: i % 2 for i, sc in enumerate(configs.LLFF_TRAIN_VIEWS)}
    assert all(r["views"] == configs.LLFF_TRAIN_VIEWS[r["scene"]] and r["ms_per_step"] > 0 for r in rep["scenes"])
    solo = _bench_line(["--config", "cfg4", "--placement", "replicas", "--shard-of", "8", "--lean", "--steps", "5", "--kernel-steps", "0"], port=29722)
    assert solo["placement"] == "replicas" and [r["scene"] for r in solo["replicas"]["scenes"]] == ["fern"]
    assert solo["config"]["mlp_evals_per_step_per_gpu"] == 18 * 113 * 128 and solo["replicas"]["comm_ms"] == 0
