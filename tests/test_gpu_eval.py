"""Val / eval path on the GPU: similarity alignment of the optimised poses, test-pose
back-alignment, full-image rendering through render_by_slices on the HIP path, PSNR / SSIM / depth
metrics -- checked end to end against the CPU oracle rendering the same view from the same aligned pose.
Tolerances: image 1e-4 (same amplification as test_render_cfg1_golden), PSNR 0.01 dB, SSIM 1e-4."""
import pytest
import torch

from oracle import niw_oracle as O
from tests.test_gpu_parity import DEV, close, g, load_nerf, mk_opt

pytestmark = pytest.mark.gpu


def _poses(n, gen, scale=0.3):
    from neural_invertible_warp_amd import camera
    return camera.lie.se3_to_SE3(torch.randn(n, 6, generator=gen) * scale)


def _intr(H, W):
    return torch.tensor([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]], dtype=torch.float32)[None]


def test_llff_evaluate_full_matches_oracle_render():
    from neural_invertible_warp_amd import camera, evaluation, metrics
    from neural_invertible_warp_amd.model import barf_inn_llff
    from neural_invertible_warp_amd.util import edict
    H, W, S, n = 20, 24, 32, 6
    opt = mk_opt("cfg3_barf_inn_llff", H=H, W=W, **{"nerf.sample_intvs": S, "nerf.rand_rays": 128, "nerf.sample_stratified": False})
    opt.optim.test_photo = False
    gen = torch.Generator().manual_seed(5)
    graph = barf_inn_llff.Graph(opt).attach_warp(opt, n)
    p = O.make_nerf_params(21)
    load_nerf(graph.nerf, p)
    graph.nerf.set_progress(1.0)                                     # all encoding bands on
    # optimised poses = a similarity transform of the ground truth (+ small noise), as after training
    pose_GT = _poses(n, gen)
    sim_R = camera.lie.so3_to_SO3(torch.tensor([0.2, 0.1, -0.4]))
    c_pred = (camera.cam2world(torch.zeros(1, 1, 3), pose_GT)[:, 0] * 0.7) @ sim_R.t() + torch.tensor([0.3, 0.2, -0.1])
    R_pred = camera.lie.so3_to_SO3(torch.randn(n, 3, generator=gen) * 0.01) @ pose_GT[..., :3] @ sim_R.t()
    pose_pred = camera.pose(R=R_pred, t=(-R_pred @ c_pred[..., None])[..., 0])
    with torch.no_grad():
        graph.global_rigid.weight.copy_(pose_pred.reshape(n, 12))
    test_pose, image = _poses(1, gen), torch.rand(1, 3, H, W, generator=gen)
    ev = evaluation.LLFFEvaluator(opt, graph, g(pose_GT))
    out = ev.evaluate_full(opt, [edict(idx=torch.arange(1), image=g(image), intr=g(_intr(H, W)), pose=g(test_pose))])
    assert out.error.R.mean() < 0.03 and out.error.t.mean() < 0.03
    # oracle: same aligned test pose (host algebra, float32 on both sides), rendered on the CPU
    sim3 = graph.sim3
    pose_al = barf_inn_llff.Graph.get_pose(graph, opt, edict(pose=g(test_pose)), mode="eval").cpu()
    center, ray = O.center_and_ray(H, W, pose_al, _intr(H, W))
    with torch.no_grad():
        ref = O.render_rays(p, center, ray, 0.5, S, opt.nerf.depth.range, opt.nerf.depth.param, density_activ="softplus")
    rgb_ref = ref["rgb"].view(1, H, W, 3).permute(0, 3, 1, 2)
    close(out.maps.rgb, rgb_ref, atol=1e-4, rtol=1e-3)
    assert abs(out.res[0].psnr - metrics.psnr(rgb_ref, image).item()) < 0.01
    assert abs(out.res[0].ssim - metrics.ssim(rgb_ref.contiguous(), image).item()) < 1e-4
    assert sim3.R.shape == (3, 3)


def test_dtu_evaluate_full_matches_oracle_render():
    from neural_invertible_warp_amd import camera, evaluation, metrics
    from neural_invertible_warp_amd.align_trajectories import backtrack_from_aligning_the_trajectory
    from neural_invertible_warp_amd.model import barf_inn_dtu
    from neural_invertible_warp_amd.model.pose_models.inn import INNPoseParams
    from neural_invertible_warp_amd.util import edict
    H, W, S, n = 20, 24, 32, 3
    opt = mk_opt("cfg5_barf_inn_dtu", H=H, W=W, **{"nerf.sample_intvs": S, "nerf.rand_rays": 128, "nerf.sample_stratified": False})
    opt.optim.test_photo = False
    gen = torch.Generator().manual_seed(9)
    pose_GT = _poses(n, gen, 0.4)
    init = camera.pose.compose([_poses(n, gen, 0.05), pose_GT])                       # noisy initial poses
    pose_net = INNPoseParams(opt, num_poses=n, initial_poses_w2c=g(init), device=DEV)
    with torch.no_grad():                                                             # a learnt global correction
        pose_net.pose_global.weight.copy_(g(_poses(n, gen, 0.03).reshape(n, 12)))
    graph = barf_inn_dtu.Graph(opt, pose_net)
    p = O.make_nerf_params(33)
    load_nerf(graph.nerf, p)
    graph.nerf.set_progress(1.0)
    test_pose, image = _poses(1, gen, 0.4), torch.rand(1, 3, H, W, generator=gen)
    depth_gt, valid = torch.rand(1, H, W, generator=gen) * 4 + 1.2, torch.rand(1, H, W, generator=gen) > 0.2
    rng = torch.tensor([[1.2, 5.2]])
    ev = evaluation.DTUEvaluator(opt, graph, g(pose_GT))
    stats = ev.evaluate_poses(opt)
    assert float(stats["error_t"]) <= float(stats["error_t_before_align"]) + 1e-6
    var = edict(idx=torch.arange(1), image=g(image), intr=g(_intr(H, W)), pose=g(test_pose), depth_range=g(rng),
                depth_gt=g(depth_gt), valid_depth_gt=g(valid))
    out = ev.evaluate_full(opt, [var])
    sim = pose_net.sim3_est_to_gt_c2w
    pose_al = backtrack_from_aligning_the_trajectory(g(test_pose), sim).cpu()
    center, ray = O.center_and_ray(H, W, pose_al, _intr(H, W))
    with torch.no_grad():
        ref = O.render_rays(p, center, ray, 0.5, S, [1.2, 5.2], "metric", density_activ="softplus")
    rgb_ref = ref["rgb"].view(1, H, W, 3).permute(0, 3, 1, 2)
    assert abs(out.res[0].psnr - metrics.psnr(rgb_ref, image).item()) < 0.01
    assert abs(out.res[0].ssim - metrics.ssim(rgb_ref.contiguous(), image).item()) < 1e-4
    want = metrics.compute_depth_metrics(edict(depth=ref["depth"], depth_gt=depth_gt, valid_depth_gt=valid), float(sim.s))
    assert abs(out.res[0].abs_err - want[0]) < 1e-3 and abs(out.res[0].rms_err - want[1]) < 1e-3


def test_llff_test_time_pose_refinement_gradient_matches_oracle():
    """First iteration of evaluate_test_time_photometric_optim (barf_inn_llff.py:218-234): d loss / d se3 of the
    test-pose correction through lie.se3_to_SE3 -> pose composition -> rays -> HIP render, vs oracle autograd.
    Tolerance 2e-2 of the gradient's max (positions pass through the 2^9*pi encoding band)."""
    import types
    from neural_invertible_warp_amd import camera, evaluation
    from neural_invertible_warp_amd.model import barf_inn_llff
    from neural_invertible_warp_amd.util import edict
    from tests.test_gpu_parity import _capture_rng
    H, W, S, n, R = 20, 24, 32, 6, 96
    opt = mk_opt("cfg3_barf_inn_llff", H=H, W=W, **{"nerf.sample_intvs": S, "nerf.rand_rays": R, "nerf.sample_stratified": False})
    opt.optim.test_iter = 3
    gen = torch.Generator().manual_seed(11)
    graph = barf_inn_llff.Graph(opt).attach_warp(opt, n)
    p = O.make_nerf_params(4)
    load_nerf(graph.nerf, p)
    graph.nerf.set_progress(1.0)
    pose_GT = _poses(n, gen)
    with torch.no_grad():
        graph.global_rigid.weight.copy_(camera.pose.compose([_poses(n, gen, 0.02), pose_GT]).reshape(n, 12))
    ev = evaluation.LLFFEvaluator(opt, graph, g(pose_GT))
    _, graph.sim3 = ev.prealign_cameras(opt, *ev.get_all_training_poses(opt))
    test_pose, image, intr = _poses(1, gen), torch.rand(1, 3, H, W, generator=gen), _intr(H, W)
    idx = torch.randperm(H * W, generator=gen)[:R]
    w = torch.nn.Parameter(torch.zeros(1, 6, device=DEV))
    var = edict(idx=torch.arange(1), image=g(image), intr=g(intr), pose=g(test_pose), pose_refine_test=camera.lie.se3_to_SE3(w))
    with _capture_rng(None, g(idx)):
        var = graph.forward(opt, var, mode="test-optim")
    loss = graph.compute_loss(opt, var, mode="test-optim")
    loss.render.backward()
    # oracle: identical algebra on the CPU
    copt = edict(device="cpu", optim=edict(test_photo=True))
    sim3 = edict({k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in graph.sim3.items()})
    wc = torch.zeros(1, 6, requires_grad=True)
    pose_c = barf_inn_llff.Graph.get_pose(types.SimpleNamespace(sim3=sim3), copt,
                                          edict(pose=test_pose, pose_refine_test=camera.lie.se3_to_SE3(wc)), mode="test-optim")
    center, ray = O.center_and_ray(H, W, pose_c, intr)
    ref = O.render_rays(p, center[:, idx], ray[:, idx], 0.5, S, opt.nerf.depth.range, opt.nerf.depth.param, density_activ="softplus")
    loss_c = O.mse_loss(ref["rgb"], O.gather_pixels(image, idx))
    loss_c.backward()
    close(loss.render, loss_c, atol=1e-6)
    assert (w.grad.cpu() - wc.grad).abs().max() <= 2e-2 * wc.grad.abs().max(), (w.grad, wc.grad)
    # and the optimisation loop itself runs and moves the correction
    out = ev.evaluate_test_time_photometric_optim(opt, edict(idx=torch.arange(1), image=g(image), intr=g(intr), pose=g(test_pose)))
    assert torch.isfinite(out.se3_refine_test).all() and out.se3_refine_test.abs().max() > 0


def test_checkpoint_resume_continues_bit_exactly(tmp_path):
    """train 2 steps, save in the reference's wire format, restore into a fresh trainer: the third
    step (same RNG state) gives bit-identical losses and weights as continuing without the round trip."""
    from neural_invertible_warp_amd import checkpoint, configs, engine
    from neural_invertible_warp_amd.util import edict
    n = 4

    def fresh():
        opt = configs.cfg3_barf_inn_llff(device=DEV)
        opt.H, opt.W, opt.nerf.sample_intvs, opt.nerf.rand_rays = 24, 32, 32, 64 * n
        opt.loss_weight.global_alignment = 2
        opt.output_path = str(tmp_path)
        tr = engine.INNTrainer(opt, n, warp_perturb=0.02, seed=3)
        return opt, tr, engine.synthetic_scene(opt, n, seed=1)

    opt, tr, var0 = fresh()
    for _ in range(2):
        tr.train_iteration(edict(var0))
    checkpoint.save_checkpoint(opt, tr, ep=None, it=tr.it)
    torch.manual_seed(77)
    loss_a = tr.train_iteration(edict(var0))
    opt2, tr2, _ = fresh()
    assert checkpoint.restore_checkpoint(opt2, tr2, resume=True) == (None, 2)
    torch.manual_seed(77)
    loss_b = tr2.train_iteration(edict(var0))
    for k in loss_a:
        assert torch.equal(torch.as_tensor(loss_a[k]).detach().cpu(), torch.as_tensor(loss_b[k]).detach().cpu()), k
    for (k, a), (_, b) in zip(tr.graph.state_dict().items(), tr2.graph.state_dict().items()):
        assert torch.equal(a, b), k
    for i in range(len(tr.m)):
        assert torch.equal(tr.m[i], tr2.m[i]) and torch.equal(tr.v[i], tr2.v[i])


def test_train_entry_point_runs_resumes_and_evaluates(tmp_path, capsys):
    """The reference's command line / Model call sequence (train.py:9-32) on the procedural scene: trains, validates
    (pose alignment + held-out PSNR), writes model.ckpt in the reference's layout, resumes from it, evaluates."""
    import os
    from neural_invertible_warp_amd import train
    args = ["--model=barf_inn_llff", "--yaml=barf_inn_llff", "--barf_c2f=[0.1,0.5]", "--loss_weight.global_alignment=2",
            "--data.dataset=synthetic", "--data.image_size=[24,32]", "--nerf.rand_rays=288", "--nerf.sample_intvs=32",
            "--freq.val=3", "--freq.ckpt=3", "--freq.scalar=2", f"--output_root={tmp_path}", "--name=t"]
    m = train.main(args + ["--max_iter=6"])
    out = capsys.readouterr().out
    assert "[val it 3]" in out and "[val it 6]" in out and "[train it 2]" in out
    path = m.opt.output_path
    assert os.path.exists(f"{path}/model.ckpt") and os.path.exists(f"{path}/model/3.ckpt") and os.path.exists(f"{path}/model/6.ckpt")
    ck = torch.load(f"{path}/model.ckpt", weights_only=False)
    assert ck["iter"] == 6 and "warp_mlp.lin0_a_0.weight_g" in ck["graph"] and "optim_pose" in ck
    m2 = train.main(args + ["--max_iter=8", "--resume"])
    assert m2.iter_start == 6 and m2.it == 8
    res = m2.evaluate_full(m2.opt)
    assert len(res.res) == len(m2.test_data) and all(r.psnr > 0 for r in res.res)
    assert os.path.exists(f"{path}/quant.txt") and os.path.exists(f"{path}/quant_pose.txt")


def test_train_entry_point_vanilla_nerf(tmp_path, capsys):
    """--model=nerf --yaml=nerf_llff_repr (BASELINE configs[0] shape family: GT poses, coarse + fine networks, density noise): since round 6
    the Model drives engine.NeRFTrainer -- one niw_train_step call + one fused Adam launch per iteration -- and still writes / resumes the
    reference's model.ckpt dict (graph, optim with one param group per network, sched, epoch, iter)"""
    import os
    from neural_invertible_warp_amd import train
    args = ["--model=nerf", "--yaml=nerf_llff_repr", "--data.dataset=synthetic", "--data.image_size=[24,32]", "--nerf.rand_rays=288",
            "--nerf.sample_intvs=16", "--nerf.sample_intvs_fine=16", "--freq.val=4", "--freq.ckpt=4", "--freq.scalar=2",
            f"--output_root={tmp_path}", "--name=v"]
    m = train.main(args + ["--max_iter=4"])
    out = capsys.readouterr().out
    assert m.trainer is not None and m.trainer.fused is not None, getattr(m.trainer, "fused_fallback_reason", "the Model did not choose the engine")
    assert "[val it 0]" in out and "[val it 4]" in out and "render_fine=" in out
    ck = torch.load(f"{m.opt.output_path}/model.ckpt", weights_only=False)
    assert ck["iter"] == 4 and "nerf_fine.mlp_rgb.1.bias" in ck["graph"] and len(ck["optim"]["param_groups"]) == 2
    assert set(ck["optim"]["state"]) and ck["sched"]["last_epoch"] == 4
    m2 = train.main(args + ["--max_iter=6", "--resume"])
    assert m2.iter_start == 4 and m2.it == 6 and os.path.exists(f"{m.opt.output_path}/model/4.ckpt")
    # the resumed run continues the SAME training: its Adam moments came from the checkpoint
    assert float(m2.trainer.m[0].abs().sum()) > 0 and m2.trainer.it == 6


def test_train_entry_point_dtu_with_learnable_poses(tmp_path, capsys):
    """--model=barf_inn_dtu --yaml=barf_inn_dtu (BASELINE cfg 5 family): noisy initial poses, INNPoseParams, two optimizers,
    validation through the pairwise pose alignment + back-aligned test poses, resume.  Since round 6 the Model drives engine.INNTrainer
    (family "dtu": one niw_train_step call + one fused Adam launch) and writes / resumes the reference's model.ckpt dict."""
    from neural_invertible_warp_amd import train
    args = ["--model=barf_inn_dtu", "--yaml=barf_inn_dtu", "--barf_c2f=[0.1,0.5]", "--loss_weight.global_alignment=3", "--data.dataset=dtu", "--data.synthetic_fallback", "--data.image_size=[24,32]", "--nerf.rand_rays=192",
            "--nerf.sample_intvs=32", "--data.train_sub=3", "--freq.val=3", "--freq.ckpt=3", "--freq.scalar=1", "--optim.test_iter=3",
            f"--output_root={tmp_path}", "--name=d"]
    m = train.main(args + ["--max_iter=3"])
    out = capsys.readouterr().out
    assert "[val it 3]" in out and "rot " in out and "global_alignment=" in out
    assert m.trainer is not None and m.trainer.family == "dtu" and m.trainer.fused is not None, "the DTU Model did not choose the engine's one-call iteration"
    ck = torch.load(f"{m.opt.output_path}/model.ckpt", weights_only=False)
    assert "pose_net.pose_embedding.lin0_a_0.weight_g" in ck["graph"] and "optim_pose" in ck and ck["iter"] == 3
    assert hasattr(m.pose_net, "sim3_est_to_gt_c2w")
    assert len(ck["optim_pose"]["param_groups"]) == 2 and ck["sched_pose"]["last_epoch"] == 3
    m2 = train.main(args + ["--max_iter=5", "--resume"])
    assert m2.iter_start == 3 and m2.it == 5 and float(m2.trainer.m[-2].abs().sum()) > 0          # the pose network's Adam moments came back
    res = m2.evaluate_full(m2.opt)
    assert all(r.psnr > 0 for r in res.res)


def test_full_image_render_is_independent_of_the_slice_size():
    """nerf.eval_slice_rays (bigger slices than the reference's rand_rays) must not change a deterministic render"""
    from neural_invertible_warp_amd.model import nerf
    H, W = 20, 24
    opt = mk_opt("cfg1_nerf_llff_repr", H=H, W=W, **{"nerf.sample_intvs": 16, "nerf.sample_intvs_fine": 16, "nerf.rand_rays": 64,
                                                     "nerf.sample_stratified": False, "nerf.density_noise_reg": None})
    graph = nerf.Graph(opt)
    load_nerf(graph.nerf, O.make_nerf_params(3)); load_nerf(graph.nerf_fine, O.make_nerf_params(4))
    pose, intr = g(_poses(1, torch.Generator().manual_seed(2))), g(_intr(H, W))
    with torch.no_grad():
        a = graph.render_by_slices(opt, pose, intr=intr, mode="eval")
        opt.nerf.eval_slice_rays = H * W
        b = graph.render_by_slices(opt, pose, intr=intr, mode="eval")
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_no_grad_renders_use_the_non_saving_kernel():
    """Under torch.no_grad() the MLP must not run in training mode (saving 9.4 KB per sample) just because its
    parameters require grad: needs_input_grad ignores the grad mode."""
    from neural_invertible_warp_amd import ops
    from neural_invertible_warp_amd.model import nerf
    opt = mk_opt("cfg1_nerf_llff_repr", H=16, W=20, **{"nerf.sample_intvs": 16, "nerf.sample_intvs_fine": 16, "nerf.rand_rays": 64})
    graph = nerf.Graph(opt)
    pose, intr = g(_poses(1, torch.Generator().manual_seed(2))), g(_intr(16, 20))
    ops.TIMING.enabled = True
    ops.TIMING.reset()
    try:
        with torch.no_grad():
            graph.render_by_slices(opt, pose, intr=intr, mode="eval")
        torch.cuda.synchronize()
        names = set(ops.TIMING.summary())
        assert names == {"render_fwd"}, names                          # niw_render_fwd calls niw_mlp_fwd with save = NULL
        ops.TIMING.reset()
        graph.render(opt, pose, intr=intr, ray_idx=torch.arange(32, device=DEV), mode="train")
        torch.cuda.synchronize()
        assert "mlp_fwd_train" in set(ops.TIMING.summary())
    finally:
        ops.TIMING.enabled = False
        ops.TIMING.reset()


def _scene_writer(script, header, first, last="def main"):
    import os
    src = open(os.path.join(os.path.dirname(__file__), "golden", script)).read()
    ns = {"__name__": "scene_only"}
    exec("import os\nimport numpy as np\n" + header + "\n" + src[src.index(first):src.index(last)], ns)
    return ns


def test_train_entry_point_on_dataset_files(tmp_path, capsys):
    """The two dataset parsers feed the engines end to end: an LLFF directory (poses_bounds.npy + images/) through
    --model=barf_inn_llff and a DTU scan (cameras.npz, PFM depth, masks) through --model=barf_inn_dtu, both written by the generators
    of the reference-pinned fixtures; a missing dataset is an error, not a silent switch to the procedural scene."""
    from neural_invertible_warp_amd import train
    llff_root, dtu_root = str(tmp_path / "llff"), str(tmp_path / "dtu")
    _scene_writer("make_golden_data.py", "N, FH, FW, H, W = 7, 40, 56, 30, 40", "def write_scene")["write_scene"](llff_root)
    _scene_writer("make_golden_dtu_data.py", "N_VIEWS, H, W = 49, 12, 16", "def rotation")["write_scene"](dtu_root)
    common = ["--nerf.sample_intvs=16", "--freq.val=2", "--freq.ckpt=2", "--freq.scalar=1", "--optim.test_iter=2", f"--output_root={tmp_path}", "--max_iter=2"]
    m = train.main(["--model=barf_inn_llff", "--yaml=barf_inn_llff", "--barf_c2f=[0.1,0.5]", "--loss_weight.global_alignment=4", f"--data.root={llff_root}", "--data.image_size=[30,40]", "--data.val_ratio=0.3",
                    "--nerf.rand_rays=60", "--name=l"] + common)
    assert len(m.train_data) == 5 and len(m.test_data) == 2 and m.train_data.all.image.shape == (5, 3, 30, 40)
    assert "[val it 2]" in capsys.readouterr().out
    m = train.main(["--model=barf_inn_dtu", "--yaml=barf_inn_dtu", "--barf_c2f=[0.1,0.5]", "--loss_weight.global_alignment=3", f"--data.root={dtu_root}", "--data.scene=scan65", "--data.image_size=[12,16]", "--data.dtu.split_type=pixelnerf",
                    "--data.dtu.train_sub=3", "--data.dtu.val_sub=2", "--nerf.rand_rays=48", "--name=d"] + common)
    assert len(m.train_data) == 3 and m.train_data.all.depth_range.shape == (3, 2) and list(m.train_data.render_img_id) == [25, 22, 28]
    assert "[val it 2]" in capsys.readouterr().out
    with pytest.raises(FileNotFoundError, match="synthetic_fallback"):
        train.main(["--model=barf_inn_llff", "--yaml=barf_inn_llff", f"--data.root={tmp_path}/nowhere", "--name=x"] + common)


def test_bench_reads_a_real_llff_directory_when_one_is_there(tmp_path):
    """bench.py --llff-root DIR (or NIW_LLFF_ROOT): the scene found there feeds the trainer through data/llff.py -- the parser pinned to
    the reference's (/root/reference data/llff.py:28-72) -- and the line says `data: llff:fern`; without it the line says `synthetic`.
    The directory is written by the generator of the reference-pinned fixture: 20 frames (18 train views after the 10 % hold-out, like
    the public fern), 60 x 80 files declared 3024 x 4032 in poses_bounds.npy, resized to the bench's 300 x 400."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    llff_root = str(tmp_path / "llff")
    _scene_writer("make_golden_data.py", "N, FH, FW, H, W = 20, 60, 80, 300, 400", "def write_scene")["write_scene"](llff_root)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--config", "cfg3", "--lean", "--steps", "3", "--kernel-steps", "0"]
    r = subprocess.run(cmd + ["--llff-root", llff_root], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["data"] == "llff:fern" and line["config"]["rays_per_gpu"] == 18 * 113 and line["value"] > 0
    r = subprocess.run(cmd, cwd=root, env=dict(env, NIW_LLFF_ROOT=str(tmp_path / "nowhere")), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["data"] == "synthetic"
