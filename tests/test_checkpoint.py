"""Checkpoint wire format against the description of checkpoints written by the
reference's own util.save_checkpoint (tests/golden/make_golden_ckpt.py -> ckpt_structure.json): same top-level
keys, same state-dict keys / shapes / dtypes, weights regenerated from the same seeds hash identically per key,
optimizer and scheduler dicts have the reference's layout and values, and a reference-layout checkpoint
restores into the trainer.  Host-side: the trainer is built on the CPU device and never runs a kernel here."""
import json
import os

import pytest
import torch

from neural_invertible_warp_amd import checkpoint, configs, engine
from oracle import niw_oracle as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ckpt_structure.json")))


def _load(module, params):
    sd = module.state_dict()
    with torch.no_grad():
        for k, v in params.items():
            sd[k].copy_(v)


def make_trainer(device="cpu"):
    B, s = GOLD["B"], GOLD["seeds"]
    opt = configs.cfg2_nerf_inn_llff_hier(device=device)
    opt.barf_c2f = [0.1, 0.5]
    opt.max_iter = GOLD["optim"]["max_iter"]
    tr = engine.INNTrainer(opt, B)
    g = tr.graph
    _load(g.nerf, O.make_nerf_params(seed=s["coarse"]))
    _load(g.nerf_fine, O.make_nerf_params(seed=s["fine"]))
    _load(g.warp_mlp, O.make_warp_params(seed=s["warp"], perturb=0.02))
    with torch.no_grad():
        g.warp_latent.weight.copy_(O.make_latent(s["latent"], B))
    return opt, tr


def test_saved_checkpoint_has_the_reference_layout(tmp_path):
    opt, tr = make_trainer()
    opt.output_path = str(tmp_path)
    checkpoint.save_checkpoint(opt, tr, ep=None, it=0, latest=True)
    ck = torch.load(tmp_path / "model.ckpt", weights_only=False)
    ref = GOLD["ckpt_iter0"]
    assert set(ck.keys()) == set(ref["keys"])
    assert set(ck["graph"].keys()) == set(ref["graph"].keys())
    assert list(ck["graph"].keys()) == GOLD["ckpt_iter1"]["graph_keys"]                # same order too
    for k, want in ref["graph"].items():
        got = ck["graph"][k]
        assert list(got.shape) == want["shape"] and str(got.dtype) == "torch." + want["dtype"], k
        f = got.double().reshape(-1)
        assert abs(float(f.sum()) - want["sum"]) <= 1e-9 * max(1.0, abs(want["sum"])), k    # identical seeded weights
        assert abs(float((f * f).sum()) - want["sumsq"]) <= 1e-9 * max(1.0, want["sumsq"]), k


def _fake_moments(tr, it):
    gen = torch.Generator().manual_seed(3)
    tr.it = it
    for m, v in zip(tr.m, tr.v):
        m.copy_(torch.randn(m.shape, generator=gen) * 1e-3)
        v.copy_(torch.rand(v.shape, generator=gen) * 1e-6)


def test_optimizer_and_scheduler_state_match_the_reference_format():
    opt, tr = make_trainer()
    _fake_moments(tr, 1)
    sd = checkpoint.optimizer_state_dicts(tr)
    ref = GOLD["ckpt_iter1"]
    for name in ("optim", "optim_pose"):
        got, want = sd[name], ref[name]
        assert len(got["param_groups"]) == len(want["param_groups"])
        for gg, wg in zip(got["param_groups"], want["param_groups"]):
            assert gg["params"] == wg["params"]
            assert set(gg.keys()) == set(wg.keys())
            for k in ("lr", "initial_lr", "eps", "weight_decay"):
                assert gg[k] == pytest.approx(wg[k], rel=1e-12), (name, k)
            assert list(gg["betas"]) == list(wg["betas"])
        assert {str(k) for k in got["state"]} == set(want["state"].keys())              # e.g. no entry for nerf.progress
        for k, st in got["state"].items():
            w = want["state"][str(k)]
            assert set(st.keys()) == set(w.keys())
            assert float(st["step"]) == w["step"]["tensor"]["sum"]
            for kk in ("exp_avg", "exp_avg_sq"):
                assert list(st[kk].shape) == w[kk]["tensor"]["shape"]
    for name in ("sched", "sched_pose"):
        got, want = sd[name], ref[name]
        assert set(got.keys()) == set(want.keys())
        for k, v in want.items():
            if isinstance(v, list):
                assert got[k] == pytest.approx(v, rel=1e-12), (name, k)
            else:
                assert got[k] == pytest.approx(v, rel=1e-12) if isinstance(v, float) else got[k] == v, (name, k)


def test_round_trip_through_the_wire_format(tmp_path):
    opt, tr = make_trainer()
    opt.output_path = str(tmp_path)
    _fake_moments(tr, 7)
    tr.graph.nerf.set_progress(7 / opt.max_iter)
    with torch.no_grad():
        tr.graph.global_rigid.weight.add_(0.25)
    checkpoint.save_checkpoint(opt, tr, ep=None, it=7)
    assert os.path.exists(tmp_path / "model" / "7.ckpt")
    opt2, tr2 = make_trainer()
    opt2.output_path = str(tmp_path)
    with torch.no_grad():                                                       # start from different weights
        for p in tr2.graph.parameters():
            p.mul_(0.5)
    ep, it = checkpoint.restore_checkpoint(opt2, tr2, resume=7)
    assert (ep, it) == (None, 7) and tr2.it == 7
    for (k, a), (_, b) in zip(tr.graph.state_dict().items(), tr2.graph.state_dict().items()):
        assert torch.equal(a, b), k
    for i in range(len(tr.m)):
        assert torch.equal(tr.m[i], tr2.m[i]) and torch.equal(tr.v[i], tr2.v[i])
    # the kernels' flat buffers still alias the restored Parameters
    assert tr2.graph.nerf.flat_params.data_ptr() == tr2._flats()[0].data_ptr() == tr2.graph.nerf.mlp_feat[0].weight.data_ptr()
    assert tr2.graph.warp_mlp.flat_params.data_ptr() == tr2._flats()[2].data_ptr()
    assert torch.equal(tr2._flats()[2], tr._flats()[2]) and torch.equal(tr2._flats()[0], tr._flats()[0])
    assert tr2.graph.nerf.progress_host == pytest.approx(7 / opt.max_iter)
    # weights-only load (opt.load) leaves the optimizer state alone
    opt3, tr3 = make_trainer()
    assert checkpoint.restore_checkpoint(opt3, tr3, load_name=str(tmp_path / "model.ckpt")) == (None, None)
    assert tr3.it == 0 and float(tr3.m[0].abs().max()) == 0.0
    assert torch.equal(tr3.graph.global_rigid.weight, tr.graph.global_rigid.weight)
