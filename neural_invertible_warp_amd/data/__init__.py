"""Datasets feeding the render path: `llff` parses the public LLFF layout (`poses_bounds.npy` + `images/`),
`dtu` the pixelNeRF / DVR packaging of the DTU scans (`cameras.npz`, PFM depth, IDR masks) -- both pinned to what the reference's
own loaders return on procedurally written scenes -- and `synthetic` is the procedural stand-in used where no dataset is mounted
(bench.py, tests, the GPU box)."""
import importlib
import os

import torch

from ..util import edict


def scene_directory(opt):
    """where the files of `opt.data.dataset` / `opt.data.scene` are expected"""
    name = opt.data.dataset
    root = opt.data.get("root") or "data/{}".format(name)
    return os.path.join(root, "rs_dtu_4", "DTU", opt.data.scene) if name == "dtu" else os.path.join(root, opt.data.scene)


def open_splits(opt, eval_split="val"):
    """-> (train dataset, evaluation dataset), every view pre-loaded and its tensors resident on `opt.device` (`.all`), as the
    reference engines do once before training (model/nerf_inn_llff.py:22-32).  A dataset whose files are missing is an ERROR unless
    `data.synthetic_fallback` is set, in which case the procedural scene takes its place -- loudly."""
    name = opt.data.dataset
    if name != "synthetic" and not os.path.isdir(scene_directory(opt)):
        if not opt.data.get("synthetic_fallback"):
            raise FileNotFoundError("dataset '{}': {} not found (set --data.root, or --data.synthetic_fallback to train on the "
                                    "procedural stand-in scene)".format(name, scene_directory(opt)))
        print("[niw] WARNING: {} not found -- training on the PROCEDURAL stand-in scene (data.synthetic_fallback)".format(scene_directory(opt)), flush=True)
        name = "synthetic"
    module = importlib.import_module("{}.{}".format(__name__, name))
    sub = opt.data.get(opt.data.dataset, {}) if isinstance(opt.data.get(opt.data.dataset), dict) else {}
    train = module.Dataset(opt, split="train", subset=opt.data.get("train_sub") or sub.get("train_sub"))
    held_out = module.Dataset(opt, split="test" if opt.data.get("val_on_test") or opt.data.dataset == "dtu" else eval_split,
                              subset=opt.data.get("val_sub") or sub.get("val_sub"))
    for d in (train, held_out):
        d.prefetch_all_data(opt)
        d.all = edict({k: (v.to(opt.device) if isinstance(v, torch.Tensor) else v) for k, v in d.all.items()})
    return train, held_out
