"""LLFF dataset parser against what the reference's data/llff.py returns on the same
procedurally written scene (tests/golden/make_golden_data.py -> llff_dataset.npz).  Host-side."""
import importlib.util
import os

import numpy as np
import torch

from neural_invertible_warp_amd.data import llff, synthetic
from neural_invertible_warp_amd.util import edict

HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, "golden", "llff_dataset.npz"))


def _scene_writer():
    spec = importlib.util.spec_from_file_location("make_golden_data_scene", os.path.join(HERE, "golden", "make_golden_data.py"))
    src = open(spec.origin).read()
    ns = {"__name__": "scene_only", "__file__": spec.origin}
    # only the scene writer is needed (the module's top imports the reference-side helpers)
    start, end = src.index("def write_scene"), src.index("def main")
    exec("import os\nimport numpy as np\nN, FH, FW, H, W = 7, 40, 56, 30, 40\n" + src[start:end], ns)
    return ns["write_scene"]


def test_llff_parser_matches_reference(tmp_path):
    pb = _scene_writer()(str(tmp_path))
    assert np.array_equal(pb, G["poses_bounds"])                     # the identical scene was recreated
    opt = edict(H=30, W=40, data=edict(root=str(tmp_path), scene="fern", image_size=[30, 40], center_crop=None, val_ratio=0.3))
    for split in ("train", "val"):
        ds = llff.Dataset(opt, split=split)
        allv = ds.prefetch_all_data(opt)
        np.testing.assert_allclose(allv.pose.numpy(), G[f"{split}_pose"], atol=1e-6)
        np.testing.assert_allclose(allv.intr.numpy(), G[f"{split}_intr"], rtol=1e-6)
        np.testing.assert_allclose(ds.get_all_camera_poses(opt).numpy(), G[f"{split}_all_poses"], atol=1e-6)
        np.testing.assert_allclose(torch.stack([t[2] for t in ds.list]).numpy(), G[f"{split}_bounds"], rtol=1e-6)
        assert np.array_equal(allv.idx.numpy(), G[f"{split}_idx"])
        np.testing.assert_allclose(allv.image.numpy(), G[f"{split}_image"], atol=1e-6)


def test_synthetic_dataset_interface():
    opt = edict(H=12, W=16, data=edict(scene="synthetic"))
    ds = synthetic.Dataset(opt, split="train", n_views=5)
    allv = ds.prefetch_all_data(opt)
    assert allv.image.shape == (5, 3, 12, 16) and allv.intr.shape == (5, 3, 3) and allv.pose.shape == (5, 3, 4)
    assert ds.get_all_camera_poses(opt).shape == (5, 3, 4) and len(ds) == 5
    assert ds[2]["image"].shape == (3, 12, 16)
