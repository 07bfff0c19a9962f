"""DTU dataset parser against what the reference's data/dtu.py returns on the same procedurally written
scan (tests/golden/make_golden_dtu_data.py -> dtu_dataset.npz), plus known-answer tests of the two pieces the reference delegates
to cv2, which is absent from the build image (projection-matrix decomposition, resampling).  Host-side."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from neural_invertible_warp_amd.data import dtu
from neural_invertible_warp_amd.util import edict

HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, "golden", "dtu_dataset.npz"))


def _generator():
    src = open(os.path.join(HERE, "golden", "make_golden_dtu_data.py")).read()
    ns = {"__name__": "scene_only"}
    start, end = src.index("def rotation"), src.index("def main")
    exec("import os\nimport numpy as np\nN_VIEWS, H, W = 49, 12, 16\n" + src[start:end], ns)      # the scene writer only
    return ns


@pytest.fixture(scope="module")
def scene(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("dtu"))
    ns = _generator()
    cams = ns["write_scene"](root)
    for k in ("world_mat_0", "scale_mat_0", "world_mat_25"):
        assert np.array_equal(cams[k], G["cam." + k])                              # the identical scene was recreated
    return root, ns["scene_options"]


@pytest.mark.parametrize("tag,over", [("plain", {}), ("masked", dict(mask_img=True, increase_depth_range_by_x_percent=0.1))])
def test_dtu_parser_matches_reference(scene, tag, over):
    root, scene_options = scene
    opt = scene_options(edict, root, **over)
    for split in ("train", "test"):
        ds = dtu.Dataset(opt, split=split)
        allv = ds.prefetch_all_data(opt)
        pre = f"{tag}.{split}."
        assert np.array_equal(np.asarray(ds.render_img_id), G[pre + "view_numbers"])
        np.testing.assert_allclose(allv.pose.numpy(), G[pre + "pose"], atol=2e-6)
        np.testing.assert_allclose(ds.get_all_camera_poses(opt).numpy(), G[pre + "all_poses"], atol=2e-6)
        np.testing.assert_allclose(allv.intr.numpy(), G[pre + "intr"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(allv.image.numpy(), G[pre + "image"], atol=1e-7)
        np.testing.assert_allclose(allv.depth_gt.numpy(), G[pre + "depth_gt"], rtol=1e-7)
        assert np.array_equal(allv.valid_depth_gt.numpy(), G[pre + "valid_depth_gt"])
        assert np.array_equal(allv.fg_mask.numpy(), G[pre + "fg_mask"])
        np.testing.assert_allclose(allv.depth_range.numpy(), G[pre + "depth_range"], rtol=1e-7)
        assert np.array_equal(allv.idx.numpy(), G[pre + "idx"])
        assert allv.image.shape[1:] == (3, 12, 16) and len(allv.rgb_path) == len(ds)


def test_dtu_splits_match_reference(scene):
    root, scene_options = scene
    opt = scene_options(edict, root, train_sub=None, val_sub=None)
    assert np.array_equal(dtu.Dataset(opt, split="train").render_img_id, G["split.pixelnerf.train"])
    assert np.array_equal(dtu.Dataset(opt, split="test").render_img_id, G["split.pixelnerf.test"])
    opt = scene_options(edict, root, split_type=None, train_sub=None, val_sub=None)
    assert np.array_equal(dtu.Dataset(opt, split="test").render_img_id, G["split.hold8.test"])


def test_projection_decomposition_known_answer():
    rng = np.random.default_rng(0)
    for _ in range(20):
        K = np.array([[300 + 50 * rng.random(), rng.normal(), 200 + rng.normal()], [0, 310 + 50 * rng.random(), 150 + rng.normal()], [0, 0, 1.0]])
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        R = Q if np.linalg.det(Q) > 0 else -Q
        C = rng.normal(size=3) * 3
        P = (K * rng.uniform(0.2, 5.0)) @ np.concatenate([R, -(R @ C)[:, None]], axis=1)
        K2, R2, C2 = dtu.decompose_projection(P)
        np.testing.assert_allclose(K2, K, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(R2, R, atol=1e-9)
        np.testing.assert_allclose(C2, C, atol=1e-9)


def test_pfm_both_byte_orders(tmp_path):
    ns = _generator()
    a = np.arange(12, dtype=np.float32).reshape(3, 4) * 1.5
    for le in (True, False):
        ns["write_pfm"](str(tmp_path / "d.pfm"), a, little_endian=le)
        b, scale = dtu.read_pfm(str(tmp_path / "d.pfm"))
        assert np.array_equal(a, b) and scale == 1.0
    (tmp_path / "bad.pfm").write_bytes(b"P6\n1 1\n255\n\0\0\0")
    with pytest.raises(ValueError):
        dtu.read_pfm(str(tmp_path / "bad.pfm"))


def test_resampling_rules():
    a = np.arange(16, dtype=np.float32).reshape(4, 4)
    assert dtu.resample(a, (4, 4), "nearest") is a
    assert np.array_equal(dtu.resample(a, (2, 2), "nearest"), a[::2, ::2])                    # src = floor(dst * scale)
    assert np.array_equal(dtu.resample(a, (8, 8), "nearest"), np.repeat(np.repeat(a, 2, 0), 2, 1))
    half = dtu.resample(a, (2, 2), "linear")                                                  # centres of 2x2 cells: their means
    np.testing.assert_allclose(half, [[2.5, 4.5], [10.5, 12.5]])
    ramp = np.tile(np.arange(4, dtype=np.float32), (4, 1))
    up = dtu.resample(ramp, (4, 8), "linear")
    assert up.shape == (4, 8) and np.all(np.diff(up, axis=1) >= 0) and up.min() == 0 and up.max() == 3
