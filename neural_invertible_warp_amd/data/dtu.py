"""DTU scans in the pixelNeRF / DVR packaging, producing what the reference's `data/dtu.py` `Dataset` hands to
the engine: per view `idx, image [3,H,W], intr [3,3], pose [3,4]` (world-to-camera, OpenCV axes), `depth_range [2]`, and for
evaluation `depth_gt [H,W], valid_depth_gt [H,W], fg_mask [1,H,W]`, collated in `.all`.  Written from the file formats:

  <root>/rs_dtu_4/DTU/<scan>/image/NNNNNN.png    49 rectified views at 300 x 400; the file stem is the view number
  <root>/rs_dtu_4/DTU/<scan>/cameras.npz         per view `world_mat_N` = the full projection P = K [R|t] (4x4, top 3 rows used) and
                                                 `scale_mat_N` = the normalisation (diag 300, translation) of the DVR convention
  <root>/Depths/<scan>/depth_map_NNNN.pfm        ground-truth depth (PFM, millimetres before scaling)
  <root>/submission_data/idrmasks/<scan>/[mask/]NNN.png   foreground masks (the five IDR scans keep them in a `mask/` sub-folder)

Cameras (reference data/dtu.py:190-255).  P[:, :3] = K R factors by an RQ decomposition into an upper-triangular K with positive
diagonal (normalised to K[2,2] = 1) and a rotation R (world-to-camera); the camera centre is C = -(K R)^-1 P[:, 3].  The scene is
then normalised: C <- (C - scale_mat[:3, 3]) / 300, and the depth maps are divided by the same 300 (the reference hard-codes the
factor and asserts that every scale matrix carries it; so does this loader).  Poses are stored camera-to-world and handed out
world-to-camera.  Depth sampling range: [1.2, 5.2] in the normalised units, optionally widened by
`data.dtu.increase_depth_range_by_x_percent` (:110-111, :362-364).

Splits (:117-150): `pixelnerf` trains on views [25, 22, 28, 40, 44, 48, 0, 8, 13] (the first `train_sub` of them: 1, 3 or 9) and tests
on the views that are neither training nor in the excluded set; `pixelnerf_reduced_testset` and `all` as in the reference; otherwise
every `dtuhold`-th view is a test view.

Images stay at their native size unless `data.dtu.resize` / `resize_factor` ask otherwise (bilinear; intrinsics follow); depth maps
and masks are brought to the image size by nearest-neighbour / bilinear-then-floor resampling.  cv2 and imageio, which the reference
uses for the decomposition, the resampling and PNG reading, are replaced by numpy / scipy / PIL.
"""
import os
import re

import numpy as np
import torch

from ..util import edict

TRAIN_VIEWS_PIXELNERF = [25, 22, 28, 40, 44, 48, 0, 8, 13]
EXCLUDED_VIEWS_PIXELNERF = [3, 4, 5, 6, 7, 16, 17, 18, 19, 20, 21, 36, 37, 38, 39]
TRAIN_VIEWS_REDUCED = TRAIN_VIEWS_PIXELNERF + [24, 30, 41, 47, 43, 29, 45, 34, 33]
TEST_VIEWS_REDUCED = [1, 2, 9, 10, 11, 12, 14, 15, 23, 26, 27, 31, 32, 35, 42, 46]
IDR_SCANS = ("scan40", "scan55", "scan63", "scan110", "scan114")
WORLD_SCALE = 300.0


def read_pfm(path):
    """Portable float map -> (array [H,W] or [H,W,3] float32, top row first; scale).  Header: `Pf` (grey) or `PF` (colour), then
    `width height`, then a scale whose SIGN gives the byte order (negative = little endian); rows are stored bottom-up."""
    with open(path, "rb") as f:
        kind = f.readline().decode("ascii").strip()
        if kind not in ("PF", "Pf"):
            raise ValueError("{}: not a PFM file (header {!r})".format(path, kind))
        dims = re.fullmatch(r"(\d+)\s+(\d+)", f.readline().decode("ascii").strip())
        if not dims:
            raise ValueError("{}: malformed PFM dimensions".format(path))
        width, height = int(dims.group(1)), int(dims.group(2))
        scale = float(f.readline().decode("ascii").strip())
        data = np.fromfile(f, dtype="<f4" if scale < 0 else ">f4")
    channels = 3 if kind == "PF" else 1
    if data.size != width * height * channels:
        raise ValueError("{}: {} samples for a {}x{}x{} map".format(path, data.size, height, width, channels))
    data = data.reshape((height, width, 3) if channels == 3 else (height, width))
    return np.flipud(data).astype(np.float32), abs(scale)


def decompose_projection(P):
    """P [3,4] = K [R | -R C]  ->  (K [3,3] upper triangular, positive diagonal, K[2,2] = 1; R [3,3] world-to-camera rotation;
    C [3] camera centre).  RQ decomposition of the left 3x3 block with the signs fixed so that K's diagonal is positive (what
    cv2.decomposeProjectionMatrix returns)."""
    import scipy.linalg
    M = np.asarray(P, dtype=np.float64)[:3, :3]
    K, R = scipy.linalg.rq(M)
    signs = np.sign(np.diag(K))
    signs[signs == 0] = 1.0
    K, R = K * signs[None, :], R * signs[:, None]          # K D and D R with D = diag(signs), D D = I
    center = -np.linalg.solve(M, np.asarray(P, dtype=np.float64)[:3, 3])
    return K / K[2, 2], R, center


def resample(array, shape, mode):
    """[h,w(,c)] -> [H,W(,c)]: `nearest` takes source pixel floor(i * h / H) (OpenCV's INTER_NEAREST rule); `linear` interpolates
    between pixel centres with edge clamping (INTER_LINEAR).  Same-size requests return the input unchanged."""
    h, w = array.shape[:2]
    H, W = int(shape[0]), int(shape[1])
    if (h, w) == (H, W):
        return array
    if mode == "nearest":
        rows = np.minimum((np.arange(H) * (h / H)).astype(np.int64), h - 1)
        cols = np.minimum((np.arange(W) * (w / W)).astype(np.int64), w - 1)
        return array[rows][:, cols]
    y = np.clip((np.arange(H) + 0.5) * (h / H) - 0.5, 0, h - 1)
    x = np.clip((np.arange(W) + 0.5) * (w / W) - 0.5, 0, w - 1)
    y0, x0 = np.floor(y).astype(np.int64), np.floor(x).astype(np.int64)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    fy, fx = (y - y0).reshape(-1, *([1] * (array.ndim - 1))), (x - x0).reshape(1, -1, *([1] * (array.ndim - 2)))
    a = array.astype(np.float64)
    top = a[y0][:, x0] * (1 - fx) + a[y0][:, x1] * fx
    bottom = a[y1][:, x0] * (1 - fx) + a[y1][:, x1] * fx
    return (top * (1 - fy) + bottom * fy).astype(np.float32)


def split_views(opt, n_views):
    """-> (train view numbers, test view numbers) before `train_sub` / `val_sub`"""
    kind = opt.data.dtu.get("split_type")
    everything = list(range(n_views))
    if kind == "pixelnerf":
        train = list(TRAIN_VIEWS_PIXELNERF)
        return train, [v for v in range(49) if v not in train and v not in EXCLUDED_VIEWS_PIXELNERF]
    if kind == "pixelnerf_reduced_testset":
        return list(TRAIN_VIEWS_REDUCED), list(TEST_VIEWS_REDUCED)
    if kind == "all":
        return everything, everything
    hold = int(opt.data.dtu.get("dtuhold") or 8)
    return [v for v in everything if v % hold != 0], [v for v in everything if v % hold == 0]


class Dataset(torch.utils.data.Dataset):
    raw_H, raw_W = 300, 400
    near_depth, far_depth = 1.2, 5.2

    def __init__(self, opt, split="train", subset=None):
        super().__init__()
        self.opt, self.split = opt, split
        if not opt.H or not opt.W:
            opt.H, opt.W = self.raw_H, self.raw_W
        self.root = opt.data.get("root") or "data/dtu"
        self.scene = opt.data.scene
        self.scaling_factor = 1.0 / WORLD_SCALE
        scan_dir = os.path.join(self.root, "rs_dtu_4", "DTU", self.scene)
        image_dir = os.path.join(scan_dir, "image")
        if not os.path.isdir(image_dir):
            raise FileNotFoundError(image_dir)
        files = sorted(os.listdir(image_dir))
        view_numbers = [int(os.path.splitext(f)[0]) for f in files]
        cameras = np.load(os.path.join(scan_dir, "cameras.npz"))
        self.intrinsics, self.all_poses_c2w = self._read_cameras(cameras, view_numbers)

        train, test = split_views(opt, len(files))
        if opt.get("pose", {}).get("dtu_reconstruction"):
            train = list(range(len(files)))
        dtu = opt.data.dtu
        train = train[:dtu.get("train_sub") or None]
        test = test[:dtu.get("val_sub") or None]
        chosen = train if "train" in split else test
        if subset:
            chosen = chosen[:subset]
        # positions in the sorted file list (== view numbers for complete scans)
        self.render_img_id = np.asarray(chosen, dtype=np.int64)
        self.render_rgb_files = [os.path.join(image_dir, files[i]) for i in chosen]
        self.render_poses_c2w = self.all_poses_c2w[chosen]
        self.render_intrinsics = self.intrinsics[chosen]
        mask_dir = os.path.join(self.root, "submission_data", "idrmasks", self.scene, *(["mask"] if self.scene in IDR_SCANS else []))
        self.render_masks_files = [os.path.join(mask_dir, "{:03d}.png".format(i)) for i in chosen]
        self.depth_dir = os.path.join(self.root, "Depths")
        self.list = list(self.render_rgb_files)

    def _read_cameras(self, cameras, view_numbers):
        intr, c2w = [], []
        for v in view_numbers:
            K, R, center = decompose_projection(cameras["world_mat_{}".format(v)][:3])
            key = "scale_mat_{}".format(v)
            if key in cameras:
                S = np.asarray(cameras[key], dtype=np.float64)
                if not np.allclose(np.diagonal(S[:3, :3]), WORLD_SCALE):
                    raise ValueError("{}: scale matrix {} instead of {} -- depth maps and poses would disagree".format(
                        key, np.diagonal(S[:3, :3]), WORLD_SCALE))
                center = center - S[:3, 3]
            pose = np.eye(4)
            pose[:3, :3], pose[:3, 3] = R.T, center * self.scaling_factor
            k4 = np.eye(4)
            k4[:3, :3] = K
            intr.append(k4)
            c2w.append(pose.astype(np.float32))
        return np.stack(intr), np.stack(c2w)

    def __len__(self):
        return len(self.render_rgb_files)

    def get_all_camera_poses(self, opt):
        """ground-truth world-to-camera poses of the split [N,3,4]"""
        return torch.from_numpy(np.linalg.inv(self.render_poses_c2w.astype(np.float64))[:, :3]).float()

    def read_depth(self, path):
        return read_pfm(path)[0] * np.float32(self.scaling_factor)

    def _resize_plan(self, h, w):
        dtu = self.opt.data.dtu
        if dtu.get("crop") or dtu.get("crop_ratio"):
            # empty in every reference yaml; the reference loader honours them when set (data/dtu.py:407-419) -- refuse rather than
            # silently train on un-cropped images with un-cropped intrinsics
            raise NotImplementedError("data.dtu.crop / data.dtu.crop_ratio are not implemented by this loader (unset in all reference configs)")
        size, factor = dtu.get("resize"), dtu.get("resize_factor")
        if factor:
            H, W = int(round(h * factor)), int(round(w * factor))
        elif isinstance(size, int):
            edge = max(h, w) if dtu.get("resize_by", "max") == "max" else min(h, w)
            H, W = int(round(h * size / edge)), int(round(w * size / edge))
        elif size:
            H, W = int(size[0]), int(size[1])
        else:
            return h, w
        return H + H % 2, W + W % 2                                     # even sizes, as the reference enforces

    def __getitem__(self, idx):
        import PIL.Image
        opt = self.opt
        with PIL.Image.open(self.render_rgb_files[idx]) as im:
            rgb = np.asarray(im.convert("RGB"), dtype=np.float32)
        h, w = rgb.shape[:2]
        view = int(self.render_img_id[idx])
        mask = np.ones((h, w), dtype=bool)
        if os.path.exists(self.render_masks_files[idx]):
            with PIL.Image.open(self.render_masks_files[idx]) as im:
                mask = np.asarray(im.convert("RGB"), dtype=np.float32)[:, :, 0] / 255.0 == 1
        depth_file = os.path.join(self.depth_dir, self.scene, "depth_map_{:04d}.pfm".format(view))
        depth = self.read_depth(depth_file) if os.path.exists(depth_file) else np.zeros((h, w), dtype=np.float32)

        intr = self.render_intrinsics[idx].copy()
        H, W = self._resize_plan(h, w)
        if (H, W) != (h, w):
            rgb = resample(rgb, (H, W), "linear")
            intr[0] *= W / w
            intr[1] *= H / h
        depth = resample(depth, (H, W), "nearest")
        mask = np.floor(resample(mask.astype(np.float32), (H, W), "linear")).astype(bool)
        image = torch.from_numpy(np.ascontiguousarray(rgb.transpose(2, 0, 1)) / np.float32(255.0))
        valid = depth > 0
        if opt.data.dtu.get("mask_img"):
            m = torch.from_numpy(mask)[None].float()
            image = image * m + 1 - m                                   # white, not black, background
            valid = valid & mask
        widen = float(opt.data.dtu.get("increase_depth_range_by_x_percent") or 0.0)
        w2c = np.linalg.inv(self.render_poses_c2w[idx].astype(np.float64))[:3]
        return dict(idx=idx, rgb_path=self.render_rgb_files[idx], image=image, intr=intr[:3, :3].astype(np.float32),
                    pose=w2c.astype(np.float32), depth_gt=depth, valid_depth_gt=valid, fg_mask=mask[None],
                    depth_range=torch.tensor([self.near_depth * (1 - widen), self.far_depth * (1 + widen)], dtype=torch.float32),
                    scene=self.scene)

    def prefetch_all_data(self, opt):
        """every view of the split, stacked (the engine moves the tensors of `.all` to the device once)"""
        views = [self[i] for i in range(len(self))]
        out = edict()
        for key in views[0]:
            vals = [v[key] for v in views]
            out[key] = vals if isinstance(vals[0], str) else torch.stack([torch.as_tensor(x) for x in vals])
        self.all = out
        return out
