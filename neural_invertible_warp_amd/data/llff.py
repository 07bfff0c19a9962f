"""LLFF forward-facing scenes.  Produces what the reference's `data/llff.py` `Dataset` hands to the
engine -- `idx`, `image [N,3,H,W]`, `intr [N,3,3]`, `pose [N,3,4]` (world-to-camera) stacked in `.all` -- and keeps its public
surface (`Dataset(opt, split, subset)`, `len`, `[i]`, `.list`, `.focal`, `prefetch_all_data`, `get_all_camera_poses`,
`get_camera`, `get_image`, `preprocess_*`), because the engine and the evaluators are written against it.  The camera maths is
restated from the file format rather than from the reference's call sequence, and runs for all views at once:

`poses_bounds.npy` is one row of 17 numbers per image.  The first 15 form a 3x5 matrix `[a0 a1 a2 t | hwf]`: `a0..a2` are the
camera axes "down, right, backwards" in world coordinates, `t` the camera position, and the last column holds image height,
width and focal length in pixels.  The last two numbers are the near / far depth bounds of the view.

  1. axes: right-handed "right, up, backwards" frame  X = [a1, -a0, a2]                              (data/llff.py:49-53)
  2. scale: positions and bounds are divided by 0.75 x the smallest near bound                       (:57-59)
  3. centring: with c = mean position, u = normalised mean up axis, b = normalised mean backward axis and r = u x b, every
     camera is expressed in the frame M = [r u b] (not re-orthogonalised):  X <- M^T X,  t <- M^T (t - c)  (:63-72)
  4. training convention: world-to-camera matrices whose cameras look down +z and whose world is flipped likewise (so that the
     average camera is the identity the optimisation starts from).  With F = diag(1,-1,-1):
     R = F X^T F,  trans = -F X^T t                                                                (:107-121)

Images are centre-cropped (`data.center_crop`), resized to `opt.W x opt.H` and scaled to [0,1]; the pinhole intrinsics
(focal, principal point at the raw image centre) follow the crop and the resize (data/base.py:89-110).  The last
`val_ratio` of the sorted image list is the validation split (:32-33).
"""
import os

import numpy as np
import torch

from ..util import edict

_FLIP = np.diag([1.0, -1.0, -1.0])


def read_poses_bounds(path):
    """-> (axes [N,3,3] right/up/back columns, position [N,3], bounds [N,2], (H, W, focal)) exactly as stored (step 1 only)"""
    raw = np.load(path).astype(np.float64)
    if raw.ndim != 2 or raw.shape[1] != 17:
        raise ValueError("{}: expected [N,17] rows of a 3x5 camera block plus two depth bounds, got {}".format(path, raw.shape))
    block = raw[:, :15].reshape(-1, 3, 5)
    down, right, back = block[:, :, 0], block[:, :, 1], block[:, :, 2]
    axes = np.stack([right, -down, back], axis=2)
    return axes, block[:, :, 3].copy(), raw[:, 15:].copy(), tuple(block[0, :, 4])


def normalise_rig(axes, position, bounds):
    """steps 2 and 3 -> (axes, position, bounds) in the scaled, centred frame"""
    scale = 1.0 / (0.75 * bounds.min())
    position, bounds = position * scale, bounds * scale
    unit = lambda v: v / max(np.linalg.norm(v), 1e-12)
    up, back = unit(axes[:, :, 1].mean(axis=0)), unit(axes[:, :, 2].mean(axis=0))
    frame = np.stack([np.cross(up, back), up, back], axis=1)                 # M = [r u b]
    return frame.T @ axes, (position - position.mean(axis=0)) @ frame, bounds


def world_to_camera(axes, position):
    """step 4 -> [N,3,4]"""
    Rt = np.swapaxes(axes, 1, 2)                                             # X^T
    R = _FLIP @ Rt @ _FLIP
    trans = -(_FLIP @ Rt @ position[:, :, None])
    return np.concatenate([R, trans], axis=2)


class Dataset(torch.utils.data.Dataset):
    raw_H, raw_W = 3024, 4032

    def __init__(self, opt, split="train", subset=None):
        super().__init__()
        self.opt, self.split = opt, split
        ratio = opt.data.get("center_crop")
        self.crop_H = int(self.raw_H * ratio) if ratio is not None else self.raw_H
        self.crop_W = int(self.raw_W * ratio) if ratio is not None else self.raw_W
        if not (opt.H and opt.W):                          # no explicit size: render at the (cropped) file resolution
            opt.H, opt.W = (self.crop_H, self.crop_W)
        self.root = opt.data.get("root") or "data/llff"
        self.path = os.path.join(self.root, opt.data.scene)
        self.path_image = os.path.join(self.path, "images")
        names = sorted(os.listdir(self.path_image))

        axes, position, bounds, (h, w, focal) = read_poses_bounds(os.path.join(self.path, "poses_bounds.npy"))
        if (int(h), int(w)) != (self.raw_H, self.raw_W):
            raise ValueError("{}: images of {}x{}, expected {}x{}".format(self.path, int(h), int(w), self.raw_H, self.raw_W))
        if len(names) != len(axes):
            raise ValueError("{}: {} images but {} camera rows".format(self.path, len(names), len(axes)))
        self.focal = torch.tensor(focal, dtype=torch.float32)
        axes, position, bounds = normalise_rig(axes, position, bounds)
        rig = torch.from_numpy(np.concatenate([axes, position[:, :, None]], axis=2)).float()     # camera-to-world, before step 4
        w2c = torch.from_numpy(world_to_camera(axes, position)).float()
        bounds = torch.from_numpy(bounds).float()

        held_out = int(len(names) * opt.data.val_ratio)
        keep = slice(0, len(names) - held_out) if split == "train" else slice(len(names) - held_out, len(names))
        if held_out == 0 and split != "train":
            keep = slice(0, 0)
        ids = list(range(len(names)))[keep][:subset or None]
        # `.list` keeps the reference's per-view record (file name, centred camera-to-world pose, bounds)
        self.list = [(names[i], rig[i], bounds[i]) for i in ids]
        self._w2c = w2c[ids]

    def __len__(self): return len(self.list)

    # ---- cameras
    def get_all_camera_poses(self, opt):
        """ground-truth world-to-camera poses of the split [N,3,4]"""
        return self._w2c.clone()

    def get_camera(self, opt, idx):
        f = float(self.focal)
        intr = torch.tensor([[f, 0.0, self.raw_W / 2], [0.0, f, self.raw_H / 2], [0.0, 0.0, 1.0]])
        return intr, self._w2c[idx].clone()

    def preprocess_camera(self, opt, intr, pose):
        """principal point follows the centre crop, both rows follow the resize"""
        shift = torch.tensor([(self.raw_W - self.crop_W) / 2, (self.raw_H - self.crop_H) / 2])
        zoom = torch.tensor([opt.W / self.crop_W, opt.H / self.crop_H])
        out = intr.clone()
        out[:2, 2] -= shift
        out[:2] *= zoom[:, None]
        return out, pose.clone()

    # ---- images
    def get_image(self, opt, idx):
        from PIL import Image
        with Image.open(os.path.join(self.path_image, self.list[idx][0])) as im:
            return im.copy() if im.mode in ("RGB", "L") else im.convert("RGB")

    def preprocess_image(self, opt, image):
        if (self.crop_H, self.crop_W) != (self.raw_H, self.raw_W):            # data.center_crop given
            x0, y0 = (image.width - self.crop_W) // 2, (image.height - self.crop_H) // 2
            image = image.crop((x0, y0, x0 + self.crop_W, y0 + self.crop_H))
        if (image.width, image.height) != (opt.W, opt.H) and opt.data.image_size[0] is not None:
            image = image.resize(size=(opt.W, opt.H))
        pixels = np.array(image, dtype=np.uint8)
        pixels = pixels.reshape(pixels.shape[0], pixels.shape[1], -1)
        return torch.from_numpy(pixels).permute(2, 0, 1).float() / 255

    # ---- samples
    def __getitem__(self, idx):
        intr, pose = self.preprocess_camera(self.opt, *self.get_camera(self.opt, idx))
        return dict(idx=idx, image=self.preprocess_image(self.opt, self.get_image(self.opt, idx)), intr=intr, pose=pose)

    def prefetch_all_data(self, opt):
        """every view of the split, stacked (the engine moves `.all` to the device once)"""
        views = [self[i] for i in range(len(self))]
        self.all = edict({key: torch.stack([torch.as_tensor(v[key]) for v in views]) for key in ("idx", "image", "intr", "pose")})
        return self.all
