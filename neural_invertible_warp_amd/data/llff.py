"""LLFF scenes: mirror of the reference's data/llff.py (`Dataset`, :17-140) and of the pieces of
data/base.py it relies on (`preprocess_image` :89-102, `preprocess_camera` :104-110, the collate of
`prefetch_all_data`).  Host-side, runs once before training; everything ends up as four stacked
tensors (`idx, image [N,3,H,W], intr [N,3,3], pose [N,3,4]`) resident on the device.

Conventions (data/llff.py:45-72, :107-134): `poses_bounds.npy` rows are [3x5 | near far]; the 3x4 part
is a camera-to-world pose in [down, right, backwards] order that is re-ordered to [right, up, backwards],
translations and bounds are scaled by 1 / (0.75 * min bound), poses are re-centred on their average, and
`parse_raw_camera` turns them into world-to-camera [right, down, forwards] matrices facing +z.
"""
import os

import numpy as np
import torch
import torch.nn.functional as torch_F

from .. import camera
from ..util import edict


class Dataset(torch.utils.data.Dataset):
    raw_H, raw_W = 3024, 4032

    def __init__(self, opt, split="train", subset=None):
        super().__init__()
        self.opt, self.split = opt, split
        crop = opt.data.get("center_crop") if hasattr(opt.data, "get") else None
        self.crop_H = int(self.raw_H * crop) if crop is not None else self.raw_H
        self.crop_W = int(self.raw_W * crop) if crop is not None else self.raw_W
        if not opt.H or not opt.W:
            opt.H, opt.W = self.crop_H, self.crop_W
        self.root = opt.data.get("root") or "data/llff"
        self.path = "{}/{}".format(self.root, opt.data.scene)
        self.path_image = "{}/images".format(self.path)
        image_fnames = sorted(os.listdir(self.path_image))
        poses_raw, bounds = self.parse_cameras_and_bounds(opt)
        self.list = list(zip(image_fnames, poses_raw, bounds))
        num_val_split = int(len(self) * opt.data.val_ratio)          # the last 10 % are held out (:32-33)
        self.list = self.list[:-num_val_split] if split == "train" else self.list[-num_val_split:]
        if subset:
            self.list = self.list[:subset]

    def __len__(self):
        return len(self.list)

    def parse_cameras_and_bounds(self, opt):
        data = torch.tensor(np.load("{}/poses_bounds.npy".format(self.path)), dtype=torch.float32)
        cam_data = data[:, :-2].view([-1, 3, 5])
        poses_raw = cam_data[..., :4].clone()
        poses_raw[..., 0], poses_raw[..., 1] = cam_data[..., 1], -cam_data[..., 0]
        raw_H, raw_W, self.focal = cam_data[0, :, -1]
        assert self.raw_H == raw_H and self.raw_W == raw_W
        bounds = data[:, -2:].clone()
        scale = 1. / (bounds.min() * 0.75)
        poses_raw[..., 3] *= scale
        bounds *= scale
        return self.center_camera_poses(opt, poses_raw), bounds

    def center_camera_poses(self, opt, poses):
        center = poses[..., 3].mean(dim=0)
        v1 = torch_F.normalize(poses[..., 1].mean(dim=0), dim=0)
        v2 = torch_F.normalize(poses[..., 2].mean(dim=0), dim=0)
        v0 = torch.linalg.cross(v1, v2)
        pose_avg = torch.stack([v0, v1, v2, center], dim=-1)[None]
        return camera.pose.compose([poses, camera.pose.invert(pose_avg)])

    def parse_raw_camera(self, opt, pose_raw):
        pose_flip = camera.pose(R=torch.diag(torch.tensor([1, -1, -1])))
        pose = camera.pose.compose([pose_flip, pose_raw[:3]])          # OpenGL -> OpenCV axes
        pose = camera.pose.invert(pose)                                # c2w -> w2c
        return camera.pose.compose([pose_flip, pose])                  # face +z, like the identity initialisation

    def get_all_camera_poses(self, opt):
        return torch.stack([self.parse_raw_camera(opt, tup[1]) for tup in self.list], dim=0)

    def get_image(self, opt, idx):
        import PIL.Image
        with PIL.Image.open("{}/{}".format(self.path_image, self.list[idx][0])) as im:
            return im.convert("RGB") if im.mode not in ("RGB", "L") else im.copy()

    def get_camera(self, opt, idx):
        intr = torch.tensor([[self.focal, 0, self.raw_W / 2], [0, self.focal, self.raw_H / 2], [0, 0, 1]]).float()
        return intr, self.parse_raw_camera(opt, self.list[idx][1])

    def preprocess_image(self, opt, image):
        if self.crop_H != self.raw_H or self.crop_W != self.raw_W:
            left, top = (image.width - self.crop_W) // 2, (image.height - self.crop_H) // 2
            image = image.crop((left, top, left + self.crop_W, top + self.crop_H))
        if opt.data.image_size[0] is not None:
            image = image.resize((opt.W, opt.H))
        arr = np.asarray(image, dtype=np.uint8)
        if arr.ndim == 2:
            arr = arr[..., None]
        return torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255)

    def preprocess_camera(self, opt, intr, pose):
        intr, pose = intr.clone(), pose.clone()
        intr[0, 2] -= (self.raw_W - self.crop_W) / 2
        intr[1, 2] -= (self.raw_H - self.crop_H) / 2
        intr[0] *= opt.W / self.crop_W
        intr[1] *= opt.H / self.crop_H
        return intr, pose

    def __getitem__(self, idx):
        opt = self.opt
        intr, pose = self.preprocess_camera(opt, *self.get_camera(opt, idx))
        return dict(idx=idx, image=self.preprocess_image(opt, self.get_image(opt, idx)), intr=intr, pose=pose)

    def prefetch_all_data(self, opt):
        samples = [self[i] for i in range(len(self))]
        self.all = edict(idx=torch.tensor([s["idx"] for s in samples]), image=torch.stack([s["image"] for s in samples]),
                         intr=torch.stack([s["intr"] for s in samples]), pose=torch.stack([s["pose"] for s in samples]))
        return self.all
