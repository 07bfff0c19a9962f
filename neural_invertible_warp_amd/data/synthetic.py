"""Procedural scene with the dataset interface of data/llff.py (SURVEY section 8d synthetic inputs): uniform
random images, pin-hole intrinsics fx = fy = 0.8 W with the principal point at the image centre, ground-truth
poses drawn as small se(3) perturbations of the identity.  Used wherever no dataset is mounted."""
import torch

from .. import camera
from ..util import edict


class Dataset(torch.utils.data.Dataset):
    def __init__(self, opt, split="train", subset=None, n_views=None, seed=0):
        super().__init__()
        self.opt, self.split = opt, split
        n = n_views or (18 if split == "train" else 2)
        if subset:
            n = min(n, subset)
        gen = torch.Generator().manual_seed(seed + (0 if split == "train" else 1000))
        self.images = torch.rand(n, 3, opt.H, opt.W, generator=gen)
        self.intr = torch.tensor([[0.8 * opt.W, 0, opt.W / 2], [0, 0.8 * opt.W, opt.H / 2], [0, 0, 1]], dtype=torch.float32).repeat(n, 1, 1)
        self.poses = camera.lie.se3_to_SE3(torch.randn(n, 6, generator=gen) * 0.05)
        self.list = list(range(n))
        # DTU-style samples carry the metric depth range of the scene (data/dtu.py); the DTU graphs read var.depth_range[0]
        self.depth_range = torch.tensor(opt.nerf.depth.range, dtype=torch.float32).repeat(n, 1) if opt.data.get("dataset") == "dtu" else None

    def __len__(self):
        return len(self.list)

    def get_all_camera_poses(self, opt):
        return self.poses.clone()

    def __getitem__(self, idx):
        return dict(idx=idx, image=self.images[idx], intr=self.intr[idx], pose=self.poses[idx])

    def prefetch_all_data(self, opt):
        self.all = edict(idx=torch.arange(len(self)), image=self.images, intr=self.intr, pose=self.poses)
        if self.depth_range is not None:
            self.all.depth_range = self.depth_range
        return self.all
