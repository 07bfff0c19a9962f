"""Mirror of the reference's model/barf_inn_dtu.py `Graph` (524-599) and `NeRF` (601-622): DTU,
learnable poses through `INNPoseParams` (the warp lives on the pose network that the engine
passes into the graph, barf_inn_dtu.py:325-336) + BARF coarse-to-fine encoding."""
import torch

from .. import camera
from . import barf_inn_llff, nerf_inn_dtu


class NeRF(barf_inn_llff.NeRF):
    """reference barf_inn_dtu.py:601-622 -- identical c2f override to the LLFF model."""


class Graph(nerf_inn_dtu.Graph):

    def __init__(self, opt, pose_net):
        torch.nn.Module.__init__(self)
        self.pose_net = pose_net
        self.nerf = NeRF(opt)
        if opt.nerf.fine_sampling:
            self.nerf_fine = NeRF(opt)
        self.pose_eye = torch.eye(3, 4).to(opt.device)

    def get_pose(self, opt, var, mode=None, iter=None):
        return self.get_w2c_pose(opt, var, mode, iter)

    def get_w2c_pose(self, opt, var, mode=None, iter=None):
        """reference barf_inn_dtu.py:538-567.  The val / eval branch needs the sim3 trajectory
        alignment of the pose-evaluation mixin (barf_inn_dtu.py:173-299), which is outside the render
        path; an already aligned pose can be supplied as var.pose_aligned."""
        if mode == "train":
            assert iter is not None, "ERROR: Iteration is needed for the c2f embedding in INN"
            return self.pose_net.get_warped_rays_in_world(var, mode, iter)
        if mode in ["val", "eval", "test-optim", "test"]:
            if "pose_aligned" not in var:
                raise NotImplementedError("sim3 test-pose alignment is evaluation tooling outside the render path; "
                                          "pass var.pose_aligned (w2c poses in the optimised frame)")
            pose = var.pose_aligned
            if opt.optim.test_photo and mode != "val":
                pose = camera.pose.compose([var.pose_refine_test, pose])
            return pose
        raise ValueError(mode)

    def get_c2w_pose(self, opt, var, mode=None):
        return camera.pose.invert(self.get_w2c_pose(opt, var, mode))
