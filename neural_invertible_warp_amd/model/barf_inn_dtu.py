"""Mirror of the reference's model/barf_inn_dtu.py `Graph` (524-599) and `NeRF` (601-622): DTU,
learnable poses through `INNPoseParams` (the warp lives on the pose network that the engine
passes into the graph, barf_inn_dtu.py:325-336) + BARF coarse-to-fine encoding."""
import torch

from .. import camera
from ..align_trajectories import backtrack_from_aligning_the_trajectory
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
        """reference barf_inn_dtu.py:538-567.  val / eval: the ground-truth test pose is brought into the
        frame of the optimised poses with the est->gt similarity stored on the pose network by the
        evaluator (evaluation.DTUEvaluator.validate, reference :370-382)."""
        if mode == "train":
            assert iter is not None, "ERROR: Iteration is needed for the c2f embedding in INN"
            return self.pose_net.get_warped_rays_in_world(var, mode, iter)
        if mode in ["val", "eval", "test-optim", "test"]:
            assert hasattr(self.pose_net, "sim3_est_to_gt_c2w")
            sim = self.pose_net.sim3_est_to_gt_c2w
            if sim.type != "traj_align":
                raise ValueError(sim.type)
            pose = backtrack_from_aligning_the_trajectory(var.pose, sim)
            if opt.optim.test_photo and mode != "val":
                pose = camera.pose.compose([var.pose_refine_test, pose])
            return pose
        raise ValueError(mode)

    def get_c2w_pose(self, opt, var, mode=None):
        return camera.pose.invert(self.get_w2c_pose(opt, var, mode))
