"""`model=tensorf` on the MI355X renderer: the reference's model/tensorf.py (known camera poses, plain
TensorVMSplit, alpha-mask updates + AABB shrink on the yaml's schedule) on top of the same Graph / NeRF as
bat_hip -- only the pose handling differs (model/nerf.py:703-704: `get_pose` returns the dataset pose)."""
import torch

from .. import ops
from . import bat_hip
from .bat_hip import NeRF  # noqa: F401  (same scene owner and schedules)


class Graph(bat_hip.Graph):
    def get_pose(self, opt, var, mode=None):
        return var.pose

    def resolve_blur(self, opt, mode):
        return None, None, None, None  # the coarse-to-fine blur is bat-only (model/tensorf.py:175)


class Model(bat_hip.Model):
    def build_networks(self, opt, n_views=None):
        self.graph = Graph(opt).to(opt.device)

    def setup_optimizer(self, opt):
        nerf = self.graph.nerf
        self.optim = nerf._get_optimizer(opt)
        nerf.get_current_optimizer = lambda: self.optim

        def register(o):
            self.optim = o
        nerf.register_new_optimizer = register
        self.optim_pose = self.sched_pose = None

    def train_iteration(self, opt, var):
        """model/base.py:154-172 with the schedule progress of model/tensorf.py."""
        g = self.graph
        g.it = self.it
        self.optim.zero_grad()
        var = g.forward(opt, var, mode="train")
        loss = g.compute_loss(opt, var, mode="train")
        loss = self.summarize_loss(opt, var, loss)
        ops.backward(loss.all)
        self.optim.step()
        self.optim.zero_grad()
        self.it += 1
        g.nerf.set_progress(self.it / opt.max_iter)
        return loss
