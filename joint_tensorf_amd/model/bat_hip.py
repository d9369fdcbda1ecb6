"""`bat_hip`: drop-in for the reference's `model/bat.py` (+ the parts of model/tensorf.py,
model/nerf.py and model/base.py it inherits) on the training hot path.

Same classes and call contract as the reference:
    Graph.forward(opt, var, mode) -> var with rgb [B,r,3], depth/opacity [B,r,1], ray_idx, current_pose
    Graph.compute_loss(opt, var, mode) -> dict of 0-dim tensors
    Graph.get_pose / render / render_rays / render_by_slices
    NeRF: resolution / sample-count / learning-rate schedule owner (model/tensorf.py:278-524)
    Model: build_networks / setup_optimizer / train_iteration (model/bat.py:30-116, base.py:154-172)
What differs is *how* a forward is executed: pose composition, ray generation for the sampled pixels
only, sampling, interpolation, MLP, compositing and their backward are HIP kernels
(joint_tensorf_amd/csrc), and the schedule scalars stay on the host (no `.cpu()` reads of Parameters).
Logging / visualisation / dataset loading of the reference engine are out of scope.
"""
import math
import os

import numpy as np
import torch

from .. import ops
from .. import tensorf_repr
from ..options import Opt


_SCHEDULE_XS = {}


def interp_schedule(x, schedule, left=0.0, right=1.0):
    """util.interp_schedule (util.py:217-225) on host floats."""
    x = float(x)
    assert left <= x <= right
    key = (left, right, len(schedule))
    xs = _SCHEDULE_XS.get(key)
    if xs is None:
        xs = _SCHEDULE_XS[key] = np.linspace(left, right, len(schedule))
    return float(np.interp(x, xs, schedule))


def _has(o, k):
    return (k in o) if isinstance(o, dict) else hasattr(o, k)


class NeRF(torch.nn.Module):
    """tensorf.NeRF + bat.NeRF: owns the scene tensors and every schedule that resizes them."""

    def __init__(self, opt):
        super().__init__()
        self.device = opt.device
        self.register_new_optimizer = None
        self.get_current_optimizer = None
        self.lr_decay_duration = opt.max_iter if opt.optim.lr_decay_iters < 0 else opt.optim.lr_decay_iters
        self.lr_decay_factor = opt.optim.lr_decay_target_ratio ** (1 / self.lr_decay_duration)
        self.update_alphamask_iters = opt.train_schedule.update_alphamask_iters
        self.upsample_list = opt.train_schedule.upsample_iters
        self.bbox = torch.tensor(opt.data.scene_bbox).to(torch.float).view(2, 3)
        n_list = torch.round(torch.exp(torch.linspace(np.log(opt.train_schedule.n_voxel_init),
                                                      np.log(opt.train_schedule.n_voxel_final),
                                                      len(self.upsample_list) + 1))).long().tolist()[1:]
        self.reset(opt, bbox=opt.data.scene_bbox, n_voxel_list=n_list, n_voxels=opt.train_schedule.n_voxel_init,
                   alphamask_resolution=self._find_resolution(opt, opt.train_schedule.n_voxel_init),
                   lr_basis=opt.optim.lr_basis, lr_index=opt.optim.lr_index,
                   TV_weight_color=opt.loss_weight.TV_color, TV_weight_density=opt.loss_weight.TV_density)
        self.define_network(opt)
        # Parameters so that they are checkpointed like the reference's (model/bat.py:373-374);
        # the host mirrors are what the schedule code reads (no device->host sync per iteration).
        self.progress = torch.nn.Parameter(torch.tensor(0.0))
        self.test_time_progress = torch.nn.Parameter(torch.tensor(0.0))
        self.progress_host = 0.0
        self.test_time_progress_host = 0.0
        # the device copies are written when somebody reads the module's state (a fill launch per iteration otherwise)
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module.flush_progress())

    def set_progress(self, value):
        self.progress_host = float(value)

    def set_test_time_progress(self, value):
        self.test_time_progress_host = float(value)

    def flush_progress(self):
        """the host mirrors -> the checkpointed Parameters (model/bat.py:373-374)"""
        with torch.no_grad():
            self.progress.fill_(self.progress_host)
            self.test_time_progress.fill_(self.test_time_progress_host)

    def reset(self, opt, bbox, n_voxel_list, n_voxels, alphamask_resolution, lr_basis, lr_index, TV_weight_color,
              TV_weight_density):
        self.bbox = torch.as_tensor(bbox).to(torch.float).view(2, 3).cpu()  # host state (a checkpoint may hand a device tensor)
        self.n_voxel_list = list(n_voxel_list)
        self.n_voxels = n_voxels
        self.resolution = self._find_resolution(opt, self.n_voxels)
        self.n_samples = self._find_n_samples(opt, self.resolution)
        self.alphamask_resolution = alphamask_resolution
        self.lr_basis, self.lr_index = lr_basis, lr_index
        self.TV_weight_color, self.TV_weight_density = TV_weight_color, TV_weight_density
        opt.loss_weight.TV_color = TV_weight_color
        opt.loss_weight.TV_density = TV_weight_density

    def define_network(self, opt):
        arch = opt.arch
        dens = [int(c) for c in arch.tensorf.density_components]
        app = [int(c) for c in arch.tensorf.color_components]
        cls = getattr(tensorf_repr, arch.tensorf.model)
        self.tensorf = cls(self.bbox, self.resolution, opt.device, density_n_comp=dens, appearance_n_comp=app,
                           app_dim=arch.shading.app_dim, near_far=opt.nerf.depth.range,
                           shadingMode=arch.shading.model, alphaMask_thres=opt.train_schedule.alpha_mask_threshold,
                           density_shift=arch.density_shift, distance_scale=arch.distance_scale,
                           pos_pe=arch.shading.pose_pe, view_pe=arch.shading.view_pe, fea_pe=arch.shading.fea_pe,
                           featureC=arch.shading.mlp_hidden_dim, step_ratio=opt.nerf.step_ratio,
                           fea2denseAct=arch.feature_to_density_activation, dtype=torch.float32,
                           volume_init_scale=arch.tensorf.volume_init_scale,
                           rayMarch_weight_thres=arch.tensorf.rayMarch_weight_thres,
                           volume_init_bias=arch.tensorf.volume_init_bias)

    def _find_resolution(self, opt, n_voxels):
        lo, hi = self.bbox[0], self.bbox[1]
        voxel = ((hi - lo).prod() / n_voxels).pow(1 / 3)
        # (the box follows the AABB shrink of an alpha-mask update onto the device: the yaml's per-axis scale goes where it is)
        scale = torch.as_tensor(opt.train_schedule.resolution_scale_init, dtype=lo.dtype, device=lo.device)
        return ((hi - lo) / voxel * scale).long().tolist()

    def _find_n_samples(self, opt, resolution):
        return min(int(opt.nerf.sample_intvs), int(np.linalg.norm(resolution) / opt.nerf.step_ratio))

    def _get_optimizer(self, opt, it=0, lr_basis=None, lr_index=None):
        if lr_basis is None and lr_index is None:
            reset = opt.optim.lr_upsample_reset and it in self.upsample_list
            scale = 1.0 if reset else opt.optim.lr_decay_target_ratio ** (it / opt.max_iter)
            self.lr_basis, self.lr_index = opt.optim.lr_basis * scale, opt.optim.lr_index * scale
        else:
            self.lr_basis, self.lr_index = lr_basis, lr_index
        groups = self.tensorf.get_optparam_groups(self.lr_index, self.lr_basis)
        if opt.optim.algo == "Adam":
            # same update rule, state and param-group interface as the reference's torch.optim.Adam(betas=(0.9,
            # 0.99)) (model/tensorf.py:474-475); on the GPU all tensors are stepped by one HIP launch (optim.VMAdam)
            if str(opt.device).startswith("cuda"):
                from ..optim import VMAdam
                return VMAdam(groups, betas=(0.9, 0.99))
            return torch.optim.Adam(groups, betas=(0.9, 0.99))
        return getattr(torch.optim, opt.optim.algo)(groups)

    def update_schedule(self, opt, it):
        """model/tensorf.py:399-447."""
        assert self.register_new_optimizer is not None and self.get_current_optimizer is not None
        if it in self.upsample_list:
            if it == self.upsample_list[0]:
                opt.train_schedule.resolution_scale_init = [1.0, 1.0, 1.0]
            self.n_voxels = self.n_voxel_list.pop(0)
            self.resolution = self._find_resolution(opt, self.n_voxels)
            self.tensorf.upsample_volume_grid(self.resolution)
            self.n_samples = self._find_n_samples(opt, self.resolution)
            self.register_new_optimizer(self._get_optimizer(opt, it))
        else:
            for g in self.get_current_optimizer().param_groups:
                g["lr"] = g["lr"] * self.lr_decay_factor
            self.lr_basis *= self.lr_decay_factor
            self.lr_index *= self.lr_decay_factor
        if it in self.update_alphamask_iters:
            self._update_alphamask(it)
        if opt.loss_weight.TV_density > 0:
            opt.loss_weight.TV_density *= self.lr_decay_factor
            self.TV_weight_density = opt.loss_weight.TV_density
        if opt.loss_weight.TV_color > 0:
            opt.loss_weight.TV_color *= self.lr_decay_factor
            self.TV_weight_color = opt.loss_weight.TV_color

    def _update_alphamask(self, it):
        """model/tensorf.py:480-489: only while the grid is below 256^3 (never the case at the iterations the BAT
        yamls name); the first update also shrinks the box.  As in the reference the optimizer is NOT rebuilt
        here: it is rebuilt by the next upsampling."""
        if it not in self.update_alphamask_iters:
            return
        r = self.resolution
        if r[0] * r[1] * r[2] < 256 ** 3:
            self.alphamask_resolution = r
            new_aabb = self.tensorf.updateAlphaMask(tuple(r))
            if it == self.update_alphamask_iters[0]:
                self.tensorf.shrink(new_aabb)
                self.bbox = new_aabb

    def freeze_scene(self, opt):
        self.tensorf.freeze_scene(opt)

    def unfreeze_scene(self, opt):
        self.tensorf.unfreeze_scene(opt)

    # ---- state besides the state_dict (model/tensorf.py:491-524) ---------------------------------------------------
    def get_reset_kwargs(self):
        return {"bbox": self.bbox, "n_voxel_list": self.n_voxel_list, "n_voxels": self.n_voxels,
                "alphamask_resolution": self.alphamask_resolution, "lr_basis": self.lr_basis, "lr_index": self.lr_index,
                "TV_weight_color": self.TV_weight_color, "TV_weight_density": self.TV_weight_density}

    def save_param_state(self):
        ckpt = self.tensorf.save_param_state()
        ckpt.update({"nerf_reset_kwargs": self.get_reset_kwargs()})
        return ckpt

    def load_param_state(self, opt, ckpt):
        """Resize the scene to the checkpointed grid (the state_dict that follows would not fit a fresh model after
        an upsampling or a shrink) and, when training resumes, rebuild the optimizer at the checkpointed rates."""
        kw = dict(ckpt["nerf_reset_kwargs"])
        self.reset(opt=opt, **kw)
        self.tensorf.load_param_state(ckpt)
        self.tensorf.to(opt.device)
        if self.register_new_optimizer is not None:
            self.register_new_optimizer(self._get_optimizer(opt, ckpt["iter"], kw["lr_basis"], kw["lr_index"]))


class Graph(torch.nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.it = 0
        self.nerf = NeRF(opt)
        self.nerf.get_parent = lambda: self
        self.pose_eye = torch.eye(3, 4, device=opt.device)
        self.tvloss = tensorf_repr.TVLoss()
        self.sim3 = None
        # ray-sharded data parallelism (Model.enable_data_parallel).  ray_shard = (mode, rank, world):
        #   "pixel"  every rank renders ALL views on the points rank::world of the iteration's pixel lattice,
        #   "view"   every rank renders the FULL lattice on the views rank::world (the caller hands it those views only;
        #            global_views = the view count the lattice stride is derived from, model/nerf.py:660-662)
        #            -- either way the ranks together render exactly the iteration one process would;
        #   "offset" every rank renders a whole lattice of its own, shifted inside the shared draw's class
        #            (dist.rank_lattice_offsets: weak scaling).
        self.ray_shard = None
        self.global_views = None
        self.eval_graph = None
        self.lattice_override = None  # callable(step) -> (ray_idx, grid_H, grid_W); see graphed.GraphedTrainStep

    def save_param_state(self):
        """model/tensorf.py:270-276"""
        return self.nerf.save_param_state()

    def load_param_state(self, opt, ckpt):
        self.nerf.load_param_state(opt, ckpt)

    # ---- pose (model/bat.py:341-367) -------------------------------------------------------------
    def get_pose(self, opt, var, mode=None):
        if mode == "train":
            # all views in their own order (the BAT loops train on every view in every iteration): the gather
            # weight[idx] and its scatter-add backward are the identity -- a dozen tiny launches saved per iteration
            full = self._is_all_views(var.idx, self.se3_refine.weight.shape[0])
            var.se3_refine = self.se3_refine.weight if full else self.se3_refine.weight[var.idx]
            if opt.data.dataset == "blender":
                noise = (self.pose_noise if full else self.pose_noise[var.idx]) if opt.camera.noise else None
                if noise is not None:
                    var.pose_noise = noise
                return ops.train_pose(var.se3_refine, noise, var.pose)
            return ops.train_pose(var.se3_refine, None, self.pose_eye)
        if mode in ("val", "eval", "test-optim"):
            # the view's pose in the optimised cameras' frame does not change while its refinement is optimised: callers that
            # iterate (test-time optimisation) hand it over once as var.pose_aligned instead of a dozen small launches per call
            pose = var.get("pose_aligned")
            if pose is None:
                pose = self.aligned_pose(var.pose)
            if opt.optim.test_photo and mode != "val":
                se3 = var.get("se3_refine_test")
                if mode == "test-optim" and se3 is not None and se3.requires_grad and torch.is_grad_enabled() \
                        and se3.dim() == 2 and se3.shape[0] == pose.shape[0]:   # (one refinement per view: no broadcasting)
                    # compose([se3_to_SE3(se3), pose]) (model/bat.py:360-363) as ONE launch each way: the pose kernel of the
                    # training path with the aligned pose as its base (round 6; was two matrix products, a sum and a cat on top
                    # of var.pose_refine_test, with their autograd nodes)
                    return ops.train_pose(se3, None, pose)
                pr = var.pose_refine_test
                pose = torch.cat([pose[..., :3] @ pr[..., :3], pose[..., :3] @ pr[..., 3:] + pose[..., 3:]], -1)
            return pose
        return var.pose

    def aligned_pose(self, pose):
        """model/bat.py:354-359: a ground-truth pose carried into the frame of the optimised cameras (self.sim3)"""
        sim3 = self.sim3
        R, t = pose[..., :3], pose[..., 3:]
        center = (-R.transpose(-1, -2) @ t)[..., 0]  # camera centres in world coordinates
        center_aligned = (center - sim3.t0) / sim3.s0 @ sim3.R * sim3.s1 + sim3.t1
        R_aligned = R @ sim3.R
        t_aligned = (-R_aligned @ center_aligned[..., None])[..., 0]
        return torch.cat([R_aligned, t_aligned[..., None]], -1)

    def _is_all_views(self, idx, n):
        """idx == arange(n)?  Decided once per index tensor (one host read), remembered by its storage."""
        if not torch.is_tensor(idx) or idx.dim() != 1 or idx.shape[0] != n:
            return False
        key = (idx.data_ptr(), idx._version, n)
        memo = self.__dict__.setdefault("_all_views_memo", {})
        if key not in memo:
            if len(memo) > 64:
                memo.clear()
            memo[key] = bool(torch.equal(idx, torch.arange(n, device=idx.device, dtype=idx.dtype)))
        return memo[key]

    def _lattice_base(self, opt, step, ny, nx):
        """pixel indices of the (ny, nx) lattice at offset (0, 0), cached: the lattice of an iteration is base + ox +
        oy W -- one launch instead of two aranges, a meshgrid and three elementwise kernels."""
        key = (int(step), int(ny), int(nx), int(opt.W), str(opt.device))
        memo = self.__dict__.setdefault("_lattice_memo", {})
        if key not in memo:
            sx = torch.arange(nx, device=opt.device) * step
            sy = torch.arange(ny, device=opt.device) * step
            memo[key] = (sx[None, :] + sy[:, None] * opt.W).reshape(-1)
        return memo[key]

    def _lattice_at(self, opt, step, ny, nx, ox, oy):
        """pixel indices of the lattice at offset (ox, oy): base + ox + oy W, remembered per offset (step^2 of them, a few
        thousand entries of a few KB: the add is one launch of an iteration's ~40)."""
        key = (int(step), int(ny), int(nx), int(ox), int(oy), int(opt.W), str(opt.device))
        memo = self.__dict__.setdefault("_lattice_at_memo", {})
        t = memo.get(key)
        if t is None:
            if len(memo) >= 8192:
                memo.clear()
            t = memo[key] = self._lattice_base(opt, step, ny, nx) + (ox + oy * opt.W)
        return t

    @staticmethod
    def lattice_step(opt, batch_size):
        """pixel stride of the all_view_rand_grid lattice (model/nerf.py:660-662)."""
        rays_per_view = opt.nerf.n_rays // batch_size
        return math.ceil((opt.H * opt.W // rays_per_view) ** 0.5)

    # ---- forward (model/nerf.py:650-679) ---------------------------------------------------------
    def forward(self, opt, var, mode=None):
        batch_size = len(var.idx)
        pose = self.get_pose(opt, var, mode=mode)
        var.current_pose = pose
        if mode in ("train", "test-optim"):
            strat = opt.nerf.ray_sampling_strategy
            if strat == "all_view_rand_rays":
                var.ray_idx = torch.randperm(opt.H * opt.W, device=opt.device)[:opt.nerf.n_rays // batch_size]
            elif strat == "all_view_rand_grid" and self.lattice_override is not None:
                # hipGraph capture / replay (graphed.GraphedTrainStep): the lattice is computed on the device from
                # offsets that live in device memory; the host draws happen outside the graph
                step = self.lattice_step(opt, batch_size)
                var.ray_idx, var.grid_H, var.grid_W = self.lattice_override(step)
                var.ray_grid_step = step
            elif strat == "all_view_rand_grid":
                shard = self.ray_shard
                step = self.lattice_step(opt, self.global_views or batch_size)
                ox, oy = np.random.randint(step), np.random.randint(step)  # shared by all ranks (same seeded generator)
                if shard is not None and shard[0] == "offset":  # same draw, same count, other pixels
                    from ..dist import rank_lattice_offsets
                    ox, oy = rank_lattice_offsets(ox, oy, step, opt.W, opt.H, shard[1], shard[2])
                nx, ny = len(range(ox, opt.W, step)), len(range(oy, opt.H, step))
                var.ray_idx = self._lattice_at(opt, step, ny, nx, ox, oy)
                var.ray_grid_step, var.grid_H, var.grid_W = step, ny, nx
                if shard is not None and shard[0] == "pixel":
                    var.ray_idx = var.ray_idx[shard[1]::shard[2]]
                    var.grid_H = var.grid_W = None  # a shard is not a 2-D grid (compute_loss: TV_depth)
                    # the photometric term is the mean over the GLOBAL ray batch: local mean x local / global count
                    var.dp_render_scale = len(range(shard[1], nx * ny, shard[2])) / float(nx * ny)
                elif shard is not None and shard[0] == "view":
                    var.dp_render_scale = batch_size / float(self.global_views or batch_size)
                elif shard is not None:
                    var.dp_render_scale = 1.0 / shard[2]
            else:
                assert strat == "single_view_rand_rays"
                var.ray_idx = torch.randperm(opt.H * opt.W, device=opt.device)[:opt.nerf.n_rays]
            ret = None
            if mode == "test-optim":
                var.pop("fused_render_loss", None)
                ret = self.render_test_fused(opt, pose, var)   # one launch for render + loss + backward, when it applies
            if ret is None:
                ret = self.render(opt, pose, intr_inv=var.intr_inv, ray_idx=var.ray_idx, mode=mode, intr=var.intr)
        elif _has(opt.nerf, "eval_graph") and opt.nerf.eval_graph and not torch.is_grad_enabled():
            # the whole sliced render of a view as one hipGraph, replayed per held-out view (graphed.GraphedEvalRender)
            if self.eval_graph is None:
                from ..graphed import GraphedEvalRender
                self.eval_graph = GraphedEvalRender(self)
            ret = self.eval_graph.render(opt, pose, var.intr_inv, var.intr)
        else:
            ret = self.render_by_slices(opt, pose, intr_inv=var.intr_inv, mode=mode, intr=var.intr)
        var.update(ret)
        return var

    # ---- render (model/tensorf.py:144-167) -------------------------------------------------------
    def render(self, opt, pose, intr_inv=None, ray_idx=None, mode=None, intr=None):
        if ray_idx is None:
            ray_idx = torch.arange(opt.H * opt.W, device=pose.device)
        ndc_near = float(opt.arch.ndc_near_plane) if _has(opt.arch, "ndc_near_plane") else 1.0
        # rays for the sampled pixels only, NDC folded in (camera.py:231-261, 303-340)
        center, ray = ops.ray_gen(pose, intr_inv, intr, ray_idx, opt.W, ndc=bool(opt.camera.ndc), ndc_near=ndc_near)
        return self.render_rays(opt, center, ray, mode, n_views=len(pose), n_pixels_per_view=center.shape[1])

    def render_test_fused(self, opt, pose, var):
        """Test-time pose optimisation (model/bat.py:265-292) through the single-launch kernel (csrc/jt_fused.hip): the
        scene is frozen, so render + photometric loss + the backward to the rays are ONE launch and nothing is recorded in
        between.  Returns None when the staged path must run instead: opt.optim.test_fused not set, a factor blur
        active for this call (LLFF's test_kernel_schedule), scene parameters that want gradients, or no GT images."""
        if not (_has(opt.optim, "test_fused") and opt.optim.test_fused):
            return None   # opt-in: measured on MI355X it does not beat the staged kernels
        tf = self.nerf.tensorf
        if not (pose.is_cuda and torch.is_grad_enabled() and torch.is_tensor(var.get("image"))
                and not any(p.requires_grad for p in tf.parameters())):
            return None
        if getattr(tf, "_pose_fused_ok", None) is None:
            # the single-launch kernel exists for the two BAT shading configurations only (jt_pose_fused_workspace_bytes
            # returns 0 otherwise): any other scene takes the staged path instead of raising; asked once per
            # scene object, BEFORE any host draw is consumed
            from .._lib import fused_lib
            tf._pose_fused_ok = fused_lib().jt_pose_fused_workspace_bytes(
                tf._render_cfg(self.nerf.n_samples, bool(opt.camera.ndc), bool(opt.nerf.setbg_opaque)).scene()) > 0
        if not tf._pose_fused_ok:
            return None
        # the host draws of render_rays happen whichever path runs (same random streams)
        pd, pc, c2f_mode, ksize = self.resolve_blur(opt, "test-optim")
        if c2f_mode is not None:
            self._blur_memo = (pd, pc, c2f_mode, ksize)   # render_rays must not draw a second time
            return None
        if opt.data.dataset != "blender":
            tf.near_far[0] = interp_schedule(self.nerf.progress_host, opt.tensorf_near_plane_schedule)
            opt.nerf.depth.range[0] = tf.near_far[0]
        view_pe = interp_schedule(self.nerf.progress_host, opt.c2f_view_pe_schedule) if _has(opt, "c2f_view_pe_schedule") else 1.0
        fea_pe = interp_schedule(self.nerf.progress_host, opt.c2f_fea_pe_schedule) if _has(opt, "c2f_fea_pe_schedule") else 1.0
        ndc_near = float(opt.arch.ndc_near_plane) if _has(opt.arch, "ndc_near_plane") else 1.0
        center, ray = ops.ray_gen(pose, var.intr_inv, var.intr, var.ray_idx, opt.W, ndc=bool(opt.camera.ndc), ndc_near=ndc_near)
        B, r = center.shape[0], center.shape[1]
        render, rgb, depth, opacity = tf.render_pose_fused(
            opt, center.reshape(-1, 3), ray.reshape(-1, 3), var.image, var.ray_idx, r, white_bg=opt.nerf.setbg_opaque,
            ndc_ray=opt.camera.ndc, N_samples=self.nerf.n_samples, view_pe_progress=view_pe, fea_pe_progress=fea_pe)
        return Opt(rgb=rgb.view(B, r, 3), depth=depth.view(B, r, 1), opacity=opacity.view(B, r, 1), fused_render_loss=render)

    def render_by_slices(self, opt, pose, intr_inv=None, mode=None, intr=None):
        """model/nerf.py:728-740.  Rays are independent and nothing is random at eval time, so the slice size only
        sets the launch granularity: opt.nerf.eval_slice_rays (default 32768 pixels per view and slice, against
        the reference's opt.nerf.n_rays = 2048) keeps the 800x800 render at 20 launches per kernel instead of 313."""
        acc = dict(rgb=[], depth=[], opacity=[])
        step = max(int(opt.nerf.n_rays), int(opt.nerf.eval_slice_rays) if _has(opt.nerf, "eval_slice_rays") else 32768)
        for c in range(0, opt.H * opt.W, step):
            ray_idx = torch.arange(c, min(c + step, opt.H * opt.W), device=opt.device)
            ret = self.render(opt, pose, intr_inv=intr_inv, ray_idx=ray_idx, mode="vis", intr=intr)
            for k in acc:
                acc[k].append(ret[k])
        return Opt({k: torch.cat(v, dim=1) for k, v in acc.items()})

    # ---- schedule glue (model/tensorf.py:169-267) -------------------------------------------------
    def resolve_blur(self, opt, mode):
        if not (opt.model in ("bat", "bat_hip") and opt.c2f_mode != "None"):
            return None, None, None, None
        c2f_mode = opt.c2f_mode
        if c2f_mode not in ("uniform-gaussian", "uniform-average"):
            raise Exception("unknown c2f_mode")
        pd = interp_schedule(self.nerf.progress_host, opt.c2f_schedule_density)
        pc = interp_schedule(self.nerf.progress_host, opt.c2f_schedule_color)
        if mode != "vis" and _has(opt, "c2f_random_density_blur") and opt.c2f_random_density_blur:
            if mode == "train" and _has(opt, "sync_2d_3d_scales") and opt.sync_2d_3d_scales:
                scale = self.scale
            else:
                scale = np.random.choice(opt.c2f_random_density_scale_pool)
            pd = pd * scale
        if mode == "test-optim" and opt.data.dataset == "llff":
            pd = interp_schedule(self.nerf.test_time_progress_host, opt.optim.test_kernel_schedule)
        if max(pd, pc) < 0.001:
            return None, None, None, None
        return pd, pc, c2f_mode, opt.c2f_kernel_size

    def render_rays(self, opt, center, ray, mode=None, n_views=None, n_pixels_per_view=None):
        batch_size = n_views if n_views else center.shape[0]
        dim1 = n_pixels_per_view if n_pixels_per_view else center.shape[1]
        memo = self.__dict__.pop("_blur_memo", None)   # render_test_fused already took this call's draw
        pd, pc, c2f_mode, ksize = memo if memo is not None else self.resolve_blur(opt, mode)
        tf = self.nerf.tensorf
        if opt.data.dataset != "blender":
            tf.near_far[0] = interp_schedule(self.nerf.progress_host, opt.tensorf_near_plane_schedule)
            opt.nerf.depth.range[0] = tf.near_far[0]
        view_pe = interp_schedule(self.nerf.progress_host, opt.c2f_view_pe_schedule) if _has(opt, "c2f_view_pe_schedule") else 1.0
        fea_pe = interp_schedule(self.nerf.progress_host, opt.c2f_fea_pe_schedule) if _has(opt, "c2f_fea_pe_schedule") else 1.0
        # a training render: tell the scene class which weights Model.summarize_loss will put on the regularisers (the
        # upstream gradient of their sums), so that value and gradient come out of one launch (ops.RenderRays)
        tf.__dict__.pop("reg_weights_hint", None)
        provider = self.__dict__.get("reg_weight_provider")
        if mode == "train" and provider is not None:
            tf.reg_weights_hint = provider(opt)
        rgb, depth, opacity = tf.forward(
            opt, center=center.reshape(-1, 3), ray_dir=ray.reshape(-1, 3), white_bg=opt.nerf.setbg_opaque,
            is_train=(mode == "train" and opt.nerf.sample_stratified),
            is_test_optim=(mode == "test-optim") and (opt.data.dataset == "llff"), ndc_ray=opt.camera.ndc,
            N_samples=self.nerf.n_samples, c2f_parameter_density=pd, c2f_parameter_color=pc, c2f_mode=c2f_mode,
            c2f_kernel_size=ksize, fea_pe_progress=fea_pe, view_pe_progress=view_pe)
        return Opt(rgb=rgb.view(batch_size, dim1, 3), depth=depth.view(batch_size, dim1, 1),
                   opacity=opacity.view(batch_size, dim1, 1))

    # ---- losses (model/tensorf.py:96-142, base.py:259-261) ----------------------------------------
    @staticmethod
    def MSE_loss(pred, label=0):
        return ((pred.contiguous() - label) ** 2).nanmean()

    def compute_loss(self, opt, var, mode=None):
        loss = Opt()
        batch_size = len(var.idx)
        if opt.loss_weight.render is not None and mode == "test-optim" and var.get("fused_render_loss") is not None:
            loss.render = var.fused_render_loss   # came out of the render launch itself (Graph.render_test_fused)
        elif opt.loss_weight.render is not None:
            edge_on = False
            if _has(opt, "edge_mask_on_render_loss") and opt.edge_mask_on_render_loss:
                edge_on = (self.it % 2 == 0) if (_has(opt, "alternate_edge_loss") and opt.alternate_edge_loss) else True
            use_edge = edge_on and mode == "train" and self.it < opt.edge_mask_before_iter
            soft = use_edge and _has(opt, "soft_edge_loss") and opt.soft_edge_loss
            fused = var.rgb.is_cuda and mode in ("train", "test-optim") and not soft
            if fused:
                # GT gather + (edge-split) nanmean MSE in one kernel each way (jt_render_loss_*)
                if use_edge:
                    loss.render = ops.render_loss(var.rgb, var.image, var.ray_idx, var.train_edge_masks,
                                                  opt.edge_loss_factor, opt.non_edge_loss_factor)
                else:
                    loss.render = ops.render_loss(var.rgb, var.image, var.ray_idx)
            else:
                image = var.image.view(batch_size, 3, opt.H * opt.W).permute(0, 2, 1)
                if mode in ("train", "test-optim"):
                    image = image[:, var.ray_idx]
                if use_edge:
                    m = var.train_edge_masks[:, var.ray_idx].view(batch_size, len(var.ray_idx), 1).expand(-1, -1, 3)
                    if soft:
                        m = m * opt.edge_loss_factor + opt.non_edge_loss_factor
                        loss.render = self.MSE_loss(var.rgb * m, image * m)
                    else:
                        loss.render = opt.edge_loss_factor * self.MSE_loss(var.rgb * m, image * m) + \
                            opt.non_edge_loss_factor * self.MSE_loss(var.rgb * (1 - m), image * (1 - m))
                else:
                    loss.render = self.MSE_loss(var.rgb, image)
        tf = self.nerf.tensorf
        tf.reg_with_tv = (float(opt.loss_weight.TV_density or 0) != 0.0, float(opt.loss_weight.TV_color or 0) != 0.0)
        loss.L1 = tf.density_L1()
        loss.TV_density = tf.TV_loss_density(self.tvloss)
        loss.TV_color = tf.TV_loss_app(self.tvloss)
        if mode == "train" and opt.nerf.ray_sampling_strategy == "all_view_rand_grid" and "TV_depth" in opt.loss_weight \
                and var.get("grid_H") is None:
            # a pixel shard of the lattice has no neighbours to difference; the BAT yamls weight the term 0.0
            if float(opt.loss_weight.TV_depth or 0.0) != 0.0:
                raise NotImplementedError("TV_depth with a non-zero weight under pixel-sharded data parallelism")
        elif mode == "train" and opt.nerf.ray_sampling_strategy == "all_view_rand_grid" and "TV_depth" in opt.loss_weight:
            if float(opt.loss_weight.TV_depth or 0.0) == 0.0 and var.depth.is_cuda:
                # weighted 0.0 (both BAT yamls): the value is only logged -- one launch instead of a dozen elementwise ones
                loss.TV_depth = ops.tv_depth_value(var.depth, batch_size, var.grid_H, var.grid_W)
            else:
                d = var.depth.reshape(batch_size, var.grid_H, var.grid_W)
                loss.TV_depth = torch.pow(d[:, 1:, :] - d[:, :-1, :], 2).sum() / var.grid_H + \
                    torch.pow(d[:, :, 1:] - d[:, :, :-1], 2).sum() / var.grid_W
            if self.it > opt.loss_weight.TV_depth_until_iters:
                opt.loss_weight.TV_depth = 0.0
        return loss


class Model(torch.nn.Module):
    """Training driver on the hot path: bat.Model / tensorf.Model / base.Model.train_iteration."""

    def __init__(self, opt):
        super().__init__()
        self.it = 0
        self.iter_start = self.epoch_start = 0
        self.train_data = self.test_data = None

    # ---- the lifecycle train_3d.py:70-80,95-107 drives ---------------------------------------------------------
    def load_dataset(self, opt, eval_split="val", train_split="train"):
        """base.Model.load_dataset + nerf.Model.load_dataset (model/base.py:25-34, model/nerf.py:35-40): the training
        views are prefetched and kept resident on the device.  `opt.data.train_views` / `opt.data.test_views` may hold
        ready-made `var`-layout dicts (idx, image, pose, intr, intr_inv); otherwise joint_tensorf_amd.data.load picks the
        reference's loader when importable and the synthetic scene of SURVEY 8(d) if not."""
        from .. import data as jdata
        sub = opt.data.get("train_sub", None)
        tv = opt.data.get("train_views", None)
        self.train_data = jdata.DictDataset(opt, tv, train_split) if tv is not None else jdata.load(opt, train_split, sub)
        self.train_loader = self.train_data.setup_loader(opt, shuffle=False)
        if opt.data.get("val_on_test", False):
            eval_split = "test"
        sub = opt.data.get("test_sub" if eval_split == "test" else "val_sub", None)
        ev = opt.data.get("test_views", None)
        self.test_data = jdata.DictDataset(opt, ev, eval_split) if ev is not None else jdata.load(opt, eval_split, sub)
        self.test_loader = self.test_data.setup_loader(opt, shuffle=False)
        self.train_data.prefetch_all_data(opt)
        self.train_data.all = Opt({k: (v.to(opt.device) if torch.is_tensor(v) else v)
                                   for k, v in dict(self.train_data.all).items()})
        self.n_train_views = len(self.train_data.all.idx)

    def build_networks(self, opt, n_views=None):
        """model/bat.py:30-47; n_views defaults to the loaded training set (the reference reads len(self.train_data))."""
        if n_views is None:
            n_views = len(self.train_data)
        self.graph = Graph(opt).to(opt.device)
        if opt.camera.noise:
            se3_noise = torch.randn(n_views, 6, device=opt.device) * opt.camera.noise
            # pose_noise = se3_to_SE3(noise) (model/bat.py:32-36): the exp kernel with identity base
            eye = torch.eye(3, 4, device=opt.device)
            noise = ops.train_pose(se3_noise.float(), None, eye)
            self.graph.pose_noise = torch.nn.Parameter(noise.detach(), requires_grad=False)
        self.graph.se3_refine = torch.nn.Embedding(n_views, 6).to(opt.device)
        torch.nn.init.zeros_(self.graph.se3_refine.weight)
        self._install_reg_weight_provider()

    def _install_reg_weight_provider(self):
        """Graph.render_rays asks this for the regularisers' loss weights of the iteration being rendered: the device-resident
        weight vector while a hipGraph is captured / replayed (ops.LOSS_WEIGHTS_STATIC, poked in front of every replay), the
        host values of fused_loss_weights otherwise -- exactly what summarize_loss will multiply the sums by."""
        import weakref
        me = weakref.ref(self)

        def provider(opt):
            m = me()
            if m is None or not _has(opt, "loss_weight") or opt.loss_weight.render is None:
                return None
            if ops.LOSS_WEIGHTS_STATIC is not None:
                return ops.LOSS_WEIGHTS_STATIC[1:4]
            try:
                return tuple(m.fused_loss_weights(opt)[1:])
            except (AttributeError, KeyError, TypeError):
                return None
        self.graph.__dict__["reg_weight_provider"] = provider

    def setup_optimizer(self, opt):
        nerf = self.graph.nerf
        self.optim = nerf._get_optimizer(opt)
        nerf.get_current_optimizer = lambda: self.optim

        def register(o):
            self.optim = o
        nerf.register_new_optimizer = register
        if opt.optim.pose_algo == "Adam" and str(opt.device).startswith("cuda"):
            # the pose step through the same one-launch kernel as the factors' (torch's fused Adam is two: the device-side
            # step counter, then the update); Adam's defaults, Adam's state keys (optim.VMAdam)
            from ..optim import VMAdam
            self.optim_pose = VMAdam([dict(params=self.graph.se3_refine.parameters(), lr=opt.optim.lr_pose)])
        else:
            algo = getattr(torch.optim, opt.optim.pose_algo)
            self.optim_pose = algo([dict(params=self.graph.se3_refine.parameters(), lr=opt.optim.lr_pose)])
        self.sched_pose = None
        if opt.optim.sched_pose:
            assert opt.optim.sched_pose.type == "ExponentialLR"
            gamma = (opt.optim.lr_pose_end / opt.optim.lr_pose) ** (1.0 / opt.max_iter)
            self.sched_pose = torch.optim.lr_scheduler.ExponentialLR(self.optim_pose, gamma=gamma)

    # ---- ray-sharded data parallelism (the reference is single-GPU, options.py:126) -------------------
    def enable_data_parallel(self, opt, rank, world, shard="pixel", group=None, force=False):
        """One process per GPU, torch.distributed already initialised ("nccl" = RCCL over xGMI).  The iteration's rays
        are split over the ranks (Graph.ray_shard: "pixel" | "view" | "offset"), the scene gradients are SUM-all-reduced
        inside the renderer's backward (ops.set_data_parallel), the pose gradients after it (train_iteration), the
        optimizer steps are replicated.  Host draws must come from identically seeded generators on every rank."""
        assert shard in ("pixel", "view", "offset"), shard
        ops.set_data_parallel(world, group=group, force=force)
        self.dp = Opt(rank=int(rank), world=int(world), shard=shard, group=group, force=bool(force))
        g = self.graph
        g.ray_shard = (shard, int(rank), int(world))
        g.global_views = int(g.se3_refine.weight.shape[0]) if shard == "view" else None
        self.render_loss_scale = 1.0 / world  # the per-iteration value (unequal shards) travels in var.dp_render_scale
        self._dp_local_views = None

    def local_views(self, var_all):
        """The views this rank renders, as a `var`-layout dict: all of them except under "view" sharding, where it is
        the views rank::world (sliced once per source dict and remembered: the per-view tensors are static)."""
        dp = getattr(self, "dp", None)
        if dp is None or dp.shard != "view":
            return var_all
        memo = self._dp_local_views
        if memo is None or memo[0] is not var_all:
            n = len(var_all.idx)
            from ..dist import shard_indices
            sel = torch.tensor(shard_indices(n, dp.rank, dp.world), device=var_all.idx.device)
            local = Opt({k: (v[sel] if torch.is_tensor(v) and v.shape[:1] == (n,) else v) for k, v in dict(var_all).items()})
            self._dp_local_views = memo = (var_all, local)
        return memo[1]

    def reduce_pose_gradients(self):
        """what left the renderer through the rays: the se(3) refinements' gradient, summed over the ranks (2.4 KB)"""
        dp = getattr(self, "dp", None)
        if dp is not None and (dp.world > 1 or dp.force) and not ops._DP.get("no_collectives"):
            from ..dist import allreduce_gradients
            allreduce_gradients([self.graph.se3_refine.weight], dp.world, group=dp.group, force=dp.force)

    def summarize_loss(self, opt, var, loss):
        """model/tensorf.py:31-47 (linear weights; the finiteness asserts would force a host sync per
        iteration and are left to the caller)."""
        total = 0.0
        # ray-sharded data parallelism: only the photometric term is a mean over the (global) ray batch
        render_scale = float(var.get("dp_render_scale", None) or getattr(self, "render_loss_scale", 1.0))
        # device-side guard: NaN pose / render / loss leave a bit in the status word (read by check_finite) -- in the launch
        # that forms the weighted sum
        guard = None
        if not (_has(opt, "finite_checks") and not opt.finite_checks):
            guard = [(var.get("current_pose"), ops.FINITE_POSE), (var.get("rgb"), ops.FINITE_RENDER)]
        fused = self._summarize_fused(opt, loss, render_scale, guard)
        if fused is not None:
            loss.update(all=fused)
            return loss
        for key in loss:
            assert key in opt.loss_weight, f"loss {key} not in opt.loss_weight"
            if key == "L1":
                first = opt.train_schedule.update_alphamask_iters[0]
                w = float(opt.loss_weight.L1.rest if self.it > first else opt.loss_weight.L1.init)
                total = total + w * float(getattr(self, "_reg_scale", 1.0)) * loss["L1"]
            elif opt.loss_weight[key] is not None:
                w = float(opt.loss_weight[key])
                if key == "render":
                    w *= render_scale
                else:
                    w *= float(getattr(self, "_reg_scale", 1.0))
                if w != 0.0:
                    total = total + w * loss[key]
        loss.update(all=total)
        return loss

    def _summarize_fused(self, opt, loss, render_scale, guard=None):
        """The same weighted sum, same order of terms, as ONE launch each way (ops.loss_sum) when the terms are the
        four of the BAT yamls and live on the GPU; None otherwise (the generic loop below then does it)."""
        keys = list(loss.keys())
        if not (len(keys) >= 4 and keys[:4] == ["render", "L1", "TV_density", "TV_color"]):
            return None
        if not (torch.is_tensor(loss.render) and loss.render.is_cuda and opt.loss_weight.render is not None):
            return None
        for key in keys[4:]:   # e.g. LLFF's TV_depth: fine as long as its weight keeps it out of the sum
            assert key in opt.loss_weight, f"loss {key} not in opt.loss_weight"
            if key == "all" or (opt.loss_weight[key] is not None and float(opt.loss_weight[key]) != 0.0):
                return None
        tf = self.graph.nerf.tensorf
        return ops.loss_sum(loss.render, tf._reg(), *self.fused_loss_weights(opt, render_scale), check_items=guard,
                            loss_bit=ops.FINITE_LOSS if guard is not None else 0)

    def fused_loss_weights(self, opt, render_scale=None):
        """(w_render, w_L1, w_TV_density, w_TV_color) of this iteration's weighted loss sum (model/tensorf.py:31-47)"""
        if render_scale is None:
            render_scale = float(getattr(self, "render_loss_scale", 1.0))
        first = opt.train_schedule.update_alphamask_iters[0]
        w_l1 = float(opt.loss_weight.L1.rest if self.it > first else opt.loss_weight.L1.init)
        w_tvd = float(opt.loss_weight.TV_density or 0.0) * float(getattr(self.graph.tvloss, "TVLoss_weight", 1))
        w_tvc = float(opt.loss_weight.TV_color or 0.0) * float(getattr(self.graph.tvloss, "TVLoss_weight", 1))
        rs = float(getattr(self, "_reg_scale", 1.0))   # 0 for the later ray groups of a split iteration
        return float(opt.loss_weight.render) * render_scale, w_l1 * rs, w_tvd * rs, w_tvc * rs

    def train_iteration(self, opt, var):
        """One optimisation step (model/bat.py:96-116 around model/base.py:154-172): the host-side engine in
        begin_iteration / end_iteration, everything that touches the GPU in forward_backward (tests/test_engine_trace.py
        drives the two host halves against a trace of the reference's own loop, without a GPU)."""
        self.begin_iteration(opt)
        loss = self.forward_backward(opt, var)
        self.end_iteration(opt)
        return loss

    def begin_iteration(self, opt):
        """model/bat.py:97-100 (linear warm-up of the pose learning rate) and model/base.py:118 (reproduced quirk, SURVEY
        App. B-15: the scene gradients are zeroed at the top of EVERY iteration, so opt.optim.grad_accum_iter > 1 only thins
        out the optimizer steps)."""
        self.graph.it = self.it
        if opt.optim.warmup_pose:
            pg = self.optim_pose.param_groups[0]
            pg["lr_orig"] = pg["lr"]
            pg["lr"] *= min(1, self.it / opt.optim.warmup_pose)
        self.optim.zero_grad()

    def forward_backward(self, opt, var):
        """model/base.py:157-162: forward, losses, weighted sum, backward"""
        g = self.graph
        ops.PROFILING = bool(_has(opt, "profiling") and opt.profiling)  # roctx ranges named as model/base.py:119-153
        groups = self.tape_groups(opt)
        if groups > 1:
            loss = self._forward_backward_in_groups(opt, var, groups)
        else:
            with ops.prof_range("graph.forward"):
                var = g.forward(opt, var, mode="train")
            with ops.prof_range("graph.compute_loss"):
                loss = g.compute_loss(opt, var, mode="train")
            with ops.prof_range("summarize_loss"):
                loss = self.summarize_loss(opt, var, loss)
            with ops.prof_range("loss.all.backward()"):
                ops.backward(loss.all, gradient=self._backward_seed(loss.all))  # (a cached ones tensor: no fill launch per iteration)
        self.reduce_pose_gradients()
        return loss

    def end_iteration(self, opt):
        """model/base.py:163-169 (scene step, the counter) and model/bat.py:103-114 (pose step on (it + 1) % period -- the
        counter has been incremented by then, SURVEY App. B-19 --, warm-up undone, pose scheduler, progress)."""
        if (not _has(opt.optim, "grad_accum_iter")) or (self.it % opt.optim.grad_accum_iter) == 0:
            with ops.prof_range("optim.step"):
                # (nothing has touched the gradients since the backward: the appearance factors may be stepped beside the
                #  density backward, optim.VMAdam.step)
                if getattr(self.optim, "supports_early_step", False):
                    self.optim.step(early_ok=True)
                else:
                    self.optim.step()
            self.optim.zero_grad()
        self.it += 1
        if (not _has(opt.optim, "pose_grad_accum_iter")) or (self.it % opt.optim.pose_grad_accum_iter) == 0:
            self.optim_pose.step()
            self.optim_pose.zero_grad()
        if opt.optim.warmup_pose:
            self.optim_pose.param_groups[0]["lr"] = self.optim_pose.param_groups[0]["lr_orig"]
        if self.sched_pose is not None:
            self.sched_pose.step()
        self.graph.nerf.set_progress(self.it / opt.max_iter)

    # ---- an iteration whose autograd tape would not fit a memory budget: forward + backward over ray groups -------------
    def tape_groups(self, opt):
        """How many ray groups this iteration's forward + backward is split into so that the tape of the fused appearance
        chain (1 920 bytes per shaded sample, sized for rays x samples) stays under opt.tape_budget_gb (default 16; the
        environment variable JT_TAPE_BUDGET_GB overrides; 0 = never split).  1 for every training configuration of the yamls
        (2 048 - 20 480 rays: 0.6 - 11 GB); BASELINE.json configs[3] on ONE GPU (62 500 rays x 1 000 samples: 120 GB worst case,
        86 GB sized by the shaded count) runs as 8 groups of 7 800 rays."""
        import os
        g = self.graph
        if g.ray_shard is not None or opt.nerf.ray_sampling_strategy != "all_view_rand_grid" or g.lattice_override is not None:
            return 1
        budget = float(os.environ.get("JT_TAPE_BUDGET_GB", opt.get("tape_budget_gb", 16.0) or 0.0))
        if budget <= 0:
            return 1
        need = float(opt.nerf.n_rays) * float(g.nerf.n_samples) * 1920.0
        return max(1, int(math.ceil(need / (budget * 2 ** 30))))

    def _forward_backward_in_groups(self, opt, var, groups):
        """The iteration's lattice rendered as `groups` pixel shards (lattice points k::groups of every view -- the partition
        ray-sharded data parallelism uses, Graph.ray_shard), each with its own forward, loss and backward: the photometric term
        is a mean over ALL rays, so shard k contributes its own mean x its share of the rays (var.dp_render_scale), and the
        regularisers, which do not depend on the rays, enter with the first shard only.  Gradients accumulate in .grad as
        autograd always does; a shard's tape is gone before the next shard's forward.  Every shard consumes the SAME host /
        device draws (lattice offsets, blur scale, white-background coin, jitter stream): the random streams are rewound to the
        iteration's start before each shard and left where ONE forward leaves them."""
        g = self.graph
        state = (np.random.get_state(), torch.get_rng_state(), torch.cuda.get_rng_state(torch.device(opt.device)))
        total = None
        out = None
        g._group_rays = []
        try:
            for k in range(groups):
                self._reg_scale = 1.0 if k == 0 else 0.0   # (fused_loss_weights / summarize_loss: regularisers with shard 0 only)
                if k > 0:
                    np.random.set_state(state[0])
                    torch.set_rng_state(state[1])
                    torch.cuda.set_rng_state(state[2], torch.device(opt.device))
                g.ray_shard = ("pixel", k, groups)
                v = g.forward(opt, Opt(dict(var)), mode="train")
                loss = g.compute_loss(opt, v, mode="train")
                loss = self.summarize_loss(opt, v, loss)
                ops.backward(loss.all, gradient=self._backward_seed(loss.all))
                total = loss.all.detach() if total is None else total + loss.all.detach()
                g._group_rays.append(int(v.rgb.shape[0] * v.rgb.shape[1]))
                if out is None:
                    out = loss   # (the logged terms are shard 0's; the iteration's total is in `all`)
                del v, loss
        finally:
            g.ray_shard = None
            self._reg_scale = 1.0
        out.update(all=total)
        return out

    def _backward_seed(self, t):
        key = (t.shape, t.dtype, str(t.device))
        memo = self.__dict__.setdefault("_seed_memo", {})
        if key not in memo:
            memo[key] = ops.register_unit_seed(torch.ones_like(t))
        return memo[key]

    def after_iteration(self, opt, it=None):
        """The part of nerf.Model.train's loop body that follows train_iteration (model/nerf.py:258-260).  The
        reference calls update_schedule with self.it AFTER base.train_iteration has incremented it:
        the grid therefore grows right after the training step with index upsample_iters[k] - 1."""
        self.graph.nerf.update_schedule(opt, self.it if it is None else it)

    def before_iteration(self, opt, it=None):
        """The per-iteration schedule glue at the top of nerf.Model.train's loop body (model/nerf.py:171-205): 2-D blur
        cache refresh is select_supervision's; here the ray count, the pose / scene gradient-accumulation periods, the
        pose resets and the sampling-strategy switches."""
        it = self.it if it is None else it
        ts = opt.train_schedule
        if _has(ts, "change_n_rays_after_n_iters"):
            opt.nerf.n_rays = ts.n_rays_init if it < ts.change_n_rays_after_n_iters else ts.n_rays_rest
        if _has(ts, "change_n_AccumPoseGrad_after_n_iters"):
            opt.optim.pose_grad_accum_iter = (ts.n_AccumPoseGrad_init if it < ts.change_n_AccumPoseGrad_after_n_iters
                                              else ts.n_AccumPoseGrad_rest)
        if _has(ts, "change_n_AccumGrad_after_n_iters"):
            opt.optim.grad_accum_iter = (ts.n_AccumGrad_init if it < ts.change_n_AccumGrad_after_n_iters
                                         else ts.n_AccumGrad_rest)
        if (_has(ts, "reset_pose_on_iter") and ts.reset_pose_on_iter == it) or \
                (_has(ts, "reset_pose_on_iters") and it in ts.reset_pose_on_iters):
            self.interrupt_pose(opt)
        if _has(ts, "all_view_sample_after_n_iters") and it == ts.all_view_sample_after_n_iters:
            opt.nerf.ray_sampling_strategy = "all_view_rand_rays"
        if _has(ts, "single_view_sample_after_n_iters") and it == ts.single_view_sample_after_n_iters:
            opt.nerf.ray_sampling_strategy = "single_view_rand_rays"

    @torch.no_grad()
    def interrupt_pose(self, opt):
        """model/bat.py:78-81"""
        self.graph.se3_refine.weight.mul_(0.0)

    def freeze_poses(self, opt):
        """model/bat.py:82-83 sets `requires_grad` on the Embedding MODULE (an attribute nobody reads); the evident
        intent -- no pose gradients during evaluation -- is applied to its weight as well."""
        self.graph.se3_refine.requires_grad = False
        self.graph.se3_refine.weight.requires_grad_(False)

    def unfreeze_poses(self, opt):
        self.graph.se3_refine.requires_grad = True
        self.graph.se3_refine.weight.requires_grad_(True)

    def freeze_scene(self, opt):
        self.graph.nerf.freeze_scene(opt)

    def unfreeze_scene(self, opt):
        self.graph.nerf.unfreeze_scene(opt)

    def save_param_state(self):
        """model/tensorf.py:77-83"""
        return self.graph.save_param_state()

    def load_param_state(self, opt, ckpt):
        self.graph.load_param_state(opt, ckpt)

    def save_checkpoint(self, opt, ep=0, it=0, latest=False):
        """model/base.py:235-238, file format of util.py:162-184"""
        from ..checkpoint import save_checkpoint
        return save_checkpoint(opt, self, ep=ep, it=it, latest=latest)

    def restore_checkpoint(self, opt):
        """model/base.py:60-71"""
        from ..checkpoint import restore_checkpoint
        ep = it = None
        if opt.get("resume", False):
            ep, it = restore_checkpoint(opt, self, resume=opt.resume)
        elif opt.get("load", None) is not None:
            ep, it = restore_checkpoint(opt, self, load_name=opt.load)
        self.epoch_start, self.iter_start = ep or 0, it or 0

    def setup_visualizer(self, opt):
        """model/base.py:73-88: TensorBoard / visdom / wandb writers are the reference engine's (out of scope, SURVEY
        section 2); a run with opt.tb / opt.visdom set says so instead of silently dropping them."""
        if opt.get("tb", False) or opt.get("visdom", False):
            print("joint_tensorf_amd: TensorBoard / visdom logging is not part of this build (scalars are returned by "
                  "train_iteration and printed every opt.freq.scalar iterations)")

    @torch.no_grad()
    def validate(self, opt, ep=None):
        """model/bat.py:118-122: the Procrustes alignment of the current poses, which eval-mode get_pose needs (the
        image-space validation render and its logging are the reference engine's)."""
        pose, pose_GT = self.get_all_training_poses(opt)
        _, self.graph.sim3 = self.prealign_cameras(opt, pose, pose_GT)

    def train(self, opt):
        """nerf.Model.train (model/nerf.py:150-278) without the logging / visualisation side effects: schedule glue,
        2-D supervision choice, train_iteration, update_schedule, periodic validate and checkpoints.  Returns the last
        loss dict."""
        import time
        self.graph.train()
        self.ep = 0
        if self.iter_start == 0:
            self.validate(opt, 0)
        if _has(opt, "view_sampling_n_groups"):
            ng = opt.view_sampling_n_groups
            all_views = torch.randperm(self.n_train_views, device=opt.device)
            self.group_idx = [all_views[i::ng] for i in range(ng)]
        freq = opt.get("freq", Opt())
        f_scalar, f_val, f_ckpt = (int(freq.get(k, 0) or 0) for k in ("scalar", "val", "ckpt"))
        data_all = self.local_views(self.train_data.all)  # this rank's views (all of them unless view-sharded)
        images_all = data_all.image
        loss, t0 = None, time.time()
        stepper = None
        want_graph = bool(opt.train_graph) if _has(opt, "train_graph") else True
        if want_graph and str(opt.device).startswith("cuda"):
            # every steady-state iteration replayed from a hipGraph (graphed.GraphedTrainStep): one launch per iteration.
            # On by default (`train_graph: false` turns it off): a trained scene shades a few per cent of its samples and
            # the eager step is bound by the host's ~60 launches (1.8 ms eager against 1.0 ms replayed on the converged
            # synthetic scene); iterations the stepper cannot capture (sharded ranks, first iterations of a stage) run eager
            from ..graphed import GraphedTrainStep
            # (per grid stage the stepper measures whether the replay or the eager launch is the faster one and stays with
            #  it: the late, GPU-bound stages end up eager, with the weight-gradient GEMMs forked beside the scatter)
            stepper = GraphedTrainStep(self)
        self.train_stepper = stepper  # its .stats / .decisions: what ran replayed, what ran eager and why
        for it in range(int(opt.max_iter)):
            self.it = it
            if _has(opt, "early_stop_iter") and opt.early_stop_iter == it:
                break
            self.graph.it = it
            if it < self.iter_start:
                # a resumed run passes through the skipped iterations' cache refreshes (model/nerf.py:172-176 sits in
                # front of the `continue`): the 2-D supervision cache is the one the uninterrupted run would hold
                if it % 500 == 0 and _has(opt, "blur_2d") and opt.blur_2d:
                    self._refresh_supervision(opt, images_all)
                continue
            self.before_iteration(opt, it)
            train_images, train_edge_masks, sc = self.select_supervision(opt, images_all)
            var = Opt(dict(data_all))
            var.image, var.train_edge_masks = train_images, train_edge_masks
            if _has(opt, "sync_2d_3d_scales") and opt.sync_2d_3d_scales:
                var.scale = sc
                self.graph.scale = sc
            view_idx = None
            if opt.nerf.ray_sampling_strategy == "single_view_rand_rays":
                self.view_index = self.graph.view_index = it % self.n_train_views
                view_idx = [self.view_index]
            elif _has(opt, "view_sampling_n_groups"):
                view_idx = self.group_idx[it % opt.view_sampling_n_groups]
            if view_idx is not None:
                for k in ("image", "pose", "intr_inv", "idx", "intr"):
                    var[k] = var[k][view_idx]
            loss = stepper.train_iteration(opt, var) if stepper is not None else self.train_iteration(opt, var)
            if _has(opt, "finite_checks_every_iteration") and opt.finite_checks_every_iteration:
                self.check_finite(opt, loss)
            self.after_iteration(opt)  # self.it is it + 1 here, as in the reference
            if f_scalar and self.it % f_scalar == 0:
                self.check_finite(opt, loss)  # one host read per opt.freq.scalar iterations (model/tensorf.py:43-44)
                print("it %d  loss %.6f  (%.1f it/s)" % (self.it, float(loss.all.detach()), (it + 1) / (time.time() - t0)))
            if f_val and self.it % f_val == 0:
                self.validate(opt, self.it)
            if f_ckpt and self.it % f_ckpt == 0 and _has(opt, "output_path"):
                self.save_checkpoint(opt, ep=None, it=self.it)
        self.check_finite(opt, loss)
        if _has(opt, "output_path") and opt.output_path:
            self.save_checkpoint(opt, ep=None, it=self.it, latest=True)
        return loss

    def check_finite(self, opt, loss=None):
        """The reference raises on a NaN pose (model/tensorf.py:147-151) and asserts finite loss terms
        (model/tensorf.py:43-44) every iteration, each a device->host read.  Here every training iteration leaves its
        verdict in a device-side status word (ops.finite_check inside summarize_loss: pose, rendered colours, total loss);
        this is the one host read, taken every opt.freq.scalar iterations by train() or whenever a caller asks (set
        opt.finite_checks_every_iteration to get the reference's timing).  Raises FloatingPointError naming the culprit."""
        bits = ops.read_status(opt.device)
        if bits:
            what = [n for b, n in ((ops.FINITE_POSE, "camera pose (se3_refine / pose composition)"),
                                   (ops.FINITE_RENDER, "rendered colours"), (ops.FINITE_LOSS, "loss"),
                                   (ops.FINITE_GRAD, "gradient scatter (a non-finite / out-of-range addend or an overflowed "
                                                     "fixed-point sum in the pose or factor gradients)")) if bits & b]
            raise FloatingPointError("non-finite values since the last check in: %s (iteration %d)" % ("; ".join(what), self.it))

    def generate_videos_synthesis(self, opt, eps=1e-10, it=None):
        """model/nerf.py:574-640 writes novel-view videos through ffmpeg / wandb: reference engine, out of scope.  Kept
        as a callable no-op so that the train_3d.py sequence runs through."""
        print("joint_tensorf_amd: generate_videos_synthesis is not part of this build (evaluate_full renders the test views)")
        return None

    # ---- evaluation ---------------------------------------------------------------------------
    @torch.no_grad()
    def get_all_training_poses(self, opt, pose_GT=None):
        """model/bat.py:197-210: (optimised poses, ground-truth poses) of all training views.  `pose_GT` [N,3,4]
        is what the reference reads from its dataset object (self.train_data.get_all_camera_poses)."""
        g = self.graph
        if pose_GT is None:
            pose_GT = self.train_data.get_all_camera_poses(opt)
        pose_GT = pose_GT.to(opt.device, dtype=torch.float32)
        if opt.data.dataset == "blender":
            noise = g.pose_noise if opt.camera.noise else None
            pose = ops.train_pose(g.se3_refine.weight.detach(), noise, pose_GT)
        else:
            pose = ops.train_pose(g.se3_refine.weight.detach(), None, g.pose_eye)
        return pose, pose_GT

    @staticmethod
    def _camera_centers(pose):
        R, t = pose[..., :3], pose[..., 3:]
        return (-R.transpose(-1, -2) @ t)[..., 0]

    @staticmethod
    def procrustes_analysis(X0, X1):
        """camera.py:349-366.  The 3x3 SVD runs in double on the host with the routine the reference calls."""
        t0, t1 = X0.mean(dim=0, keepdim=True), X1.mean(dim=0, keepdim=True)
        X0c, X1c = X0 - t0, X1 - t1
        s0 = (X0c ** 2).sum(dim=-1).mean().sqrt()
        s1 = (X1c ** 2).sum(dim=-1).mean().sqrt()
        U, S, V = torch.svd(((X0c / s0).t() @ (X1c / s1)).double().cpu(), some=True)
        R = (U @ V.t()).float()
        if R.det() < 0:
            R[2] *= -1
        return Opt(t0=t0[0], t1=t1[0], s0=s0, s1=s1, R=R.to(X0.device))

    @torch.no_grad()
    def prealign_cameras(self, opt, pose, pose_GT):
        """model/bat.py:212-228."""
        center_pred, center_GT = self._camera_centers(pose), self._camera_centers(pose_GT)
        try:
            sim3 = self.procrustes_analysis(center_GT, center_pred)
        except Exception:
            print("warning: SVD did not converge...")
            sim3 = Opt(t0=0, t1=0, s0=1, s1=1, R=torch.eye(3, device=opt.device, dtype=torch.float32))
        center_aligned = (center_pred - sim3.t1) / sim3.s1 @ sim3.R.t() * sim3.s0 + sim3.t0
        R_aligned = pose[..., :3] @ sim3.R.t()
        t_aligned = (-R_aligned @ center_aligned[..., None])[..., 0]
        return torch.cat([R_aligned, t_aligned[..., None]], -1), sim3

    @torch.no_grad()
    def evaluate_camera_alignment(self, opt, pose_aligned, pose_GT, eps=1e-7):
        """model/bat.py:230-238 (+ camera.rotation_distance, camera.py:342-347)."""
        Rd = pose_aligned[..., :3] @ pose_GT[..., :3].transpose(-2, -1)
        trace = Rd[..., 0, 0] + Rd[..., 1, 1] + Rd[..., 2, 2]
        R_error = ((trace - 1) / 2).clamp(-1 + eps, 1 - eps).acos()
        t_error = (pose_aligned[..., 3] - pose_GT[..., 3]).norm(dim=-1)
        return Opt(R=R_error, t=t_error)

    @torch.enable_grad()
    def evaluate_test_time_photometric_optim(self, opt, var):
        """model/bat.py:265-292: a fresh se3 [1,6] absorbs the remaining pose error of one held-out view.
        The scene is frozen for the duration, so the renderer's backward takes its pose-only route (no factor
        or weight gradients are formed; the reference forms and discards them)."""
        g = self.graph
        if _has(opt.optim, "test_graph") and opt.optim.test_graph:
            # every iteration replayed from a hipGraph (graphed.GraphedTestOptim); same state transitions and draws
            if getattr(self, "_test_optim_graph", None) is None:
                from ..graphed import GraphedTestOptim
                self._test_optim_graph = GraphedTestOptim(self)
            return self._test_optim_graph.run(opt, var)
        var.se3_refine_test = torch.nn.Parameter(torch.zeros(1, 6, device=opt.device))
        kw = dict(fused=True) if str(opt.device).startswith("cuda") else {}
        optim_pose = torch.optim.Adam([dict(params=[var.se3_refine_test], lr=opt.optim.lr_pose)], **kw)
        gamma = (opt.optim.lr_pose_test_end / opt.optim.lr_pose_test) ** (1.0 / opt.optim.test_iter)
        sched_pose = torch.optim.lr_scheduler.ExponentialLR(optim_pose, gamma=gamma)
        frozen = [p for p in g.parameters() if p.requires_grad]
        for p in frozen:
            p.requires_grad_(False)
        eye = torch.eye(3, 4, device=opt.device)
        with torch.no_grad():
            var.pose_aligned = g.aligned_pose(var.pose).contiguous()
        try:
            for it in range(opt.optim.test_iter):
                g.nerf.set_test_time_progress(it / opt.optim.test_iter)
                optim_pose.zero_grad()
                var.pose_refine_test = ops.train_pose(var.se3_refine_test, None, eye)  # se3_to_SE3 (camera.py:81-99)
                var = g.forward(opt, var, mode="test-optim")
                loss = g.compute_loss(opt, var, mode="test-optim")
                loss = self.summarize_loss(opt, var, loss)
                ops.backward(loss.all)
                optim_pose.step()
                sched_pose.step()
        finally:
            for p in frozen:
                p.requires_grad_(True)
        # NOTE (reproduced): var.pose_refine_test is refreshed at the top of an iteration only, so the eval render
        # that follows sees the refinement from before the last Adam step (model/bat.py:284, model/nerf.py:539)
        return var

    @torch.enable_grad()
    def evaluate_test_time_photometric_optim_batched(self, opt, views):
        """model/bat.py:265-292 for SEVERAL held-out views at once.  The reference optimises the 200 test
        views one after the other, 400 - 600 iterations each: 80 000 launch-bound iterations per evaluation.  The per-view
        problems are independent -- a view's own se(3) vector, its own Adam moments, its own lattice draws, a photometric loss
        over its own rays -- so V of them run as ONE iteration: a [V, 6] parameter under one Adam (element-wise: every row is
        the serial run's state), every view's rays on ITS lattice (ragged: the lattice size depends on the offsets)
        concatenated into one ray batch through the pose-only backward, one photometric mean PER VIEW, summed.  The host
        draws are taken up front view by view, iteration by iteration (two lattice offsets and the density-blur scale): every
        view sees the draws of serial OPTIMISATIONS run back to back, and the random stream ends where those leave it.  (The
        reference's evaluate_full interleaves an eval render after each view's optimisation, and that render draws nothing
        while the blur is off -- the only case this routine takes; with a blur possible it falls back to the serial loop.)
        Returns the views (dicts) with se3_refine_test [1, 6] / pose_refine_test as the serial
        routine leaves them -- pose_refine_test from the top of the LAST iteration (the reference's quirk, model/bat.py:284).
        Falls back to the serial routine when a factor blur is active at test time (per-view blur scales cannot share a
        render) or for a single view."""
        g = self.graph
        V, T, dev = len(views), int(opt.optim.test_iter), opt.device
        # could ANY call of resolve_blur during the optimisation or the eval renders come back with a kernel?  The schedule's
        # value alone ("vis"), the density value under the LARGEST random scale (a pool value above 1 lifts a density parameter
        # that is below the 1e-3 cut-off on its own over it), LLFF's test-time kernel schedule
        blur_possible = g.resolve_blur(opt, "vis")[2] is not None or (
            opt.data.dataset == "llff" and max(float(v) for v in opt.optim.test_kernel_schedule) >= 0.001)
        if not blur_possible and opt.model in ("bat", "bat_hip") and opt.c2f_mode != "None" and \
                _has(opt, "c2f_random_density_blur") and opt.c2f_random_density_blur:
            pd = interp_schedule(g.nerf.progress_host, opt.c2f_schedule_density)
            blur_possible = pd * max(float(v) for v in opt.c2f_random_density_scale_pool) >= 0.001
        if V <= 1 or blur_possible or opt.nerf.ray_sampling_strategy != "all_view_rand_grid" or \
                (_has(opt.optim, "test_fused") and opt.optim.test_fused):
            return [self.evaluate_test_time_photometric_optim(opt, v) for v in views]
        step = g.lattice_step(opt, 1)
        draw_scale = _has(opt, "c2f_random_density_blur") and opt.c2f_random_density_blur and opt.model in ("bat", "bat_hip") \
            and opt.c2f_mode != "None"
        draws = []
        for v in range(V):      # the serial run's order of host draws: Graph.forward (ox, oy), then Graph.resolve_blur (scale)
            row = []
            for it in range(T):
                ox, oy = np.random.randint(step), np.random.randint(step)
                if draw_scale:
                    np.random.choice(opt.c2f_random_density_scale_pool)
                row.append((ox, oy))
            draws.append(row)
        se3 = torch.nn.Parameter(torch.zeros(V, 6, device=dev))
        kw = dict(fused=True) if str(dev).startswith("cuda") else {}
        optim_pose = torch.optim.Adam([dict(params=[se3], lr=opt.optim.lr_pose)], **kw)
        gamma = (opt.optim.lr_pose_test_end / opt.optim.lr_pose_test) ** (1.0 / opt.optim.test_iter)
        sched_pose = torch.optim.lr_scheduler.ExponentialLR(optim_pose, gamma=gamma)
        frozen = [p for p in g.parameters() if p.requires_grad]
        for p in frozen:
            p.requires_grad_(False)
        eye = torch.eye(3, 4, device=dev)
        bv = Opt(idx=torch.arange(V, device=dev), pose=torch.cat([v["pose"].to(dev) for v in views], 0))
        intr = torch.cat([v["intr"].to(dev) for v in views], 0)
        intr_inv = torch.cat([v["intr_inv"].to(dev) for v in views], 0)
        images = torch.cat([v["image"].to(dev) for v in views], 0)      # [V, 3, H, W]
        ndc_near = float(opt.arch.ndc_near_plane) if _has(opt.arch, "ndc_near_plane") else 1.0
        w_render = float(opt.loss_weight.render)
        voff = torch.zeros(V + 1, device=dev, dtype=torch.int32)        # rewritten every iteration (launch arguments: no copy)
        pose_refine = None
        bv.se3_refine_test = se3
        with torch.no_grad():
            bv.pose_aligned = g.aligned_pose(bv.pose).contiguous()       # once: the views do not move, their refinements do
        try:
            for it in range(T):
                g.nerf.set_test_time_progress(it / T)
                optim_pose.zero_grad()
                if it == T - 1:
                    with torch.no_grad():                                  # what the eval render will see (the reference's quirk)
                        pose_refine = ops.train_pose(se3, None, eye)       # se3_to_SE3 of every view (camera.py:81-99)
                pose = g.get_pose(opt, bv, mode="test-optim")              # [V, 3, 4]: exp(se3) composed with the aligned pose
                idxs, offs = [], [0]
                for v in range(V):
                    ox, oy = draws[v][it]
                    nx, ny = len(range(ox, opt.W, step)), len(range(oy, opt.H, step))
                    idxs.append(g._lattice_at(opt, step, ny, nx, ox, oy))
                    offs.append(offs[-1] + nx * ny)
                ridx = torch.cat(idxs, 0)
                for lo in range(0, V + 1, 256):
                    ops.poke_words(voff, offs[lo:lo + 256], offset=lo)
                # every view's rays on its own lattice in ONE launch each way (jt_raygen_*_ragged), one render of all of them,
                # one photometric mean per view (jt_render_loss_views_*): per ray / per view the single-view arithmetic
                center, ray = ops.ray_gen_ragged(pose, intr_inv, intr, ridx, voff, opt.W, ndc=bool(opt.camera.ndc),
                                                 ndc_near=ndc_near)
                g._blur_memo = (None, None, None, None)                    # (this call's blur draw was taken above)
                ret = g.render_rays(opt, center[None], ray[None], mode="test-optim", n_views=1, n_pixels_per_view=center.shape[0])
                per_view = ops.render_loss_views(ret.rgb.view(-1, 3), images, ridx, voff)
                ops.backward(w_render * per_view.sum())
                optim_pose.step()
                sched_pose.step()
        finally:
            for p in frozen:
                p.requires_grad_(True)
        out = []
        for v, view in enumerate(views):
            view = Opt(dict(view))
            view.se3_refine_test = torch.nn.Parameter(se3.detach()[v:v + 1].clone())
            view.pose_refine_test = pose_refine.detach()[v:v + 1].clone()
            view.test_optimised = True
            out.append(view)
        return out

    @torch.no_grad()
    def evaluate_view(self, opt, var, eps=1e-10):
        """The per-view body of nerf.Model.evaluate_full (model/nerf.py:534-548): optional test-time pose
        optimisation, the sliced full-image render and PSNR.  SSIM / LPIPS need packages outside this build."""
        g = self.graph
        g.eval()
        if opt.model in ("barf", "bat", "bat_hip") and opt.optim.test_photo and not var.get("test_optimised", False):
            var = self.evaluate_test_time_photometric_optim(opt, var)   # (already done for views of a batched optimisation)
        var = g.forward(opt, var, mode="eval")
        invdepth = var.depth if opt.camera.ndc else 1 / (var.depth / var.opacity + eps)
        rgb_map = var.rgb.view(-1, opt.H, opt.W, 3).permute(0, 3, 1, 2)
        invdepth_map = invdepth.view(-1, opt.H, opt.W, 1).permute(0, 3, 1, 2)
        psnr = -10 * g.MSE_loss(rgb_map, var.image).log10().item()
        return Opt(psnr=psnr, rgb_map=rgb_map, invdepth_map=invdepth_map, var=var)

    def evaluate_full(self, opt, test_views=None, pose_GT=None, eps=1e-10):
        """model/bat.py:241-263 + model/nerf.py:525-572: camera alignment errors, then PSNR per held-out view.
        `test_views`: an iterable of per-view batches (idx, pose, intr, intr_inv, image); default self.test_loader, as
        in the reference's `evaluate_full(opt)`."""
        if test_views is None:
            test_views = [{k: (v.to(opt.device) if torch.is_tensor(v) else v) for k, v in dict(b).items()}
                          for b in self.test_loader]
        self.graph.eval()
        pose, pose_GT = self.get_all_training_poses(opt, pose_GT)
        pose_aligned, self.graph.sim3 = self.prealign_cameras(opt, pose, pose_GT)
        error = self.evaluate_camera_alignment(opt, pose_aligned, pose_GT)
        # held-out views are independent: with torch.distributed initialised every rank takes every world-th view
        # (no collective on the render path); the per-view PSNRs are gathered at the end
        import torch.distributed as dist
        world, rank = (dist.get_world_size(), dist.get_rank()) if (dist.is_available() and dist.is_initialized()) else (1, 0)
        test_views = list(test_views)
        mine = list(range(rank, len(test_views), world))
        # opt.optim.test_batch = V > 1: the test-time pose optimisation of V views at a time (one iteration serves V views;
        # every view's trajectory is the serial one: evaluate_test_time_photometric_optim_batched)
        V = int(opt.optim.get("test_batch", 0) or 0)
        mine_views = [Opt(dict(test_views[i])) for i in mine]
        if V > 1 and opt.model in ("barf", "bat", "bat_hip") and opt.optim.test_photo:
            done = []
            for a in range(0, len(mine_views), V):
                done += self.evaluate_test_time_photometric_optim_batched(opt, mine_views[a:a + V])
            mine_views = done
        res = [self.evaluate_view(opt, v) for v in mine_views]
        psnr = torch.full((len(test_views),), float("nan"), device=opt.device, dtype=torch.float64)
        for i, r in zip(mine, res):
            psnr[i] = r.psnr
        if world > 1:
            psnr = torch.nan_to_num(psnr, nan=0.0)
            dist.all_reduce(psnr)
        return Opt(R_error=error.R, t_error=error.t, views=res, psnr_per_view=psnr.tolist(),
                   psnr=float(psnr.mean()) if len(test_views) else float("nan"))

    # ---- 2-D blur cache of the supervising images + edge masks ----------------------------------
    @torch.no_grad()
    def process_GT_images(self, opt, images=None):
        """model/nerf.py:57-113: {scale: blurred GT images}; `images` [n,3,H,W] is what the reference reads from
        self.train_data.all.image.  The 201-tap separable blur runs on the factor-blur kernel (ops.blur_images)."""
        if images is None:
            images = self.train_data.all.image
        images = images.to(opt.device, dtype=torch.float32)
        scales = opt.c2f_alternate_2D_scale_pool if opt.c2f_alternate_2D_mode == "sample" else [0.0, 1.0]
        if opt.blur_2d_mode != "uniform-gaussian":
            raise NotImplementedError("blur_2d_mode %s (the BAT yamls use uniform-gaussian)" % opt.blur_2d_mode)
        out = dict()
        for sc in scales:
            blur_param = np.float32(interp_schedule(float(self.it / opt.max_iter), opt.blur_2d_c2f_schedule)) * np.float32(sc)
            kernel_width = float(blur_param * np.float32((opt.W + opt.H) / 2))
            if kernel_width < 0.01:
                out[sc] = images
            else:
                taps = ops.gaussian_taps(kernel_width, opt.blur_2d_c2f_kernel_size, opt.device)
                out[sc] = ops.blur_images(images, taps)
        return out

    @torch.no_grad()
    def get_edge_mask(self, opt, blurred_gt_cached_images):
        """model/nerf.py:115-149: Sobel magnitude of the channel-summed blurred image; hard mask = magnitude above
        hard_edge_mask_mean_thresh x its per-image mean (uint8 [n, H*W]), soft mask = magnitude / per-image max."""
        F = torch.nn.functional
        dev = opt.device
        Kx = torch.tensor([[1., 0., -1.], [2., 0., -2.], [1., 0., -1.]], device=dev)[None, None].expand(1, 3, -1, -1)
        Ky = torch.tensor([[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]], device=dev)[None, None].expand(1, 3, -1, -1)
        masks = dict()
        for sc, img in blurred_gt_cached_images.items():
            n = img.shape[0]
            x = F.pad(img, (1, 1, 1, 1), mode="replicate")
            GG = torch.sqrt(F.conv2d(x, Kx) ** 2 + F.conv2d(x, Ky) ** 2).view(n, opt.H * opt.W)
            if _has(opt, "soft_edge_mask") and opt.soft_edge_mask:
                masks[sc] = GG / GG.max(dim=1, keepdim=True)[0]
            else:
                thresh = opt.hard_edge_mask_mean_thresh if _has(opt, "hard_edge_mask_mean_thresh") else 1.25
                masks[sc] = (GG > GG.mean(dim=1, keepdim=True) * thresh).to(torch.uint8)
        return masks

    def _refresh_supervision(self, opt, images=None):
        """Rebuild the 2-D blur cache and the edge masks (model/nerf.py:172-176).  The new contents are written INTO the
        existing buffers when the shapes allow: the supervising tensors keep their addresses over the 80 refreshes of
        a run, so hipGraphs that read them (graphed.GraphedTrainStep) stay valid."""
        new_img = self.process_GT_images(opt, images)
        new_msk = self.get_edge_mask(opt, new_img)
        src = images if images is not None else self.train_data.all.image
        for name, new in (("blurred_gt_cached_images", new_img), ("blurred_edge_masks", new_msk)):
            old = getattr(self, name, None)
            if isinstance(old, dict) and old.keys() == new.keys() and all(
                    old[k].shape == new[k].shape and old[k].dtype == new[k].dtype and old[k].data_ptr() != src.data_ptr()
                    for k in new):
                for k in new:
                    old[k].copy_(new[k])
            else:
                # (a scale whose blur is below the cut-off comes back as the source images themselves,
                #  model/nerf.py:92-94: it gets a buffer of its own, so that later refreshes can write into it)
                setattr(self, name, {k: (v.clone() if v.data_ptr() == src.data_ptr() else v) for k, v in new.items()})

    def select_supervision(self, opt, images=None):
        """The per-iteration choice of nerf.Model.train (model/nerf.py:172-176, 209-227): refresh the caches every
        500 iterations, then draw the blur scale of this iteration's supervising images; the edge masks come from
        opt.edge_mask_use_scale.  Returns (train_images, train_edge_masks, scale)."""
        if not (_has(opt, "blur_2d") and opt.blur_2d):
            return (images if images is not None else self.train_data.all.image), None, None
        if self.it % 500 == 0 or not hasattr(self, "blurred_gt_cached_images"):
            self._refresh_supervision(opt, images)
        if _has(opt, "c2f_alternate_2D_blur") and opt.c2f_alternate_2D_blur:
            sc = np.random.choice(opt.c2f_alternate_2D_scale_pool)
            train_images = self.blurred_gt_cached_images[sc]
            key = opt.edge_mask_use_scale if _has(opt, "edge_mask_use_scale") else sc
            return train_images, self.blurred_edge_masks[key], sc
        return self.blurred_gt_cached_images[1.0], self.blurred_edge_masks[1.0], 1.0
