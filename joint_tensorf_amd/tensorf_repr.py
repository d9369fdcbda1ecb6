"""Drop-in scene representation: `BAT_VMSplit` with the reference's constructor / forward signature
(model/tensorf.py:375-397 constructs it, model/tensorf.py:246-261 calls it) and the same
state_dict key names, rendered by the HIP kernels behind include/jt_render.h.

What stays host-side torch here is orchestration the reference also does on the host (kernel taps
from a schedule scalar, regularisers over the parameters, grid upsampling); everything per-ray /
per-sample runs in joint_tensorf_amd/csrc.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib, ops

MAT_MODE = ops.MAT_MODE
VEC_MODE = ops.VEC_MODE


def _channel_last_param(t):
    """logical [1,C,H,W] values -> Parameter with the same logical shape stored [H][W][C]."""
    store = t.detach().permute(0, 2, 3, 1).contiguous()
    return torch.nn.Parameter(store.permute(0, 3, 1, 2))


class _RenderMLP(torch.nn.Module):
    """Parameter container with the reference's key names (`mlp.0/2/4.*`, tensorBase.py:101-114).
    The arithmetic of MLPRender_Fea.forward runs in the fused shade kernel."""

    kind = _lib.JT_MLP_FEA

    def __init__(self, inChanel, viewpe, feape, featureC):
        super().__init__()
        self.in_mlpC = 2 * viewpe * 3 + 2 * feape * inChanel + 3 + inChanel
        self.viewpe, self.feape = viewpe, feape
        l1 = torch.nn.Linear(self.in_mlpC, featureC)
        l2 = torch.nn.Linear(featureC, featureC)
        l3 = torch.nn.Linear(featureC, 3)
        self.mlp = torch.nn.Sequential(l1, torch.nn.ReLU(inplace=True), l2, torch.nn.ReLU(inplace=True), l3)
        torch.nn.init.constant_(self.mlp[-1].bias, 0)

    def weights(self):
        m = self.mlp
        return (m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias)


class _RenderMLPWeakView(torch.nn.Module):
    """`layer1/2/3.*` container of MLPRender_Fea_WeakView (tensorBase.py:180-196)."""

    kind = _lib.JT_MLP_WEAKVIEW

    def __init__(self, inChanel, viewpe, feape, featureC):
        super().__init__()
        self.in_mlpC = (2 * feape + 1) * inChanel
        self.mid_mlpC = 2 * viewpe * 3
        self.viewpe, self.feape = viewpe, feape
        self.layer1 = torch.nn.Linear(self.in_mlpC, featureC)
        self.layer2 = torch.nn.Linear(featureC, featureC)
        self.layer3 = torch.nn.Linear(featureC + self.mid_mlpC, 3)
        torch.nn.init.constant_(self.layer3.bias, 0)

    def weights(self):
        return (self.layer1.weight, self.layer1.bias, self.layer2.weight, self.layer2.bias, self.layer3.weight,
                self.layer3.bias)


class TVLoss(torch.nn.Module):
    """tensorBase.py:16-41 (regulariser over the parameters, host-level torch)."""

    def __init__(self, TVLoss_weight=1):
        super().__init__()
        self.TVLoss_weight = TVLoss_weight

    def forward(self, x):
        b, c, h, w = x.shape
        count_h, count_w = c * (h - 1) * w, c * h * (w - 1)
        total = 0
        if count_h > 0:
            total = total + torch.pow(x[:, :, 1:, :] - x[:, :, :h - 1, :], 2).sum() / count_h
        if count_w > 0:
            total = total + torch.pow(x[:, :, :, 1:] - x[:, :, :, :w - 1], 2).sum() / count_w
        return self.TVLoss_weight * 2 * total / b


class BAT_VMSplit(torch.nn.Module):
    """TensoRF vector-matrix scene with coarse-to-fine blurred factors, rendered by HIP kernels."""

    def __init__(self, aabb, gridSize, device, density_n_comp=8, appearance_n_comp=24, app_dim=27,
                 shadingMode="MLP_Fea", alphaMask=None, near_far=(2.0, 6.0), density_shift=-10,
                 alphaMask_thres=0.001, distance_scale=25, rayMarch_weight_thres=0.0001, pos_pe=6, view_pe=6,
                 fea_pe=6, featureC=128, step_ratio=2.0, fea2denseAct="softplus", dtype=torch.float32,
                 volume_init_scale=0.1, volume_init_bias=0.1):
        super().__init__()
        if dtype != torch.float32:
            raise NotImplementedError("the HIP path computes in fp32 (the reference yamls never set half_tensor)")
        self.device = device
        self.dtype = dtype
        self.alphaMask = alphaMask
        self.matMode = [list(m) for m in MAT_MODE]
        self.vecMode = list(VEC_MODE)
        self.kernel_density = None
        self.kernel_color = None
        self.c2f_mode = None
        # test hooks: inject the random draws the reference takes from torch's generators
        self.jitter_override = None
        self.coin_override = None
        self.reset(aabb, gridSize, density_n_comp, appearance_n_comp, app_dim, density_shift, alphaMask_thres,
                   distance_scale, rayMarch_weight_thres, fea2denseAct, near_far, step_ratio, shadingMode, pos_pe,
                   view_pe, fea_pe, featureC, volume_init_scale, volume_init_bias)

    # ---- construction (tensorBase.py:430-488, tensoRF.py:150-169) --------------------------------
    def reset(self, aabb, gridSize, density_n_comp, appearance_n_comp, app_dim, density_shift, alphaMask_thres,
              distance_scale, rayMarch_weight_thres, fea2denseAct, near_far, step_ratio, shadingMode, pos_pe,
              view_pe, fea_pe, featureC, volume_init_scale, volume_init_bias):
        self.density_n_comp = [int(c) for c in density_n_comp]
        self.app_n_comp = [int(c) for c in appearance_n_comp]
        self.app_dim = app_dim
        self.aabb = torch.as_tensor(aabb, dtype=torch.float32).view(2, 3).cpu()
        self.density_shift = density_shift
        self.alphaMask_thres = alphaMask_thres
        self.distance_scale = distance_scale
        self.rayMarch_weight_thres = rayMarch_weight_thres
        self.fea2denseAct = fea2denseAct
        self.near_far = near_far  # same list object as opt.nerf.depth.range (model/tensorf.py:382)
        self.step_ratio = step_ratio
        self.update_stepSize(gridSize)
        self.volume_init_scale = volume_init_scale
        self.volume_init_bias = volume_init_bias
        self.init_svd_volume(gridSize[0], self.device, init_scale=volume_init_scale, init_bias=volume_init_bias)
        self.shadingMode, self.pos_pe, self.view_pe, self.fea_pe, self.featureC = shadingMode, pos_pe, view_pe, fea_pe, featureC
        self.init_render_func(shadingMode, pos_pe, view_pe, fea_pe, featureC, self.device)

    def init_render_func(self, shadingMode, pos_pe, view_pe, fea_pe, featureC, device):
        if shadingMode == "MLP_Fea":
            self.renderModule = _RenderMLP(self.app_dim, view_pe, fea_pe, featureC).to(device)
        elif shadingMode == "MLP_Fea_WeakView":
            self.renderModule = _RenderMLPWeakView(self.app_dim, view_pe, fea_pe, featureC).to(device)
        else:
            raise NotImplementedError("shadingMode %r is not used by the BAT configs (SURVEY.md §2 row 3)" % shadingMode)

    def update_stepSize(self, gridSize):
        g = [int(v) for v in gridSize]
        self.aabbSize = self.aabb[1] - self.aabb[0]
        self.invaabbSize = 2.0 / self.aabbSize
        self.gridSize = torch.LongTensor(g)
        self.units = self.aabbSize / (self.gridSize - 1)
        self.stepSize = torch.mean(self.units) * self.step_ratio
        self.aabbDiag = torch.sqrt(torch.sum(torch.square(self.aabbSize)))
        self.nSamples = int((self.aabbDiag / self.stepSize).item()) + 1

    def init_svd_volume(self, res, device, init_density=True, init_app=True, init_basis=True, init_scale=0.1,
                        init_bias=0.1):
        if init_density:
            self.density_plane, self.density_line = self.init_one_svd(self.density_n_comp, self.gridSize, init_scale,
                                                                     init_bias, device)
        if init_app:
            self.app_plane, self.app_line = self.init_one_svd(self.app_n_comp, self.gridSize, init_scale, init_bias,
                                                             device)
        if init_basis:
            self.basis_mat = torch.nn.Linear(sum(self.app_n_comp), self.app_dim, bias=False).to(device)

    def init_one_svd(self, n_component, gridSize, scale, bias, device):
        g = [int(v) for v in gridSize]
        planes, lines = [], []
        for i in range(3):
            m0, m1 = MAT_MODE[i]
            # same draw order / shapes as tensoRF.py:159-169 (generated on the CPU generator, then moved)
            p = torch.abs(bias + scale * torch.randn((1, n_component[i], g[m1], g[m0]), dtype=torch.float32))
            l = torch.abs(bias + scale * torch.randn((1, n_component[i], g[VEC_MODE[i]], 1), dtype=torch.float32))
            planes.append(_channel_last_param(p.to(device)))
            lines.append(_channel_last_param(l.to(device)))
        return torch.nn.ParameterList(planes), torch.nn.ParameterList(lines)

    # ---- optimisation helpers (tensoRF.py:170-228) ------------------------------------------------
    def freeze_scene(self, opt=None):
        self.basis_mat.weight.requires_grad = False
        self.renderModule.requires_grad_(False)
        for i in range(3):
            for lst in (self.density_plane, self.density_line, self.app_plane, self.app_line):
                lst[i].requires_grad = False

    def unfreeze_scene(self, opt=None):
        self.basis_mat.weight.requires_grad = True
        self.renderModule.requires_grad_(True)
        for i in range(3):
            for lst in (self.density_plane, self.density_line, self.app_plane, self.app_line):
                lst[i].requires_grad = True

    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001):
        return [{"params": self.density_line, "lr": lr_init_spatialxyz},
                {"params": self.density_plane, "lr": lr_init_spatialxyz},
                {"params": self.app_line, "lr": lr_init_spatialxyz},
                {"params": self.app_plane, "lr": lr_init_spatialxyz},
                {"params": self.basis_mat.parameters(), "lr": lr_init_network},
                {"params": self.renderModule.parameters(), "lr": lr_init_network}]

    # The three regularisers are evaluated together by one ABI call (ops.reg_losses -> jt_reg_losses_*): one
    # pass over each factor, and one autograd node instead of ~150 tiny elementwise / reduce launches.  The
    # result is cached per forward call so that density_L1 / TV_loss_density / TV_loss_app share it.
    # `reg_with_tv` = (density, app): whether the TV terms are wanted.  Graph.compute_loss clears a flag when the
    # term's loss weight is zero: the term is then not evaluated (a TV pass reads every texel three times) and
    # reads 0 -- the reference evaluates it for its log line only.
    reg_with_tv = (True, True)

    def _reg_key(self):
        # the versions of one plane of each set and of one line: an optimizer step or a checkpoint load bumps all of
        # them (asked five times per iteration: the whole list cost 130 us of host time)
        return (self.density_plane[0]._version, self.density_plane[2]._version, self.density_line[0]._version,
                self.app_plane[0]._version, torch.is_grad_enabled(), bool(self.reg_with_tv[0]), bool(self.reg_with_tv[1]))

    def _reg(self):
        if not self.density_plane[0].is_cuda:
            raise RuntimeError("joint_tensorf_amd computes regularisers on the GPU only")
        cache = self.__dict__.setdefault("_reg_cache", {})
        key = self._reg_key()
        if cache.get("key") != key:
            cache["key"] = key
            cache["val"] = ops.reg_losses(list(self.density_plane), list(self.density_line), list(self.app_plane),
                                          list(self.app_line), *self.reg_with_tv)
        return cache["val"]

    def density_L1(self):
        """sum_i mean|P_i| + mean|L_i| over the density factors (tensoRF.py:212-216)."""
        return self._reg()[0]

    def TV_loss_density(self, reg):
        """sum_i TVLoss(density_plane_i) * 1e-2 (tensoRF.py:218-222)."""
        w = getattr(reg, "TVLoss_weight", 1)
        return self._reg()[1] if w == 1 else self._reg()[1] * w  # x * 1 is x; the multiply is a launch

    def TV_loss_app(self, reg):
        w = getattr(reg, "TVLoss_weight", 1)
        return self._reg()[2] if w == 1 else self._reg()[2] * w

    # ---- resolution changes (tensoRF.py:274-295) --------------------------------------------------
    @torch.no_grad()
    def up_sampling_VM(self, plane_coef, line_coef, res_target):
        for i in range(3):
            m0, m1 = MAT_MODE[i]
            p = F.interpolate(plane_coef[i].data.contiguous(), size=(res_target[m1], res_target[m0]), mode="bilinear",
                              align_corners=True)
            l = F.interpolate(line_coef[i].data.contiguous(), size=(res_target[VEC_MODE[i]], 1), mode="bilinear",
                              align_corners=True)
            plane_coef[i] = _channel_last_param(p)
            line_coef[i] = _channel_last_param(l)
        return plane_coef, line_coef

    @torch.no_grad()
    def upsample_volume_grid(self, res_target):
        self.app_plane, self.app_line = self.up_sampling_VM(self.app_plane, self.app_line, res_target)
        self.density_plane, self.density_line = self.up_sampling_VM(self.density_plane, self.density_line, res_target)
        self.update_stepSize(res_target)

    # ---- alpha-mask volume and AABB shrink ----------------------------------------
    def _density_cfg(self):
        """RenderCfg for the density-only entry points, in the state the last forward left (blur kernel, mask)."""
        g = self.gridSize.tolist()
        blur = getattr(self, "kernel_density", None) is not None
        m = (lambda i: (g[MAT_MODE[i][0]], g[MAT_MODE[i][1]])) if blur else (lambda i: (g[MAT_MODE[i][1]], g[MAT_MODE[i][0]]))
        return ops.RenderCfg(
            aabb=self.aabb.view(-1).tolist(), plane_hw=[m(i) for i in range(3)],
            line_len=[g[VEC_MODE[i]] for i in range(3)], n_comp_density=self.density_n_comp[0],
            n_comp_app=self.app_n_comp[0], step_size=float(self.stepSize), near_far=tuple(self.near_far),
            distance_scale=self.distance_scale, density_shift=self.density_shift,
            density_act=_lib.JT_ACT_SOFTPLUS if self.fea2denseAct == "softplus" else _lib.JT_ACT_RELU,
            weight_thres=self.rayMarch_weight_thres, n_samples=1, ndc=False, white_bg=False, app_dim=self.app_dim,
            mlp_kind=self.renderModule.kind, mlp_hidden=self.featureC, view_pe=self.view_pe, fea_pe=self.fea_pe,
            alpha_mask=self.alphaMask.kernel_args() if self.alphaMask is not None else None)

    @torch.no_grad()
    def compute_alpha(self, xyz_locs, length=1):
        """BatBase.compute_alpha (batBase.py:27-41): with the blur kernel the last forward used."""
        dP, dL = list(self.density_plane), list(self.density_line)
        if getattr(self, "kernel_density", None) is not None:
            dP, dL, _, _ = ops.blur_factors(self.kernel_density, self.kernel_density, dP, dL, list(self.app_plane),
                                            list(self.app_line))
        return ops.dense_alpha(self._density_cfg(), dP, dL, xyz_locs.reshape(-1, 3), float(length)).view(xyz_locs.shape[:-1])

    @torch.no_grad()
    def getDenseAlpha(self, gridSize=None):
        """tensorBase.py:618-634."""
        gridSize = self.gridSize.tolist() if gridSize is None else [int(v) for v in gridSize]
        dev = self.density_plane[0].device  # the scene box itself is host state (scalars of the kernel calls)
        samples = torch.stack(torch.meshgrid(torch.linspace(0, 1, gridSize[0]), torch.linspace(0, 1, gridSize[1]),
                                             torch.linspace(0, 1, gridSize[2]), indexing="ij"), -1).to(dev)
        aabb = self.aabb.to(dev)
        dense_xyz = aabb[0] * (1 - samples) + aabb[1] * samples
        alpha = self.compute_alpha(dense_xyz.view(-1, 3), self.stepSize).view(*gridSize)
        return alpha, dense_xyz

    @torch.no_grad()
    def updateAlphaMask(self, gridSize=(200, 200, 200)):
        """Occupancy volume of the scene + the world box of what it keeps (what tensorBase.py:636-661 computes): the
        one-step opacity on a lattice of `gridSize` points, dilated by two lattice cells in every direction (the 5^3
        maximum filter, separable: a running maximum along z, then y, then x), thresholded at alphaMask_thres.  The
        volume is stored [z][y][x] as the render kernels index it.  The box is read off the kept INDEX range per axis
        (lattice coordinates are monotonic along an axis), with the lattice's own fp32 coordinate expression."""
        gx, gy, gz = (int(v) for v in gridSize)
        alpha, _ = self.getDenseAlpha((gx, gy, gz))                      # [gx, gy, gz]
        vol = alpha.clamp(0, 1).permute(2, 1, 0).contiguous()            # [gz, gy, gx]
        for axis in range(3):                                            # separable 5-wide maximum, -inf padded
            moved = vol.movedim(axis, -1)
            flat = moved.reshape(1, -1, moved.shape[-1])
            vol = F.max_pool1d(flat, kernel_size=5, stride=1, padding=2).reshape(moved.shape).movedim(-1, axis)
        keep = vol >= self.alphaMask_thres
        self.alphaMask = AlphaGridMask(keep.device, self.aabb, keep.to(torch.float32))
        lo, hi = self.aabb[0].float(), self.aabb[1].float()
        box = torch.empty(2, 3)
        for a, (n, reduce_dims) in enumerate(((gx, (0, 1)), (gy, (0, 2)), (gz, (1, 2)))):
            idx = torch.nonzero(keep.any(dim=reduce_dims[1]).any(dim=reduce_dims[0])).flatten().cpu()
            s = torch.linspace(0, 1, n)[torch.stack([idx.min(), idx.max()])]
            box[:, a] = lo[a] * (1 - s) + hi[a] * s
        return box.to(keep.device)

    @torch.no_grad()
    def shrink(self, new_aabb):
        """Crop the scene to `new_aabb` (tensoRF.py:297-334's result): the texel range [first, last] that covers the box
        on every axis, every factor narrowed to it (channel-last storage kept), and -- when the occupancy volume was
        taken on another lattice than the factor grid -- the box snapped to the texels that were kept."""
        box = new_aabb.detach().to(self.aabb.device, dtype=torch.float32)    # box arithmetic on the host, fp32
        grid = self.gridSize.to(box.device)
        first = torch.round((box[0] - self.aabb[0]) / self.units).long()
        stop = torch.minimum(torch.round((box[1] - self.aabb[0]) / self.units).long() + 1, grid)  # one past the last
        count = stop - first

        def crop(p, dims):  # dims: {tensor dimension: scene axis}
            for d, axis in dims.items():
                p = p.narrow(d, int(first[axis]), int(count[axis]))
            return _channel_last_param(p.contiguous())
        for i in range(3):
            m0, m1 = MAT_MODE[i]
            for lines, planes in ((self.density_line, self.density_plane), (self.app_line, self.app_plane)):
                lines[i] = crop(lines[i].data, {2: VEC_MODE[i]})
                planes[i] = crop(planes[i].data, {2: m1, 3: m0})
        if not torch.equal(self.alphaMask.gridSize.cpu(), self.gridSize.cpu()):
            f0, f1 = first / (grid - 1), (stop - 1) / (grid - 1)
            box = torch.stack([(1 - f0) * self.aabb[0] + f0 * self.aabb[1], (1 - f1) * self.aabb[0] + f1 * self.aabb[1]])
        self.aabb = box
        self.update_stepSize([int(v) for v in count])

    # ---- checkpoint extras (tensorBase.py:508-552) ------------------------------------------------
    def get_reset_kwargs(self):
        return {"aabb": self.aabb, "gridSize": self.gridSize.tolist(), "density_n_comp": self.density_n_comp,
                "appearance_n_comp": self.app_n_comp, "app_dim": self.app_dim, "density_shift": self.density_shift,
                "alphaMask_thres": self.alphaMask_thres, "distance_scale": self.distance_scale,
                "rayMarch_weight_thres": self.rayMarch_weight_thres, "fea2denseAct": self.fea2denseAct,
                "near_far": self.near_far, "step_ratio": self.step_ratio, "shadingMode": self.shadingMode,
                "pos_pe": self.pos_pe, "view_pe": self.view_pe, "fea_pe": self.fea_pe, "featureC": self.featureC,
                "volume_init_scale": self.volume_init_scale, "volume_init_bias": self.volume_init_bias}

    def save_param_state(self):
        """tensorBase.py:531-545: reset kwargs + the alpha mask as packed bits."""
        ckpt = {"tensorf_reset_kwargs": self.get_reset_kwargs()}
        if self.alphaMask is not None:
            vol = self.alphaMask.alpha_volume.bool().cpu().numpy()
            ckpt.update({"alphaMask.shape": vol.shape, "alphaMask.mask": np.packbits(vol.reshape(-1)),
                         "alphaMask.aabb": self.alphaMask.aabb.cpu()})
        return ckpt

    def load_param_state(self, ckpt):
        """tensorBase.py:547-552."""
        self.reset(**ckpt["tensorf_reset_kwargs"])
        if "alphaMask.aabb" in ckpt:
            n = int(np.prod(ckpt["alphaMask.shape"]))
            vol = torch.from_numpy(np.unpackbits(ckpt["alphaMask.mask"])[:n].reshape(ckpt["alphaMask.shape"]))
            dev = self.density_plane[0].device
            self.alphaMask = AlphaGridMask(dev, ckpt["alphaMask.aabb"].to(dev), vol.to(device=dev, dtype=torch.float32))

    # ---- small pure helpers kept for API parity ---------------------------------------------------
    def normalize_coord(self, xyz_sampled):
        return (xyz_sampled - self.aabb[0].to(xyz_sampled.device)) * self.invaabbSize.to(xyz_sampled.device) - 1

    def feature2density(self, density_features):
        if self.fea2denseAct == "softplus":
            return F.softplus(density_features + self.density_shift)
        return F.relu(density_features + self.density_shift)

    def get_kernel(self, opt, c2f_mode, c2f_parameter, c2f_kernel_size=25, device=None):
        """batBase.py:13-25: sigma in voxels = mean(gridSize/aabbSize) * parameter (fp32).  device="cpu": the taps as
        a host tensor (a replayed hipGraph reads them from static device memory, written by jt_poke)."""
        device = self.device if device is None else device
        scale = torch.mean(self.gridSize.to(torch.float32) / (self.aabb[1] - self.aabb[0]))
        sig = (scale * c2f_parameter).to(torch.float32)
        if c2f_mode == "uniform-gaussian":
            return ops.gaussian_taps(sig, c2f_kernel_size, device)
        if c2f_mode == "uniform-average":
            return _average_kernel(float(sig), c2f_kernel_size).to(device)
        raise RuntimeError(f"invalid c2f_mode {c2f_mode}")

    def _check_flags(self, opt):
        arch = getattr(opt, "arch", None)
        if arch is None:
            return
        for k in ("abs_components", "component_wise_feature2density", "plane_feature2density", "convolve_plane_only",
                  "convolve_positive_only", "ignore_negative_split"):
            if bool(getattr(arch, k, False) if not isinstance(arch, dict) else arch.get(k, False)):
                raise NotImplementedError("arch.%s=true is outside the hot path (false in every BAT yaml)" % k)
        # the ray generator always applies the attached NDC centre shift (camera.py:308-314 with the BAT yamls' values)
        get = (lambda k, d: arch.get(k, d)) if isinstance(arch, dict) else (lambda k, d: getattr(arch, k, d))
        if not bool(get("ndc_center_shift", True)):
            raise NotImplementedError("arch.ndc_center_shift=false is not built (true in bat_llff_VM_MLP)")
        if bool(get("detach_ndc_center_shift", False)):
            raise NotImplementedError("arch.detach_ndc_center_shift=true is not built (false in bat_llff_VM_MLP)")

    def _render_cfg(self, S, ndc_ray, white_bg, plane_hw=None, view_pe_progress=1.0, fea_pe_progress=1.0, use_mask=True,
                    near_dev=None):
        """the scene description a render launch takes (ops.RenderCfg mirrors JtScene)"""
        g = self.gridSize.tolist()
        if plane_hw is None:
            plane_hw = [(g[MAT_MODE[i][1]], g[MAT_MODE[i][0]]) for i in range(3)]
        if len(set(self.density_n_comp)) != 1 or len(set(self.app_n_comp)) != 1:
            raise NotImplementedError("per-plane component counts must be equal (they are in both BAT yamls)")
        return ops.RenderCfg(
            aabb=self.aabb.view(-1).tolist(), plane_hw=plane_hw, line_len=[g[VEC_MODE[i]] for i in range(3)],
            n_comp_density=self.density_n_comp[0], n_comp_app=self.app_n_comp[0], step_size=float(self.stepSize),
            near_far=(float(self.near_far[0]), float(self.near_far[1])), distance_scale=self.distance_scale,
            density_shift=self.density_shift,
            density_act=_lib.JT_ACT_SOFTPLUS if self.fea2denseAct == "softplus" else _lib.JT_ACT_RELU,
            weight_thres=self.rayMarch_weight_thres, n_samples=S, ndc=ndc_ray, white_bg=white_bg, app_dim=self.app_dim,
            mlp_kind=self.renderModule.kind, mlp_hidden=self.featureC, view_pe=self.view_pe, fea_pe=self.fea_pe,
            view_pe_progress=view_pe_progress, fea_pe_progress=fea_pe_progress,
            alpha_mask=self.alphaMask.kernel_args() if (self.alphaMask is not None and use_mask) else None,
            near_dev=near_dev)

    def render_pose_fused(self, opt, center, ray_dir, image, ray_idx, rays_per_view, white_bg=True, ndc_ray=False,
                          N_samples=-1, view_pe_progress=1.0, fea_pe_progress=1.0):
        """Test-time pose optimisation's render (scene frozen, no blur, no jitter): loss and its gradient w.r.t. the rays
        from ONE launch (ops.render_pose_fused / csrc/jt_fused.hip).  Same arguments as forward() plus the supervising
        images [views,3,H,W] and the lattice's pixel indices; returns (render loss, rgb [R,3], depth [R], opacity [R])."""
        self._check_flags(opt)
        self.opt = opt
        self.__dict__.setdefault("_reg_cache", {}).clear()
        self.kernel_density = self.kernel_color = None
        self.c2f_mode = None
        S = N_samples if N_samples > 0 else self.nSamples
        near, far = float(self.near_far[0]), float(self.near_far[1])
        zvals = torch.linspace(near, far, S, device=center.device, dtype=torch.float32) if ndc_ray else None
        cfg = self._render_cfg(S, ndc_ray, bool(white_bg), None, view_pe_progress, fea_pe_progress, use_mask=True)
        out = ops.render_pose_fused(cfg, center, ray_dir, zvals, image, ray_idx, rays_per_view, list(self.density_plane),
                                    list(self.density_line), list(self.app_plane), list(self.app_line), self.basis_mat.weight,
                                    self.renderModule.weights())
        self.last_render_cfg = cfg
        return out

    # ---- the renderer (batBase.py:44-165) -----------------------------------------------------------
    def forward(self, opt, center, ray_dir, white_bg=True, is_train=False, ndc_ray=False, N_samples=-1,
                c2f_parameter_density=None, c2f_parameter_color=None, c2f_mode=None, c2f_kernel_size=None,
                is_test_optim=False, view_pe_progress=1.0, fea_pe_progress=1.0):
        self._check_flags(opt)
        self.opt = opt
        # regulariser sums are per forward call (they hang on this call's autograd graph) -- unless the scene is frozen (test-time
        # pose optimisation, model/bat.py:265-292: 400 iterations per view over the same factors): plain values then, kept for as
        # long as the factors' versions stand (_reg_key)
        if self.density_plane[0].requires_grad or self.app_plane[0].requires_grad or self.density_line[0].requires_grad:
            self.__dict__.setdefault("_reg_cache", {}).clear()
        dev = center.device
        S = N_samples if N_samples > 0 else self.nSamples
        near, far = float(self.near_far[0]), float(self.near_far[1])
        R = center.shape[0]
        jitter = zvals = near_dev = None
        if ndc_ray:
            zs = getattr(self, "zvals_static", None)
            if zs is not None and zs[0].numel() == S:
                near_dev = zs[0]  # its first element IS the current near plane: the depth map's "- near" follows it too
                # hipGraph capture / replay (graphed.GraphedTrainStep): the un-jittered row linspace(near, far, S) and the
                # jitter scale (far - near) / S live in static device memory, refreshed in front of every replay -- the
                # near plane follows a schedule (model/tensorf.py:230-232) and must not be baked into the graph
                zvals, zscale = zs[0].view(1, -1), zs[1]
            else:
                zvals, zscale = torch.linspace(near, far, S, device=dev, dtype=torch.float32).unsqueeze(0), (far - near) / S
            if is_train:
                u = self.jitter_override if self.jitter_override is not None else torch.rand_like(zvals)
                zvals = zvals + u.to(dev).view(1, -1) * zscale
        elif is_train:
            # (tests pin the draws with a [>= R, 1] tensor; the first R rows are this batch's)
            jitter = self.jitter_override[:R] if self.jitter_override is not None else torch.rand(R, 1, device=dev)
            jitter = jitter.to(dev)
        # blur kernels (batBase.py:91-101)
        self.c2f_mode = c2f_mode
        if c2f_mode is not None and getattr(self, "taps_static", None) is not None:
            # hipGraph capture / replay (graphed.GraphedTrainStep): the taps live in static device memory that the
            # stepper rewrites in front of every replay
            self.kernel_density, self.kernel_color = self.taps_static
        elif c2f_mode is not None:
            kd_mode = "uniform-gaussian" if is_test_optim else c2f_mode
            self.kernel_density = self.get_kernel(opt, kd_mode, c2f_parameter_density, c2f_kernel_size)
            self.kernel_color = self.get_kernel(opt, c2f_mode, c2f_parameter_color, c2f_kernel_size)
        else:
            self.kernel_density = self.kernel_color = None
        dP, dL, aP, aL = list(self.density_plane), list(self.density_line), list(self.app_plane), list(self.app_line)
        g = self.gridSize.tolist()
        plane_hw = [(g[MAT_MODE[i][1]], g[MAT_MODE[i][0]]) for i in range(3)]
        if c2f_mode is not None:
            # the reference blurs planes through a reshape that exchanges (H, W) in the shape (bateRF.py:29,
            # SURVEY.md App. B-10): the blurred plane i is [1, C, g[m0], g[m1]] and is sampled as such
            plane_hw = [(g[MAT_MODE[i][0]], g[MAT_MODE[i][1]]) for i in range(3)]
            dP, dL, aP, aL = ops.blur_factors(self.kernel_density, self.kernel_color, dP, dL, aP, aL)
        # white background: static flag or the reference's CPU coin (batBase.py:154)
        if white_bg:
            wb = True
        elif is_train:
            coin = self.coin_override if self.coin_override is not None else float(torch.rand((1,)))
            wb = coin < 0.5
        else:
            wb = False
        cfg = self._render_cfg(S, ndc_ray, wb, plane_hw, view_pe_progress, fea_pe_progress,
                               # empty-space samples are dropped only while the blur is off (batBase.py:76-82)
                               use_mask=(c2f_mode is None and c2f_parameter_density is None and c2f_parameter_color is None),
                               near_dev=near_dev)
        # blur off and a backward to come: the regularisers ride on the render node (ops.RenderRays), so that their
        # gradient is added into the render gradient in place; _reg() then finds the values in its cache
        lw = opt.get("loss_weight", None) if isinstance(opt, dict) else getattr(opt, "loss_weight", None)
        fuse_reg = (c2f_mode is None and lw is not None and torch.is_grad_enabled()
                    and all(p.requires_grad for p in dP + dL + aP))
        if fuse_reg:
            cfg.reg_flags = (float(lw.get("TV_density", 0) or 0) != 0.0, float(lw.get("TV_color", 0) or 0) != 0.0)
            # what dL/d(L1, TV_density, TV_color) of THIS call is going to be, if the caller said so (bat_hip.Graph.render_rays
            # from Model.fused_loss_weights): lets the render node write the regularisers' gradient in its forward launch
            cfg.reg_weights = self.__dict__.pop("reg_weights_hint", None)
        rgb, depth, opacity = ops.render_rays(cfg, center, ray_dir, jitter, zvals, dP, dL, aP, aL,
                                              self.basis_mat.weight, self.renderModule.weights())
        self.last_render_cfg = cfg  # the scene description of the last forward (incl. cfg.shade_lists)
        if fuse_reg and cfg.reg3 is not None:
            self.reg_with_tv = cfg.reg_flags
            self._reg_cache.update(key=self._reg_key(), val=cfg.reg3)
        return rgb, depth, opacity


class AlphaGridMask(torch.nn.Module):
    """tensorBase.py:80-98: a 0/1 occupancy volume [1,1,gz,gy,gx] over its own box."""

    def __init__(self, device, aabb, alpha_volume):
        super().__init__()
        self.device = device
        self.aabb = aabb.to(device)
        self.aabbSize = self.aabb[1] - self.aabb[0]
        self.invgridSize = 1.0 / self.aabbSize * 2
        self.alpha_volume = alpha_volume.view(1, 1, *alpha_volume.shape[-3:]).to(device=device, dtype=torch.float32).contiguous()
        self.gridSize = torch.LongTensor([alpha_volume.shape[-1], alpha_volume.shape[-2], alpha_volume.shape[-3]]).to(device)

    def normalize_coord(self, xyz_sampled):
        return (xyz_sampled - self.aabb[0]) * self.invgridSize - 1

    def sample_alpha(self, xyz_sampled):
        g = self.normalize_coord(xyz_sampled)
        return F.grid_sample(self.alpha_volume, g.view(1, -1, 1, 1, 3), align_corners=True).view(-1)

    def kernel_args(self):
        """(volume [z,y,x], lo, inv) as the render kernels take them (JtScene.mask_*, JtFactors.alpha_volume)."""
        return self.alpha_volume[0, 0], self.aabb[0].tolist(), self.invgridSize.tolist()


def _average_kernel(t, kernel_size):
    """kernels.get_average_kernel (kernels.py:25-41)."""
    if kernel_size % 2 == 0:
        kernel_size += 1
    t0 = min(math.floor(t), kernel_size // 2)
    k0 = torch.zeros(kernel_size)
    k0[kernel_size // 2 - t0:kernel_size // 2 + t0 + 1] = 1 / (t0 * 2 + 1)
    t1 = min(math.ceil(t), kernel_size // 2)
    k1 = torch.zeros(kernel_size)
    k1[kernel_size // 2 - t1:kernel_size // 2 + t1 + 1] = 1 / (t1 * 2 + 1)
    return (t % 1.0) * k1 + (1 - t % 1.0) * k0


class TensorVMSplit(BAT_VMSplit):
    """The plain TensoRF scene (tensoRF.py:136-334, `arch.tensorf.model: TensorVMSplit`, known poses): the same
    kernels with the coarse-to-fine blur arguments ignored (TensorBase.forward has none)."""

    def forward(self, opt, center, ray_dir, white_bg=True, is_train=False, ndc_ray=False, N_samples=-1,
                is_test_optim=False, view_pe_progress=1.0, fea_pe_progress=1.0, **_c2f_ignored):
        return super().forward(opt, center, ray_dir, white_bg=white_bg, is_train=is_train, ndc_ray=ndc_ray,
                               N_samples=N_samples, is_test_optim=is_test_optim, view_pe_progress=view_pe_progress,
                               fea_pe_progress=fea_pe_progress)
