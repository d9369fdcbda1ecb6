"""CPU oracle for the TensoRF-VM joint pose + radiance-field render path.

TEST INFRASTRUCTURE ONLY.  This is a torch-fp32 restatement of the reference algorithm
(Nemo1999/Joint-TensoRF) for the hot path of SURVEY.md §8(a).  It is imported only by
tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg -- as the checker or the
reported CPU baseline, never by the product path (joint_tensorf_amd/ fails loudly when its
HIP library is missing; it has no CPU fallback).

Parity is PINNED: every function here is checked against golden vectors captured from the
reference itself (tools/make_golden.py imports /root/reference in the build container; the
fixtures live in tests/golden/, see tests/test_oracle_golden.py).

Each function cites the reference file:line it restates (paths relative to the reference
repo root).  Layout conventions follow the reference: plane i is [1,C,g[m1],g[m0]] with
matMode = [[0,1],[0,2],[1,2]], line i is [1,C,g[v],1] with vecMode = [2,1,0]
(model/tensorf_repr/tensorBase.py:405-406, tensoRF.py:159-169).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

MAT_MODE = ((0, 1), (0, 2), (1, 2))
VEC_MODE = (2, 1, 0)


# ----------------------------------------------------------------------------------------------
# A1  pose: se3 -> SE3, composition                                  camera.py:50-57,81-99,122-145
# ----------------------------------------------------------------------------------------------
def _taylor(theta, kind, nth=8):
    """Taylor series used by the reference instead of closed forms (camera.py:122-145).

    kind 'A': sin(x)/x, 'B': (1-cos x)/x^2, 'C': (x-sin x)/x^3; nth=8 at the call site
    camera.py:91-93 (9 terms each)."""
    ans = torch.zeros_like(theta)
    denom = 1.0
    for i in range(nth + 1):
        if kind == "A":
            if i > 0:
                denom *= (2 * i) * (2 * i + 1)
        elif kind == "B":
            denom *= (2 * i + 1) * (2 * i + 2)
        else:
            denom *= (2 * i + 2) * (2 * i + 3)
        ans = ans + (-1) ** i * (theta ** (2 * i) / denom)
    return ans


def skew(w):
    w0, w1, w2 = w.unbind(-1)
    O = torch.zeros_like(w0)
    return torch.stack([torch.stack([O, -w2, w1], -1),
                        torch.stack([w2, O, -w0], -1),
                        torch.stack([-w1, w0, O], -1)], -2)


def se3_to_SE3(wu):
    """[...,6] (w,u) -> [...,3,4] = [R | V u]   (camera.py:81-99)."""
    w, u = wu[..., :3], wu[..., 3:]
    wx = skew(w)
    theta = w.norm(dim=-1)[..., None, None]
    I = torch.eye(3, dtype=wu.dtype, device=wu.device)
    A = _taylor(theta, "A")
    B = _taylor(theta, "B")
    C = _taylor(theta, "C")
    wx2 = wx @ wx
    R = I + A * wx + B * wx2
    V = I + B * wx + C * wx2
    return torch.cat([R, V @ u[..., None]], -1)


def compose_pair(pose_a, pose_b):
    """pose_new(x) = pose_b o pose_a (x)   (camera.py:50-57)."""
    R_a, t_a = pose_a[..., :3], pose_a[..., 3:]
    R_b, t_b = pose_b[..., :3], pose_b[..., 3:]
    return torch.cat([R_b @ R_a, R_b @ t_a + t_b], -1)


def train_pose(se3_refine, pose_noise, pose_gt):
    """pose = exp(se3_refine) o pose_noise o pose_GT   (model/bat.py:341-353).

    camera.pose.compose([a, b]) applies a first, i.e. compose([noise, gt]) = gt o noise in the
    reference's naming `pose_new(x) = pose_b o pose_a(x)` with (a,b) = (noise, gt).  pose_noise may
    be None (camera.noise false) and pose_gt may be the [3,4] identity `pose_eye` (LLFF)."""
    pose = pose_gt
    if pose_noise is not None:
        pose = compose_pair(pose_noise, pose)
    return compose_pair(se3_to_SE3(se3_refine), pose)


# ----------------------------------------------------------------------------------------------
# A2  rays for the sampled pixels only                                       camera.py:231-261
# ----------------------------------------------------------------------------------------------
def rays_for_pixels(pose, intr_inv, ray_idx, W):
    """centers, dirs [B,r,3] for pixel indices ray_idx (flat y*W+x), same lattice for every view.

    Restates camera.get_center_and_ray (camera.py:231-261) + the `[:,ray_idx]` gather of
    tensorf.Graph.render (model/tensorf.py:159-161) without building all H*W rays:
    grid_3D = [x+.5, y+.5, 1] @ intr_inv^T ; ray = grid_3D @ R ; center = -(t^T R)."""
    ray_idx = ray_idx.to(torch.long)
    x = (ray_idx % W).to(pose.dtype) + 0.5
    y = torch.div(ray_idx, W, rounding_mode="floor").to(pose.dtype) + 0.5
    hom = torch.stack([x, y, torch.ones_like(x)], -1)  # [r,3]
    grid_3D = hom[None] @ intr_inv.transpose(-1, -2)  # [B,r,3]
    R = pose[..., :3]
    t = pose[..., 3:]
    ray = grid_3D @ R
    center = -(t.transpose(-2, -1) @ R).expand(-1, ray.shape[1], -1)
    return center, ray


# ----------------------------------------------------------------------------------------------
# A3  NDC conversion                                                          camera.py:303-340
# ----------------------------------------------------------------------------------------------
def convert_ndc(center, ray, intr, near=1.0, center_shift=True, detach_shift=False):
    if center_shift:
        shift = (near - center[..., 2:]) / ray[..., 2:] * ray
        center = center + (shift.detach() if detach_shift else shift)
    cx, cy, cz = center.unbind(-1)
    rx, ry, rz = ray.unbind(-1)
    sx = (intr[:, 0, 0] / intr[:, 0, 2])[:, None]
    sy = (intr[:, 1, 1] / intr[:, 1, 2])[:, None]
    cxoz, cyoz = cx / cz, cy / cz
    rxoz, ryoz = rx / rz, ry / rz
    c = torch.stack([sx * cxoz, sy * cyoz, 1 - 2 * near / cz], -1)
    r = torch.stack([sx * (rxoz - cxoz), sy * (ryoz - cyoz), 2 * near / cz], -1)
    return c, r


# ----------------------------------------------------------------------------------------------
# A4  ray-index lattice                                                   model/nerf.py:655-673
# ----------------------------------------------------------------------------------------------
def rand_grid_ray_idx(H, W, n_rays, n_views, offset_x, offset_y):
    """all_view_rand_grid lattice (model/nerf.py:659-670); offsets are the two np.random.randint
    draws (x first, then y).  Returns (ray_idx[int64], step, grid_H, grid_W)."""
    rays_per_view = n_rays // n_views
    area_per_ray = H * W // rays_per_view
    step = math.ceil(area_per_ray ** 0.5)
    sx = torch.arange(offset_x, W, step)
    sy = torch.arange(offset_y, H, step)
    gY, gX = torch.meshgrid(sy, sx, indexing="ij")
    return (gX + gY * W).reshape(-1), step, len(sy), len(sx)


# ----------------------------------------------------------------------------------------------
# scene configuration
# ----------------------------------------------------------------------------------------------
class SceneCfg:
    """Static numbers TensorBase keeps (tensorBase.py:430-488)."""

    def __init__(self, aabb, gridSize, near_far, step_ratio=0.5, density_shift=-10.0, distance_scale=25.0,
                 fea2denseAct="softplus", rayMarch_weight_thres=1e-6, shadingMode="MLP_Fea",
                 view_pe=2, fea_pe=2, ndc_near_plane=1.0):
        self.aabb = torch.as_tensor(aabb, dtype=torch.float32).view(2, 3)
        self.gridSize = [int(g) for g in gridSize]
        self.near_far = [float(near_far[0]), float(near_far[1])]
        self.step_ratio = float(step_ratio)
        self.density_shift = float(density_shift)
        self.distance_scale = float(distance_scale)
        self.fea2denseAct = fea2denseAct
        self.rayMarch_weight_thres = float(rayMarch_weight_thres)
        self.shadingMode = shadingMode
        self.view_pe = int(view_pe)
        self.fea_pe = int(fea_pe)
        self.ndc_near_plane = float(ndc_near_plane)
        # update_stepSize (tensorBase.py:477-486), all fp32 like the reference
        g = torch.tensor(self.gridSize, dtype=torch.long)
        self.aabbSize = self.aabb[1] - self.aabb[0]
        self.invaabbSize = 2.0 / self.aabbSize
        self.units = self.aabbSize / (g - 1)
        self.stepSize = torch.mean(self.units) * self.step_ratio

    def to(self, device):
        for k in ("aabb", "aabbSize", "invaabbSize", "units", "stepSize"):
            setattr(self, k, getattr(self, k).to(device))
        return self


def find_n_samples(resolution, step_ratio, sample_intvs):
    """tensorf.NeRF._find_n_samples (model/tensorf.py:458-461)."""
    return min(int(sample_intvs), int(np.linalg.norm(resolution) / step_ratio))


def find_resolution(bbox, n_voxels, scale=(1.0, 1.0, 1.0)):
    """tensorf.NeRF._find_resolution (model/tensorf.py:449-456), fp32 like the reference."""
    bbox = torch.as_tensor(bbox, dtype=torch.float32).view(2, 3)
    size = bbox[1] - bbox[0]
    voxel = (size.prod() / n_voxels).pow(1 / 3)
    return (size / voxel * torch.tensor(scale)).long().tolist()


# ----------------------------------------------------------------------------------------------
# A5 / A5'  point sampling                                   tensorBase.py:554-571, 572-612
# ----------------------------------------------------------------------------------------------
def sample_ray(cfg, rays_o, rays_d, N_samples, jitter=None):
    """Uniform-step sampling from the AABB entry point (tensorBase.py:572-612).

    jitter: [R,1] uniform draws (one per ray) when training, None otherwise."""
    near, far = cfg.near_far
    od, dd = rays_o.detach(), rays_d.detach()
    vec = torch.where(dd == 0, torch.full_like(dd, 1e-6), dd)
    rate_a = (cfg.aabb[1] - od) / vec
    rate_b = (cfg.aabb[0] - od) / vec
    t_min = torch.minimum(rate_a, rate_b).amax(-1).clamp(min=near, max=far)
    rng = torch.arange(N_samples, dtype=rays_o.dtype, device=rays_o.device)[None]
    if jitter is not None:
        rng = rng.repeat(rays_d.shape[-2], 1) + jitter.view(-1, 1)
    step = cfg.stepSize * rng
    z = t_min[..., None] + step
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., None]
    outside = ((cfg.aabb[0] > pts) | (pts > cfg.aabb[1])).any(dim=-1)
    return pts, z, ~outside


def sample_ray_ndc(cfg, rays_o, rays_d, N_samples, jitter=None):
    """linspace(near,far,S) sampling shared by all rays (tensorBase.py:554-571).

    jitter: [1,S] uniform draws (one row shared by all rays) when training."""
    near, far = cfg.near_far
    z = torch.linspace(near, far, N_samples, dtype=rays_o.dtype, device=rays_o.device)[None]
    if jitter is not None:
        z = z + jitter.view(1, -1) * ((far - near) / N_samples)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., None]
    outside = ((cfg.aabb[0] > pts) | (pts > cfg.aabb[1])).any(dim=-1)
    return pts, z, ~outside


# ----------------------------------------------------------------------------------------------
# A6  1-D Gaussian taps                              kernels.py:16-22, batBase.py:13-25
# ----------------------------------------------------------------------------------------------
def gaussian_kernel(sigma, kernel_size):
    """Un-normalised, clamped-at-1 Gaussian taps n in [-K//2, K//2] (K=64 -> 65 taps)."""
    sigma = float(sigma)
    ns = torch.arange(-(kernel_size // 2), kernel_size // 2 + 1, dtype=torch.float32)
    s = max(sigma, 0.0001)
    k = 1 / (s * math.sqrt(2 * math.pi)) * torch.exp(-0.5 * (ns / s) * (ns / s))
    return torch.clamp(k, max=1.0)


def kernel_sigma_vox(cfg, c2f_parameter):
    """sigma in voxels = mean_xyz(gridSize / aabbSize) * parameter (batBase.py:14-18); fp32."""
    g = torch.tensor(cfg.gridSize, dtype=torch.float32, device=cfg.aabb.device)
    scale = torch.mean(g / (cfg.aabb[1] - cfg.aabb[0]))
    return (scale * c2f_parameter).to(torch.float32)


def get_kernel(cfg, c2f_parameter, kernel_size):
    return gaussian_kernel(kernel_sigma_vox(cfg, c2f_parameter), kernel_size)


def interp_schedule(x, schedule, left=0.0, right=1.0):
    """util.interp_schedule (util.py:217-225): piecewise-linear over equally spaced knots."""
    xs = np.linspace(left, right, len(schedule))
    return np.interp(float(x), xs, schedule)


# ----------------------------------------------------------------------------------------------
# A7  separable replicate-padded blur                                      bateRF.py:8-39
# ----------------------------------------------------------------------------------------------
def _corr_last(x, kernel):
    """Replicate-pad by K//2 and cross-correlate along the last axis; x: [N, L]."""
    K = kernel.numel()
    xp = F.pad(x[:, None, :], (K // 2, K // 2), mode="replicate")
    return F.conv1d(xp, kernel.view(1, 1, -1))[:, 0, :]


def blur_line(kernel, line):
    """line [1,C,L,1] -> same (bateRF.py:8-19)."""
    C, L = line.shape[1], line.shape[2]
    return _corr_last(line.reshape(C, L), kernel).reshape(1, C, L, 1)


def blur_plane(kernel, plane, gm0, gm1):
    """plane [1,C,g[m1],g[m0]] -> [1,C,gm0,gm1] (bateRF.py:21-39, called with (H,W)=(g[m0],g[m1])
    at bateRF.py:68,76,110,117).

    The reference *reshapes* (does not transpose) the [C,g[m1],g[m0]] memory to [C,H=g[m0],W=g[m1]],
    blurs along W then along H, and returns [1,C,H,W].  For g[m0]==g[m1] this is the plain separable
    blur; for non-cubic grids the memory is reinterpreted and the result has H/W swapped relative
    to the input.  Restated literally so that both cases match."""
    C = plane.shape[1]
    H, W = int(gm0), int(gm1)
    x = plane.reshape(C * H, W)
    x = _corr_last(x, kernel).reshape(C, H, W)  # along W
    x = x.permute(0, 2, 1).reshape(C * W, H)
    x = _corr_last(x, kernel).reshape(C, W, H)  # along H
    return x.permute(0, 2, 1).reshape(1, C, H, W).contiguous()


# ----------------------------------------------------------------------------------------------
# A8 / A11  VM-factor interpolation                          bateRF.py:41-94, 97-130
# ----------------------------------------------------------------------------------------------
def normalize_coord(cfg, xyz):
    """tensorBase.py:502-503."""
    return (xyz - cfg.aabb[0]) * cfg.invaabbSize - 1


def bilinear_taps(plane, gx, gy):
    """Manual `grid_sample(bilinear, align_corners=True, padding_mode='zeros')` of plane
    [1,C,H,W] at normalised (gx,gy) [P] -> [C,P].  This is the tap arithmetic the HIP kernels
    implement (ix = (g+1)/2*(size-1), floor cell, out-of-range taps contribute zero)."""
    _, C, H, W = plane.shape
    ix = (gx + 1) / 2 * (W - 1)
    iy = (gy + 1) / 2 * (H - 1)
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    fx, fy = ix - x0, iy - y0
    x0, y0 = x0.long(), y0.long()
    out = 0
    p = plane[0]
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xx, yy = x0 + dx, y0 + dy
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            v = p[:, yy.clamp(0, H - 1), xx.clamp(0, W - 1)]  # [C,P]
            out = out + v * (wx * wy * ok)[None]
    return out


def _sample_plane(plane, gx, gy, use_taps):
    if use_taps:
        return bilinear_taps(plane, gx, gy)
    grid = torch.stack([gx, gy], -1).view(1, -1, 1, 2)
    return F.grid_sample(plane, grid, mode="bilinear", align_corners=True).view(plane.shape[1], -1)


def _sample_line(line, g, use_taps):
    if use_taps:
        return bilinear_taps(line, torch.zeros_like(g), g)
    grid = torch.stack([torch.zeros_like(g), g], -1).view(1, -1, 1, 2)
    return F.grid_sample(line, grid, mode="bilinear", align_corners=True).view(line.shape[1], -1)


def _maybe_blur(cfg, planes, lines, kernel):
    if kernel is None:
        return planes, lines
    g = cfg.gridSize
    bp = [blur_plane(kernel, planes[i], g[MAT_MODE[i][0]], g[MAT_MODE[i][1]]) for i in range(3)]
    bl = [blur_line(kernel, lines[i]) for i in range(3)]
    return bp, bl


def density_feature(cfg, params, xyz_n, kernel=None, use_taps=False):
    """sigma_feat(p) = sum_i sum_c plane_i^c(p[m0],p[m1]) * line_i^c(p[v])   (bateRF.py:41-94)."""
    planes, lines = _maybe_blur(cfg, params["density_plane"], params["density_line"], kernel)
    feat = torch.zeros(xyz_n.shape[0], dtype=xyz_n.dtype, device=xyz_n.device)
    for i in range(3):
        m0, m1 = MAT_MODE[i]
        pc = _sample_plane(planes[i], xyz_n[:, m0], xyz_n[:, m1], use_taps)
        lc = _sample_line(lines[i], xyz_n[:, VEC_MODE[i]], use_taps)
        feat = feat + torch.sum(pc * lc, dim=0)
    return feat


def app_feature(cfg, params, xyz_n, kernel=None, use_taps=False):
    """[Ps, app_dim] = basis_mat(cat_i plane_i * line_i)   (bateRF.py:97-130, tensoRF.py:156)."""
    planes, lines = _maybe_blur(cfg, params["app_plane"], params["app_line"], kernel)
    pcs, lcs = [], []
    for i in range(3):
        m0, m1 = MAT_MODE[i]
        pcs.append(_sample_plane(planes[i], xyz_n[:, m0], xyz_n[:, m1], use_taps))
        lcs.append(_sample_line(lines[i], xyz_n[:, VEC_MODE[i]], use_taps))
    prod = (torch.cat(pcs) * torch.cat(lcs)).T
    return prod @ params["basis"].T


# ----------------------------------------------------------------------------------------------
# A9 / A10  density activation, alpha compositing weights        tensorBase.py:57-65, 696-700
# ----------------------------------------------------------------------------------------------
def feature2density(cfg, feat):
    if cfg.fea2denseAct == "softplus":
        return F.softplus(feat + cfg.density_shift)
    if cfg.fea2denseAct == "relu":
        return F.relu(feat + cfg.density_shift)
    raise ValueError(cfg.fea2denseAct)


def raw2alpha(sigma, dist):
    alpha = 1.0 - torch.exp(-sigma * dist)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)
    return alpha, alpha * T[:, :-1], T[:, -1:]


# ----------------------------------------------------------------------------------------------
# A12 / A12'  appearance MLPs                         tensorBase.py:43-55, 101-126, 180-214
# ----------------------------------------------------------------------------------------------
def positional_encoding(x, freqs, progress=1.0):
    """Per input channel: [sin(2^0 x), sin(2^1 x), ..., cos(2^0 x), cos(2^1 x), ...] * mask."""
    levels = torch.arange(freqs, device=x.device)
    bands = 2 ** levels
    mask = (progress * freqs - levels).clamp(min=0.0, max=1.0)
    pts = x[..., None] * bands
    pts = torch.cat([torch.sin(pts) * mask, torch.cos(pts) * mask], dim=-1)
    return pts.reshape(x.shape[:-1] + (freqs * 2 * x.shape[-1],))


# Test knob: evaluate the Linear layers in this dtype and round the result back (None = plain F.linear).  A ReLU whose
# pre-activation is within rounding of zero is a discrete decision that depends on the GEMM's summation order; two fp32
# evaluations of the same reference algorithm that differ only in that order disagree on a handful of units out of
# 10^8, and each disagreement moves the gradient of the few texels that sample touches.  tests/test_gpu_fullsize.py
# measures that sensitivity (fp32 GEMMs vs fp64-accumulated GEMMs) next to the error of the implementation under test.
LINEAR_DTYPE = None


def _linear(x, w, b):
    if LINEAR_DTYPE is None or LINEAR_DTYPE == x.dtype:
        return F.linear(x, w, b)
    return F.linear(x.to(LINEAR_DTYPE), w.to(LINEAR_DTYPE), b.to(LINEAR_DTYPE)).to(x.dtype)


def _relu(x, mask, report):
    """ReLU, or -- when a test pins the discrete decision -- multiplication by the 0/1 `mask` the implementation under
    test used; `report` collects how many units the oracle itself would have decided differently and how far from zero
    the farthest of those pre-activations is (they must all be within rounding of zero)."""
    if mask is None:
        return F.relu(x)
    own = x.detach() > 0
    diff = own != mask
    if report is not None:
        report["units"] = report.get("units", 0) + x.numel()
        report["flips"] = report.get("flips", 0) + int(diff.sum())
        # direction of the disagreements (a systematic bias of the implementation under test would show as one-sided)
        report["flips_oracle_on"] = report.get("flips_oracle_on", 0) + int((diff & own).sum())
        if diff.any():
            report["max_abs"] = max(report.get("max_abs", 0.0), float(x.detach()[diff].abs().max()))
    return x * mask.to(x.dtype)


def mlp_fea(cfg, mlp, feat, viewdirs, view_pe_progress=1.0, fea_pe_progress=1.0, relu_masks=None, relu_report=None):
    """MLPRender_Fea (tensorBase.py:116-126): [f, d, PE(f), PE(d)] -> 3x Linear -> sigmoid."""
    x = [feat, viewdirs]
    if cfg.fea_pe > 0:
        x.append(positional_encoding(feat, cfg.fea_pe, fea_pe_progress))
    if cfg.view_pe > 0:
        x.append(positional_encoding(viewdirs, cfg.view_pe, view_pe_progress))
    x = torch.cat(x, -1)
    m1, m2 = relu_masks if relu_masks is not None else (None, None)
    h = _relu(_linear(x, mlp["w1"], mlp["b1"]), m1, relu_report)
    h = _relu(_linear(h, mlp["w2"], mlp["b2"]), m2, relu_report)
    return torch.sigmoid(_linear(h, mlp["w3"], mlp["b3"]))


def mlp_weakview(cfg, mlp, feat, viewdirs, view_pe_progress=1.0, fea_pe_progress=1.0, relu_masks=None,
                 relu_report=None):
    """MLPRender_Fea_WeakView (tensorBase.py:198-214): view PE joins only at the last layer."""
    x = [feat]
    if cfg.fea_pe > 0:
        x.append(positional_encoding(feat, cfg.fea_pe, fea_pe_progress))
    x = torch.cat(x, -1)
    m1, m2 = relu_masks if relu_masks is not None else (None, None)
    h = _relu(_linear(x, mlp["w1"], mlp["b1"]), m1, relu_report)
    h = _relu(_linear(h, mlp["w2"], mlp["b2"]), m2, relu_report)
    mid = []
    if cfg.view_pe > 0:
        mid.append(positional_encoding(viewdirs, cfg.view_pe, view_pe_progress))
    mid.append(h)
    return torch.sigmoid(_linear(torch.cat(mid, -1), mlp["w3"], mlp["b3"]))


# ----------------------------------------------------------------------------------------------
# BatBase.forward                                                       batBase.py:44-165
# ----------------------------------------------------------------------------------------------
def render(cfg, params, center, ray_dir, N_samples, white_bg=True, jitter=None, ndc_ray=False,
           kernel_density=None, kernel_color=None, view_pe_progress=1.0, fea_pe_progress=1.0,
           use_taps=False, return_aux=False, alpha_mask=None, app_mask_override=None, relu_masks_override=None,
           relu_report=None):
    """rgb [R,3], depth [R], opacity [R] for rays (center, ray_dir) [R,3].

    `white_bg` is the already-resolved flag `white_bg or (is_train and coin<0.5)` (batBase.py:154);
    `jitter` is None when not training; kernels are None when blur is inactive (the caller applies
    the random-scale / cut-off logic of model/tensorf.py:175-220)."""
    viewdirs = ray_dir
    if ndc_ray:
        xyz, z, valid = sample_ray_ndc(cfg, center, viewdirs, N_samples, jitter)
        dists = torch.cat([z[:, 1:] - z[:, :-1], torch.zeros_like(z[:, :1])], -1)
        norm = torch.norm(viewdirs, dim=-1, keepdim=True)
        dists = dists * norm
        viewdirs = viewdirs / norm
    else:
        xyz, z, valid = sample_ray(cfg, center, viewdirs, N_samples, jitter)
        dists = torch.cat([z[:, 1:] - z[:, :-1], torch.zeros_like(z[:, :1])], -1)
    R, S = valid.shape
    viewdirs = viewdirs.view(-1, 1, 3).expand(R, S, 3).detach()  # arch.shading.detach_viewdirs
    if alpha_mask is not None and kernel_density is None and kernel_color is None:
        # samples in empty space are dropped -- only while the blur is off (batBase.py:76-82)
        keep = alpha_mask_sample(alpha_mask, xyz[valid].detach()) > 0
        valid = valid.clone()
        valid[valid.clone()] = keep

    sigma = torch.zeros(R, S, dtype=center.dtype, device=center.device)
    rgb = torch.zeros(R, S, 3, dtype=center.dtype, device=center.device)
    xyz_n = normalize_coord(cfg, xyz)
    sigma_feat = None
    if valid.any():
        sigma_feat = density_feature(cfg, params, xyz_n[valid], kernel_density, use_taps)
        sigma = torch.zeros_like(sigma).index_put((valid,), feature2density(cfg, sigma_feat))
    alpha, weight, bg_weight = raw2alpha(sigma, dists * cfg.distance_scale)
    app_mask = weight > cfg.rayMarch_weight_thres
    own_app_mask = app_mask
    if app_mask_override is not None:
        # tests pin this discrete decision (a weight within rounding of the threshold) to the one the implementation
        # under test took, and check separately that every sample decided differently is such a near-tie
        app_mask = app_mask_override
    if app_mask.any():
        feat = app_feature(cfg, params, xyz_n[app_mask], kernel_color, use_taps)
        mlp = mlp_weakview if cfg.shadingMode == "MLP_Fea_WeakView" else mlp_fea
        # relu_masks_override: ([n, hidden] 0/1, [n, hidden] 0/1) for the shaded samples in row-major (ray, sample) order
        rgbs = mlp(cfg, params["mlp"], feat, viewdirs[app_mask], view_pe_progress, fea_pe_progress,
                   relu_masks=relu_masks_override, relu_report=relu_report)
        rgb = rgb.index_put((app_mask,), rgbs)
    acc = torch.sum(weight, -1)
    rgb_map = torch.sum(weight[..., None] * rgb, -2)
    with torch.no_grad():
        depth = torch.sum(weight * z, -1) + (1.0 - acc) * ray_dir[..., -1]
        depth = depth - cfg.near_far[0] + 0.05
    if white_bg:
        rgb_map = rgb_map + (1.0 - acc[..., None])
    rgb_map = rgb_map.clamp(0, 1)
    if return_aux:
        aux = dict(xyz=xyz, z=z, valid=valid, sigma=sigma, alpha=alpha, weight=weight, app_mask=app_mask,
                   own_app_mask=own_app_mask,
                   rgb_samples=rgb, dists=dists, sigma_feat=sigma_feat)
        return rgb_map, depth, acc, aux
    return rgb_map, depth, acc


# ----------------------------------------------------------------------------------------------
# A16  schedule glue of Graph.render_rays                             model/tensorf.py:169-261
# ----------------------------------------------------------------------------------------------
def resolve_blur(progress, schedule_density, schedule_color, mode, random_scale=None, eps=1e-3):
    """Returns (c2f_parameter_density, c2f_parameter_color) or (None, None) when blur is dropped.

    Order matters: random density scale first (not in "vis"), then the cut-off
    `max(param_d, param_c) < eps` (model/tensorf.py:193-220)."""
    pd = interp_schedule(progress, schedule_density)
    pc = interp_schedule(progress, schedule_color)
    if mode != "vis" and random_scale is not None:
        pd = pd * random_scale
    if max(pd, pc) < eps:
        return None, None
    return pd, pc


# ----------------------------------------------------------------------------------------------
# N1  evaluation: camera pre-alignment, test-time pose optimisation, sliced eval render
#                                     camera.py:342-366, model/bat.py:212-292,354-367, model/nerf.py:525-548,728-740
# ----------------------------------------------------------------------------------------------
def camera_centers(pose):
    """world position of the camera centre of world->camera poses [...,3,4]  (camera.cam2world of the origin)."""
    R, t = pose[..., :3], pose[..., 3:]
    return (-R.transpose(-1, -2) @ t)[..., 0]


def rotation_distance(R1, R2, eps=1e-7):
    """camera.py:342-347."""
    Rd = R1 @ R2.transpose(-2, -1)
    trace = Rd[..., 0, 0] + Rd[..., 1, 1] + Rd[..., 2, 2]
    return ((trace - 1) / 2).clamp(-1 + eps, 1 - eps).acos()


def procrustes_analysis(X0, X1):
    """camera.py:349-366: similarity (t0, t1, s0, s1, R) with X1 -> X0: (X1 - t1) / s1 @ R.T * s0 + t0."""
    t0 = X0.mean(dim=0, keepdim=True)
    t1 = X1.mean(dim=0, keepdim=True)
    X0c, X1c = X0 - t0, X1 - t1
    s0 = (X0c ** 2).sum(dim=-1).mean().sqrt()
    s1 = (X1c ** 2).sum(dim=-1).mean().sqrt()
    # same LAPACK driver as the reference's Tensor.svd: with few (coplanar) cameras the smallest singular
    # vector's sign is the routine's choice
    U, S, V = torch.svd(((X0c / s0).t() @ (X1c / s1)).double(), some=True)
    R = (U @ V.t()).float()
    if R.det() < 0:
        R[2] *= -1
    return dict(t0=t0[0], t1=t1[0], s0=s0, s1=s1, R=R)


def prealign_cameras(pose, pose_GT):
    """model/bat.py:212-228: align the optimised training cameras with the ground truth up to a similarity."""
    center_pred, center_GT = camera_centers(pose), camera_centers(pose_GT)
    sim3 = procrustes_analysis(center_GT, center_pred)
    center_aligned = (center_pred - sim3["t1"]) / sim3["s1"] @ sim3["R"].t() * sim3["s0"] + sim3["t0"]
    R_aligned = pose[..., :3] @ sim3["R"].t()
    t_aligned = (-R_aligned @ center_aligned[..., None])[..., 0]
    return torch.cat([R_aligned, t_aligned[..., None]], -1), sim3


def camera_alignment_error(pose_aligned, pose_GT):
    """model/bat.py:230-238: per-view rotation angle [rad] and translation distance."""
    return (rotation_distance(pose_aligned[..., :3], pose_GT[..., :3]),
            (pose_aligned[..., 3] - pose_GT[..., 3]).norm(dim=-1))


def eval_pose(sim3, pose, pose_refine_test=None):
    """model/bat.py:354-367: a held-out GT pose moved into the optimised coordinate system (inverse of the
    similarity above), optionally followed by the test-time refinement."""
    center = camera_centers(pose)
    center_aligned = (center - sim3["t0"]) / sim3["s0"] @ sim3["R"] * sim3["s1"] + sim3["t1"]
    R_aligned = pose[..., :3] @ sim3["R"]
    t_aligned = (-R_aligned @ center_aligned[..., None])[..., 0]
    out = torch.cat([R_aligned, t_aligned[..., None]], -1)
    if pose_refine_test is not None:
        out = compose_pair(pose_refine_test, out)
    return out


def test_time_optim(cfg, params, sim3, pose, image, intr_inv, H, W, n_rays, N_samples, offsets, lr, lr_test,
                    lr_test_end, iters, white_bg=True, view_pe_progress=1.0, fea_pe_progress=1.0):
    """model/bat.py:265-292 for one held-out view of the Blender configuration (no blur at test time):
    Adam on a fresh se3 [1,6] against the photometric loss of a ray lattice (offsets = the np.random.randint
    draws, two per iteration).  Returns (se3, pose_refine_test, trace of se3 before each step, trace of
    loss.render).  Reproduced quirk: the reference refreshes `var.pose_refine_test` at the TOP of every iteration
    only (bat.py:284), so the eval render that follows (nerf.py:539) sees the refinement from before the last
    Adam step; `pose_refine_test` is that stale transform."""
    se3 = torch.zeros(1, 6, requires_grad=True)
    opt = torch.optim.Adam([dict(params=[se3], lr=lr)])
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=(lr_test_end / lr_test) ** (1.0 / iters))
    img = image.view(1, 3, H * W).permute(0, 2, 1)
    tr_se3, tr_loss = [], []
    pose_refine_test = None
    for it in range(iters):
        opt.zero_grad()
        pose_refine_test = se3_to_SE3(se3)
        p = eval_pose(sim3, pose, pose_refine_test)
        ray_idx, _, _, _ = rand_grid_ray_idx(H, W, n_rays, 1, offsets[2 * it], offsets[2 * it + 1])
        center, ray = rays_for_pixels(p, intr_inv, ray_idx, W)
        rgb, _, _ = render(cfg, params, center.reshape(-1, 3), ray.reshape(-1, 3), N_samples, white_bg=white_bg,
                           view_pe_progress=view_pe_progress, fea_pe_progress=fea_pe_progress)
        loss = render_loss(rgb.view(1, -1, 3), img[:, ray_idx])
        tr_se3.append(se3.detach().clone())
        tr_loss.append(float(loss.detach()))
        loss.backward()
        opt.step()
        sched.step()
    return se3.detach(), pose_refine_test.detach(), torch.stack(tr_se3), tr_loss


def render_by_slices(cfg, params, pose, intr_inv, H, W, n_rays, N_samples, **kw):
    """model/nerf.py:728-740: full image in slices of n_rays pixels."""
    outs = [[], [], []]
    with torch.no_grad():
        for c in range(0, H * W, n_rays):
            ray_idx = torch.arange(c, min(c + n_rays, H * W))
            center, ray = rays_for_pixels(pose, intr_inv, ray_idx, W)
            r = render(cfg, params, center.reshape(-1, 3), ray.reshape(-1, 3), N_samples, **kw)
            for o, v in zip(outs, r):
                o.append(v.view(pose.shape[0], len(ray_idx), -1))
    return [torch.cat(o, 1) for o in outs]


# ----------------------------------------------------------------------------------------------
# N3  2-D blur cache of the supervising images + Sobel edge masks           model/nerf.py:57-149
# ----------------------------------------------------------------------------------------------
def process_gt_images(images, progress, schedule, scales, kernel_size, mode="uniform-gaussian"):
    """{scale: images} (model/nerf.py:57-113): separable replicate-padded correlation along W then H with the
    taps of gaussian_kernel(width) where width = interp(progress) * scale * (W + H) / 2; widths below 0.01
    leave the images untouched."""
    assert mode == "uniform-gaussian"
    n, _, H, W = images.shape
    out = {}
    for sc in scales:
        width = torch.tensor(interp_schedule(float(progress), schedule)) * sc * (W + H) / 2
        if width < 0.01:
            out[sc] = images
            continue
        k = gaussian_kernel(width, kernel_size).float().view(1, 1, -1)
        pad = (kernel_size // 2, kernel_size // 2)
        x = images.reshape(n * 3, H, W)
        x = F.conv1d(F.pad(x, pad, mode="replicate"), k.expand(H, 1, -1), groups=H)
        x = x.permute(0, 2, 1)
        x = F.conv1d(F.pad(x, pad, mode="replicate"), k.expand(W, 1, -1), groups=W)
        out[sc] = x.permute(0, 2, 1).reshape(n, 3, H, W).contiguous()
    return out


def edge_masks(blurred, thresh=1.25, soft=False):
    """{scale: mask [n, H*W]} (model/nerf.py:115-149): Sobel magnitude of the channel-summed image, hard mask =
    magnitude > thresh * its per-image mean (uint8), soft mask = magnitude / its per-image max."""
    Kx = torch.tensor([[1., 0., -1.], [2., 0., -2.], [1., 0., -1.]])[None, None].expand(1, 3, -1, -1)
    Ky = torch.tensor([[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]])[None, None].expand(1, 3, -1, -1)
    out = {}
    for sc, img in blurred.items():
        n = img.shape[0]
        x = F.pad(img, (1, 1, 1, 1), mode="replicate")
        GG = torch.sqrt(F.conv2d(x, Kx) ** 2 + F.conv2d(x, Ky) ** 2).view(n, -1)
        if soft:
            out[sc] = GG / GG.max(dim=1, keepdim=True)[0]
        else:
            out[sc] = (GG > GG.mean(dim=1, keepdim=True) * thresh).to(torch.uint8)
    return out


# ----------------------------------------------------------------------------------------------
# N4  alpha-mask volume, AABB shrink        tensorBase.py:80-98,618-661,703-726, batBase.py:27-41, tensoRF.py:297-334
# ----------------------------------------------------------------------------------------------
def alpha_mask_sample(alpha_mask, xyz):
    """AlphaGridMask.sample_alpha (tensorBase.py:92-99): trilinear sample of the volume [D,H,W] = grid[::-1]
    over its own box; alpha_mask = (volume, aabb [2,3])."""
    vol, aabb = alpha_mask
    g = (xyz - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1
    return F.grid_sample(vol[None, None], g.view(1, -1, 1, 1, 3), align_corners=True).view(-1)


def compute_alpha(cfg, params, xyz, length, alpha_mask=None, kernel=None):
    """BatBase.compute_alpha (batBase.py:27-41): opacity of a step of `length` at arbitrary points."""
    keep = torch.ones(xyz.shape[0], dtype=torch.bool) if alpha_mask is None else alpha_mask_sample(alpha_mask, xyz) > 0
    sigma = torch.zeros(xyz.shape[0])
    if keep.any():
        feat = density_feature(cfg, params, normalize_coord(cfg, xyz[keep]), kernel, False)
        sigma[keep] = feature2density(cfg, feat)
    return 1 - torch.exp(-sigma * length)


def dense_alpha(cfg, params, grid, alpha_mask=None, kernel=None):
    """TensorBase.getDenseAlpha (tensorBase.py:618-634): alpha [g0,g1,g2] at the lattice points of the box."""
    with torch.no_grad():
        s = torch.stack(torch.meshgrid(torch.linspace(0, 1, grid[0]), torch.linspace(0, 1, grid[1]),
                                       torch.linspace(0, 1, grid[2]), indexing="ij"), -1)
        xyz = cfg.aabb[0] * (1 - s) + cfg.aabb[1] * s
        alpha = compute_alpha(cfg, params, xyz.view(-1, 3), cfg.stepSize, alpha_mask, kernel).view(*grid)
    return alpha, xyz


def update_alpha_mask(cfg, params, grid, thres, alpha_mask=None, kernel=None):
    """TensorBase.updateAlphaMask (tensorBase.py:636-661): ((volume [g2,g1,g0], aabb), new_aabb)."""
    alpha, xyz = dense_alpha(cfg, params, grid, alpha_mask, kernel)
    xyz = xyz.transpose(0, 2).contiguous()
    alpha = alpha.clamp(0, 1).transpose(0, 2).contiguous()[None, None]
    alpha = F.max_pool3d(alpha, kernel_size=5, padding=2, stride=1).view(grid[2], grid[1], grid[0])
    alpha = (alpha >= thres).float()
    valid = xyz[alpha > 0.5]
    return (alpha, cfg.aabb.clone()), torch.stack((valid.amin(0), valid.amax(0)))


def shrink(cfg, params, new_aabb, mask_grid):
    """TensorVMSplit.shrink (tensoRF.py:297-334): crop the factors to the texels covering new_aabb; returns the
    new (cfg, params).  mask_grid: grid of the alpha volume the box came from (the box is snapped to the factor
    texels when the two grids differ)."""
    g = torch.tensor(cfg.gridSize)
    t_l, b_r = (new_aabb[0] - cfg.aabb[0]) / cfg.units, (new_aabb[1] - cfg.aabb[0]) / cfg.units
    t_l, b_r = torch.round(torch.round(t_l)).long(), torch.round(b_r).long() + 1
    b_r = torch.stack([b_r, g]).amin(0)
    out = {k: v for k, v in params.items()}
    for grp in ("density", "app"):
        planes, lines = [], []
        for i in range(3):
            v = VEC_MODE[i]
            m0, m1 = MAT_MODE[i]
            lines.append(params[grp + "_line"][i].detach()[..., t_l[v]:b_r[v], :].clone())
            planes.append(params[grp + "_plane"][i].detach()[..., t_l[m1]:b_r[m1], t_l[m0]:b_r[m0]].clone())
        out[grp + "_plane"], out[grp + "_line"] = planes, lines
    if list(mask_grid) != cfg.gridSize:
        tl_r, br_r = t_l / (g - 1), (b_r - 1) / (g - 1)
        new_aabb = torch.stack([(1 - tl_r) * cfg.aabb[0] + tl_r * cfg.aabb[1], (1 - br_r) * cfg.aabb[0] + br_r * cfg.aabb[1]])
    new_cfg = SceneCfg(new_aabb.reshape(-1).tolist(), (b_r - t_l).tolist(), cfg.near_far, step_ratio=cfg.step_ratio,
                       density_shift=cfg.density_shift, distance_scale=cfg.distance_scale, fea2denseAct=cfg.fea2denseAct,
                       rayMarch_weight_thres=cfg.rayMarch_weight_thres, shadingMode=cfg.shadingMode, view_pe=cfg.view_pe,
                       fea_pe=cfg.fea_pe, ndc_near_plane=cfg.ndc_near_plane)
    new_cfg.aabb = new_aabb.float()  # keep the exact fp32 box (tolist() round-trips through double exactly)
    return new_cfg, out


# ----------------------------------------------------------------------------------------------
# A14  losses                        model/tensorf.py:96-142, base.py:259-261, tensoRF.py:212-228
# ----------------------------------------------------------------------------------------------
def mse_nanmean(pred, label):
    return ((pred.contiguous() - label) ** 2).nanmean()


def render_loss(rgb, image_at_rays, edge_mask=None, edge_factor=1.5, non_edge_factor=0.5):
    """rgb,image [B,r,3]; edge_mask [B,r] u8 or None -> scalar (model/tensorf.py:112-124)."""
    if edge_mask is None:
        return mse_nanmean(rgb, image_at_rays)
    m = edge_mask[..., None].expand(-1, -1, 3)
    edge = mse_nanmean(rgb * m, image_at_rays * m)
    non_edge = mse_nanmean(rgb * (1 - m), image_at_rays * (1 - m))
    return edge_factor * edge + non_edge_factor * non_edge


def density_L1(params):
    total = 0
    for i in range(3):
        total = total + torch.mean(torch.abs(params["density_plane"][i])) + torch.mean(torch.abs(params["density_line"][i]))
    return total


def tv_loss(x):
    """TVLoss.forward with weight 1 (tensorBase.py:16-41)."""
    b, c, h, w = x.shape
    count_h = c * (h - 1) * w
    count_w = c * h * (w - 1)
    total = 0
    if count_h > 0:
        total = total + torch.pow(x[:, :, 1:, :] - x[:, :, :h - 1, :], 2).sum() / count_h
    if count_w > 0:
        total = total + torch.pow(x[:, :, :, 1:] - x[:, :, :, :w - 1], 2).sum() / count_w
    return 2 * total / b


def tv_planes(planes):
    """TV_loss_density / TV_loss_app: sum_i reg(plane_i) * 1e-2 (tensoRF.py:218-228)."""
    total = 0
    for p in planes:
        total = total + tv_loss(p) * 1e-2
    return total


def tv_depth(depth, batch, grid_H, grid_W):
    """model/tensorf.py:133-138."""
    d = depth.reshape(batch, grid_H, grid_W)
    return torch.pow(d[:, 1:, :] - d[:, :-1, :], 2).sum() / grid_H + torch.pow(d[:, :, 1:] - d[:, :, :-1], 2).sum() / grid_W


# ----------------------------------------------------------------------------------------------
# N2  factor upsampling                                                tensoRF.py:274-295
# ----------------------------------------------------------------------------------------------
def upsample_vm(planes, lines, res_target):
    out_p, out_l = [], []
    for i in range(3):
        m0, m1 = MAT_MODE[i]
        out_p.append(F.interpolate(planes[i], size=(res_target[m1], res_target[m0]), mode="bilinear", align_corners=True))
        out_l.append(F.interpolate(lines[i], size=(res_target[VEC_MODE[i]], 1), mode="bilinear", align_corners=True))
    return out_p, out_l


# ----------------------------------------------------------------------------------------------
# parameter helpers
# ----------------------------------------------------------------------------------------------
def init_params(gridSize, density_n_comp=(16, 16, 16), app_n_comp=(48, 48, 48), app_dim=27, featureC=64,
                view_pe=2, fea_pe=2, shadingMode="MLP_Fea", scale=0.1, bias=0.0, generator=None, device="cpu"):
    """Random-init parameters with the reference's distributions (tensoRF.py:159-169:
    abs(bias + scale*randn); Linear default init; last bias 0, tensorBase.py:114,196).
    Draw ORDER differs from the reference (it builds two NeRFs) -- only the
    distribution matters for synthetic benchmarks."""
    g = generator
    gs = list(gridSize)

    def rn(*shape):
        return torch.abs(bias + scale * torch.randn(*shape, generator=g))

    def lin(out_f, in_f):
        bound = 1 / math.sqrt(in_f)
        w = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound
        b = (torch.rand(out_f, generator=g) * 2 - 1) * bound
        return w, b

    p = dict(density_plane=[], density_line=[], app_plane=[], app_line=[])
    for name, comps in (("density", density_n_comp), ("app", app_n_comp)):
        for i in range(3):
            m0, m1 = MAT_MODE[i]
            p[name + "_plane"].append(rn(1, comps[i], gs[m1], gs[m0]))
            p[name + "_line"].append(rn(1, comps[i], gs[VEC_MODE[i]], 1))
    p["basis"] = lin(app_dim, sum(app_n_comp))[0]
    if shadingMode == "MLP_Fea_WeakView":
        in1, in3 = (2 * fea_pe + 1) * app_dim, featureC + 2 * view_pe * 3
    else:
        in1, in3 = 2 * view_pe * 3 + 2 * fea_pe * app_dim + 3 + app_dim, featureC
    w1, b1 = lin(featureC, in1)
    w2, b2 = lin(featureC, featureC)
    w3, b3 = lin(3, in3)
    p["mlp"] = dict(w1=w1, b1=b1, w2=w2, b2=b2, w3=w3, b3=torch.zeros_like(b3))

    def mv(x):
        if isinstance(x, dict):
            return {k: mv(v) for k, v in x.items()}
        if isinstance(x, list):
            return [mv(v) for v in x]
        return x.to(device)

    return mv(p)


def params_from_state_dict(sd, prefix="nerf.tensorf."):
    """Map the reference's state_dict key names (SURVEY §5 checkpoint row) to the oracle's dict."""
    def get(k):
        v = sd[prefix + k]
        return torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v

    p = dict(
        density_plane=[get("density_plane.%d" % i) for i in range(3)],
        density_line=[get("density_line.%d" % i) for i in range(3)],
        app_plane=[get("app_plane.%d" % i) for i in range(3)],
        app_line=[get("app_line.%d" % i) for i in range(3)],
        basis=get("basis_mat.weight"),
    )
    if (prefix + "renderModule.mlp.0.weight") in sd:
        names = ("renderModule.mlp.0", "renderModule.mlp.2", "renderModule.mlp.4")
    else:
        names = ("renderModule.layer1", "renderModule.layer2", "renderModule.layer3")
    p["mlp"] = dict(w1=get(names[0] + ".weight"), b1=get(names[0] + ".bias"),
                    w2=get(names[1] + ".weight"), b2=get(names[1] + ".bias"),
                    w3=get(names[2] + ".weight"), b3=get(names[2] + ".bias"))
    return p


def flat_params(p):
    """Deterministic list of (name, tensor) in the reference's state_dict naming."""
    out = []
    for grp in ("density_plane", "density_line", "app_plane", "app_line"):
        for i in range(3):
            out.append(("%s.%d" % (grp, i), p[grp][i]))
    out.append(("basis_mat.weight", p["basis"]))
    for k in ("w1", "b1", "w2", "b2", "w3", "b3"):
        out.append(("mlp." + k, p["mlp"][k]))
    return out
