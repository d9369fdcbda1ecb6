"""Synthetic stand-in for the Blender / LLFF training data (no dataset can be downloaded here).

Produces the `var` dict the reference's data loaders hand to Graph.forward
(data/blender.py:19-91, data/llff.py:43-97): idx, image [N,3,H,W], intr, intr_inv, pose [N,3,4]
(world->camera, camera looking along +z), plus Bernoulli edge masks (model/nerf.py:115-149 output shape).
"""
import math

import numpy as np
import torch

from .options import Opt


def look_at(eye):
    eye = np.asarray(eye, dtype=np.float64)
    fwd = -eye / np.linalg.norm(eye)
    up = np.array([0.0, 0.0, 1.0])
    if abs(fwd @ up) > 0.99:
        up = np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], 0)
    return np.concatenate([R, (-R @ eye)[:, None]], 1).astype(np.float32)


def make_views(opt, n_views, seed=0, device="cpu", with_images=True):
    rng = np.random.RandomState(seed)
    H, W = opt.H, opt.W
    poses = []
    if opt.data.dataset == "blender":
        # cameras on the upper hemisphere of radius 4 (inside the yaml depth range [2,6]) looking at the origin
        for i in range(n_views):
            th = 2 * math.pi * rng.rand()
            ph = math.radians(5 + 75 * rng.rand())
            eye = 4.0 * np.array([math.cos(th) * math.cos(ph), math.sin(th) * math.cos(ph), math.sin(ph)])
            poses.append(look_at(eye))
        f = 0.5 * W / math.tan(0.5 * 0.69)  # camera_angle_x ~ 0.69 rad (data/blender.py:29)
    else:
        for i in range(n_views):
            P = np.eye(3, 4, dtype=np.float32)
            P[:, 3] = -np.array([0.3 * (rng.rand() - 0.5), 0.3 * (rng.rand() - 0.5), 0.05 * (rng.rand() - 0.5)])
            poses.append(P)
        f = 0.8 * W
    pose = torch.tensor(np.stack(poses))
    intr = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=torch.float32)[None].repeat(n_views, 1, 1)
    var = Opt(idx=torch.arange(n_views, device=device), pose=pose.to(device), intr=intr.to(device),
              intr_inv=intr.inverse().to(device))
    if with_images:
        g = torch.Generator().manual_seed(seed + 1)
        var.image = torch.rand(n_views, 3, H, W, generator=g).to(device)
        var.train_edge_masks = (torch.rand(n_views, H * W, generator=g) < 0.5).to(torch.uint8).to(device)
    return var


@torch.no_grad()
def bake_blobs(tensorf, n_blobs=12, seed=0, amplitude=60.0, radius=(0.12, 0.3), background=-12.0):
    """Overwrite the DENSITY factors with a few opaque Gaussian blobs (the "structured" scene).

    exp(-|p - c|^2 / 2 s^2) factorises into an (x, y) plane times a z line, i.e. one blob is exactly one rank-1
    VM component of plane 0 / line 0; the last component is the constant `background` (what a trained field's empty
    space looks like: softplus(-12 - 10) gives weights of 1e-11, far below rayMarch_weight_thres, where the
    zero feature of an untouched grid still gives 4e-6); all other density components are zeroed.  Only the few
    samples at a blob's surface are shaded -- the regime of a trained scene (a few per cent of the in-box samples), against
    the random-init field in which every in-box sample is shaded."""
    rng = np.random.RandomState(seed)
    dev = tensorf.density_plane[0].device
    lo, hi = tensorf.aabb[0].to(dev).float(), tensorf.aabb[1].to(dev).float()
    C = tensorf.density_plane[0].shape[1]
    n_blobs = min(int(n_blobs), C - 1)
    for p in list(tensorf.density_plane) + list(tensorf.density_line):
        p.zero_()
    plane, line = tensorf.density_plane[0], tensorf.density_line[0]   # plane 0: (x -> W, y -> H); line 0: z
    H, W, L = plane.shape[2], plane.shape[3], line.shape[2]
    X = lo[0] + (hi[0] - lo[0]) * torch.linspace(0, 1, W, device=dev)
    Y = lo[1] + (hi[1] - lo[1]) * torch.linspace(0, 1, H, device=dev)
    Z = lo[2] + (hi[2] - lo[2]) * torch.linspace(0, 1, L, device=dev)
    for k in range(n_blobs):
        c = (lo + (hi - lo) * torch.tensor(0.25 + 0.5 * rng.rand(3), device=dev, dtype=torch.float32))
        s = float(radius[0] + (radius[1] - radius[0]) * rng.rand())
        plane[0, k] = amplitude * torch.exp(-((X[None, :] - c[0]) ** 2 + (Y[:, None] - c[1]) ** 2) / (2 * s * s))
        line[0, k, :, 0] = torch.exp(-((Z - c[2]) ** 2) / (2 * s * s))
    plane[0, C - 1] = background
    line[0, C - 1] = 1.0
    return n_blobs
