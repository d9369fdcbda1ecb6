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
