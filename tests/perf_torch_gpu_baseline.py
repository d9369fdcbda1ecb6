#!/usr/bin/env python3
"""Not a pytest file.  Times the reference ALGORITHM in stock torch ops on the GPU (the parity-pinned oracle
with device="cuda"): the "reference single-GPU PyTorch rays/s" denominator of BASELINE.md §3.2, on the same
synthetic scene / stage bench.py uses.

  python tests/perf_torch_gpu_baseline.py --stage 4 --steps 5 [--full-raygen]

--full-raygen additionally builds the rays of ALL H*W pixels of all views and indexes afterwards, which is
what the reference's camera.get_center_and_ray + tensorf.Graph.render do (camera.py:231-261,
model/tensorf.py:154-161); without it only the sampled pixels are generated (leaner than the reference)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import tensorf_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="bat_blender_VM")
    ap.add_argument("--stage", type=int, default=-1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--full-raygen", action="store_true")
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args()
    from joint_tensorf_amd.options import make_options
    from joint_tensorf_amd.synthetic import make_views
    import bench
    dev = args.device
    opt = make_options(args.config, device=dev)
    stage, it0 = bench.stage_setup(opt, args.stage)
    n_rays = opt.train_schedule.n_rays_init if it0 < opt.train_schedule.change_n_rays_after_n_iters else opt.train_schedule.n_rays_rest
    bbox = opt.data.scene_bbox
    res = O.find_resolution(bbox, opt.train_schedule.n_voxel_init)
    S = O.find_n_samples(res, opt.nerf.step_ratio, opt.nerf.sample_intvs)
    cfg = O.SceneCfg(bbox, res, list(opt.nerf.depth.range), step_ratio=opt.nerf.step_ratio).to(dev)
    g = torch.Generator().manual_seed(0)
    params = O.init_params(res, scale=0.1, bias=0.0, generator=g, device=dev)
    leaves = [v for _, v in O.flat_params(params)]
    for v in leaves:
        v.requires_grad_(True)
    B = int(opt.data.num_views)
    var = make_views(opt, B, seed=0, device=dev)
    se3 = torch.zeros(B, 6, device=dev, requires_grad=True)
    noise = O.se3_to_SE3(torch.randn(B, 6, generator=g).to(dev) * 0.15)
    optim = torch.optim.Adam(leaves, lr=1e-2, betas=(0.9, 0.99))
    optim_pose = torch.optim.Adam([se3], lr=1e-3)
    progress = it0 / opt.max_iter
    H, W = opt.H, opt.W
    image = var.image.view(B, 3, H * W).permute(0, 2, 1)
    rays_done = 0

    def step():
        nonlocal rays_done
        pd, pc = O.resolve_blur(progress, opt.c2f_schedule_density, opt.c2f_schedule_color, "train",
                                np.random.choice(opt.c2f_random_density_scale_pool))
        kd = kc = None
        if pd is not None:
            kd = O.get_kernel(cfg, pd, opt.c2f_kernel_size).to(dev)
            kc = O.get_kernel(cfg, pc, opt.c2f_kernel_size).to(dev)
        step_px = int(np.ceil((H * W // (n_rays // B)) ** 0.5))
        ox, oy = np.random.randint(step_px), np.random.randint(step_px)
        sx = torch.arange(ox, W, step_px, device=dev)
        sy = torch.arange(oy, H, step_px, device=dev)
        gY, gX = torch.meshgrid(sy, sx, indexing="ij")
        ray_idx = (gX + gY * W).view(-1)
        pose = O.train_pose(se3, noise, var.pose)
        if args.full_raygen:
            allidx = torch.arange(H * W, device=dev)
            c_all, r_all = O.rays_for_pixels(pose, var.intr_inv, allidx, W)
            c, r = c_all[:, ray_idx], r_all[:, ray_idx]
        else:
            c, r = O.rays_for_pixels(pose, var.intr_inv, ray_idx, W)
        jit = torch.rand(c.shape[0] * c.shape[1], 1, device=dev)
        rgb, _, _ = O.render(cfg, params, c.reshape(-1, 3), r.reshape(-1, 3), S, white_bg=True, jitter=jit,
                             kernel_density=kd, kernel_color=kc)
        rgb = rgb.view(B, -1, 3)
        loss = O.render_loss(rgb, image[:, ray_idx]) + 8e-5 * O.density_L1(params) \
            + 0.0 * O.tv_planes(params["density_plane"]) + 0.0 * O.tv_planes(params["app_plane"])
        optim.zero_grad()
        optim_pose.zero_grad()
        loss.backward()
        optim.step()
        optim_pose.step()
        rays_done += rgb.shape[0] * rgb.shape[1]

    for _ in range(args.warmup):
        step()
    if dev.startswith("cuda"):
        torch.cuda.synchronize()
    rays_done = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dev.startswith("cuda"):
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(kind="torch-stock-ops", device=dev, config=args.config, stage=stage, grid=res, S=S,
                          rays_per_iter=rays_done / args.steps, ms_per_step=dt / args.steps * 1e3,
                          rays_per_s=rays_done / dt, full_raygen=args.full_raygen)))


if __name__ == "__main__":
    main()
