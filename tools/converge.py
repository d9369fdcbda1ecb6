#!/usr/bin/env python3
"""The reference's acceptance flow on the self-consistent synthetic scene (model/bat.py:211-263, model/nerf.py:525-572):
train the joint pose + field optimisation from perturbed cameras (camera.noise) on images RENDERED from a known field,
then align the recovered cameras to the ground truth (Procrustes) and report rotation / translation error and held-out
PSNR -- through `bat_hip.Model`'s own lifecycle (load_dataset -> build_networks -> setup_optimizer -> train ->
evaluate_full).  Prints one JSON line per checkpoint of the run and a final one.

  python tools/converge.py [--config bat_blender_VM] [--compress 10] [--image-size 200] [--views 40] [--graph]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(args, device="cuda:0"):
    from joint_tensorf_amd.model import bat_hip
    from joint_tensorf_amd.options import compress_schedule, make_options
    over = dict(data=dict(synthetic="rendered", num_views=args.views, num_test_views=args.test_views, gt_res=args.gt_res))
    if getattr(args, "llff_baseline", 0):
        over["data"]["llff_baseline"] = args.llff_baseline
    if getattr(args, "llff_focus", 0):
        over["data"]["llff_focus"] = args.llff_focus
    if getattr(args, "gt_z_range", None):
        over["data"]["gt_z_range"] = [float(v) for v in args.gt_z_range.split(",")]
    if getattr(args, "gt_wall", 0):
        over["data"]["gt_wall"] = args.gt_wall
    for k in ("gt_stairs", "gt_blobs", "gt_stairs_near", "llff_zspread"):
        if getattr(args, k, 0):
            over["data"][k] = getattr(args, k)
    if getattr(args, "gt_blob_radius", None):
        over["data"]["gt_blob_radius"] = [float(v) for v in args.gt_blob_radius.split(",")]
    opt = make_options(args.config, device=device, **over)
    if args.image_size:
        h, w = opt.data.image_size
        s = args.image_size / float(max(h, w))
        opt.data.image_size = [int(round(h * s)), int(round(w * s))]
        opt.H, opt.W = opt.data.image_size
    if args.n_voxel_final:
        opt.train_schedule.n_voxel_final = args.n_voxel_final
    if args.noise is not None:
        opt.camera.noise = args.noise
    if args.n_rays:
        opt.train_schedule.n_rays_init = opt.train_schedule.n_rays_rest = opt.nerf.n_rays = args.n_rays
    compress_schedule(opt, args.compress)
    if args.max_iter:
        opt.early_stop_iter = args.max_iter
    opt.seed = args.seed
    opt.optim.test_iter = args.test_iter
    opt.optim.test_photo = args.test_iter > 0
    opt.train_graph = bool(args.graph)
    opt.freq = dict(scalar=0, val=0, ckpt=0)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    model = bat_hip.Model(opt)
    model.load_dataset(opt, eval_split="test")
    model.build_networks(opt)
    model.setup_optimizer(opt)
    return opt, model


def pose_errors(opt, model):
    pose, pose_gt = model.get_all_training_poses(opt)
    aligned, _ = model.prealign_cameras(opt, pose, pose_gt)
    err = model.evaluate_camera_alignment(opt, aligned, pose_gt)
    return float(np.rad2deg(err.R.mean().item())), float(err.t.mean().item())


def relative_rotation_error(opt, model):
    """Alignment-free: mean angle between R_i R_0^T of the recovered cameras and of the ground truth (degrees).  A common
    rotation of the whole rig -- what the Procrustes fit of a small planar camera cloud cannot pin down -- cancels."""
    pose, pose_gt = model.get_all_training_poses(opt)
    rel = pose[:, :, :3] @ pose[:1, :, :3].transpose(-2, -1)
    rel_gt = pose_gt[:, :, :3] @ pose_gt[:1, :, :3].transpose(-2, -1)
    d = rel @ rel_gt.transpose(-2, -1)
    tr = d[:, 0, 0] + d[:, 1, 1] + d[:, 2, 2]
    return float(np.rad2deg(((tr - 1) / 2).clamp(-1 + 1e-7, 1 - 1e-7).acos()[1:].mean().item()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="bat_blender_VM")
    ap.add_argument("--compress", type=float, default=10.0, help="divide every iteration-denominated schedule key by this")
    ap.add_argument("--image-size", type=int, default=200, help="longer image side in pixels (0 = the yaml's)")
    ap.add_argument("--views", type=int, default=40)
    ap.add_argument("--test-views", type=int, default=4)
    ap.add_argument("--gt-res", type=int, default=128)
    ap.add_argument("--n-voxel-final", type=int, default=0)
    ap.add_argument("--n-rays", type=int, default=0)
    ap.add_argument("--noise", type=float, default=None)
    ap.add_argument("--max-iter", type=int, default=0, help="stop early at this (compressed) iteration")
    ap.add_argument("--test-iter", type=int, default=0, help="test-time pose optimisation iterations per held-out view")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--report-every", type=int, default=0)
    ap.add_argument("--llff-baseline", type=float, default=0.0)
    ap.add_argument("--llff-focus", type=float, default=0.0)
    ap.add_argument("--gt-z-range", default=None, help="LLFF scene: fraction range of the box's z the blobs sit in, e.g. 0.6,0.9")
    ap.add_argument("--gt-wall", type=float, default=0.0, help="LLFF scene: z fraction of the back wall")
    ap.add_argument("--gt-stairs", type=int, default=0, help="LLFF scene: back wall in this many depth steps across the picture")
    ap.add_argument("--gt-stairs-near", type=float, default=0.0, help="LLFF scene: z fraction of the nearest wall step")
    ap.add_argument("--gt-blobs", type=int, default=0)
    ap.add_argument("--llff-zspread", type=float, default=0.0, help="LLFF scene: depth of the camera cloud / its width (1/6)")
    ap.add_argument("--gt-blob-radius", default=None, help="LLFF scene: blob radius range, e.g. 0.15,0.35")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    opt, model = build(args)
    r0, t0 = pose_errors(opt, model)
    print(json.dumps(dict(it=0, rot_deg=round(r0, 4), trans=round(t0, 5), dataset=opt.data.dataset_class,
                          image=[opt.H, opt.W], views=args.views, max_iter=int(opt.max_iter))), flush=True)
    every = args.report_every or max(1, int(opt.max_iter) // 10)
    orig_after = model.after_iteration
    wall = [time.perf_counter()]

    def after(o, it=None):
        orig_after(o, it)
        if model.it % every == 0:
            torch.cuda.synchronize()
            r, t = pose_errors(opt, model)
            tf = model.graph.nerf.tensorf
            now = time.perf_counter()
            print(json.dumps(dict(it=model.it, rot_deg=round(r, 4), trans=round(t, 5), grid=tf.gridSize.tolist(),
                                  ms_per_iter=round((now - wall[0]) / every * 1e3, 3))), flush=True)
            wall[0] = time.perf_counter()
    model.after_iteration = after
    t_start = time.perf_counter()
    loss = model.train(opt)
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t_start
    r1, t1 = pose_errors(opt, model)
    rrel = relative_rotation_error(opt, model)
    res = model.evaluate_full(opt)
    print(json.dumps(dict(final=True, iterations=model.it, train_seconds=round(t_train, 2), loss=round(float(loss.all), 6),
                          rot_deg_start=round(r0, 4), rot_deg_end=round(r1, 4), trans_start=round(t0, 5),
                          trans_end=round(t1, 5), rot_gain=round(r0 / max(r1, 1e-9), 1),
                          trans_gain=round(t0 / max(t1, 1e-9), 1), rot_rel_deg_end=round(rrel, 4), test_psnr=round(res.psnr, 2),
                          psnr_per_view=[round(p, 2) for p in res.psnr_per_view])), flush=True)
    st = getattr(model, "train_stepper", None)
    if st is not None:  # which grid stages ran replayed, which eager: (iteration, grid, choice, host ms eager, GPU ms replayed)
        print(json.dumps(dict(launch=st.stats, launch_mode_per_stage=st.decisions)), flush=True)


if __name__ == "__main__":
    main()
