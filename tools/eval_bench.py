#!/usr/bin/env python3
"""Throughput of the evaluation path on one GPU (BASELINE.json configs[4] shape): full-image sliced render of one
view at the final grid, and test-time photometric pose optimisation steps (pose-only backward).
usage: python tools/eval_bench.py [--size 800] [--samples 1024] [--test-iters 50]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--test-iters", type=int, default=50)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--scene", default="random", choices=["random", "blobs"], help="see bench.py --scene")
    ap.add_argument("--graph", action="store_true", help="replay the sliced render from one hipGraph (nerf.eval_graph)")
    ap.add_argument("--fused", action="store_true",
                    help="test-time optimisation through the single-launch fused kernel (csrc/jt_fused.hip, "
                         "opt.optim.test_fused) instead of the staged kernels + pose-only backward")
    ap.add_argument("--no-render", action="store_true", help="skip the full-image render timing")
    ap.add_argument("--batch-views", default="", help="also time the BATCHED test-time optimisation (Model."
                    "evaluate_test_time_photometric_optim_batched) for these view counts, e.g. 1,8,32: ms per view-iteration")
    args = ap.parse_args()
    sys.argv = [sys.argv[0]]
    import bench
    from joint_tensorf_amd.options import make_options, Opt
    from joint_tensorf_amd.synthetic import make_views
    dev = "cuda:0"
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    np.random.seed(0)
    opt = make_options("bat_blender_VM", device=dev, data=dict(image_size=[args.size, args.size]),
                       nerf=dict(sample_intvs=args.samples), optim=dict(test_iter=args.test_iters))
    opt.nerf.eval_graph = bool(args.graph)
    opt.optim.test_graph = bool(args.graph)
    opt.optim.test_fused = bool(args.fused)
    stage, it0 = bench.stage_setup(opt, -1)
    opt.nerf.n_rays = opt.train_schedule.n_rays_rest
    model = bench.build_model(opt, it0, int(opt.data.num_views))
    if args.scene == "blobs":
        from joint_tensorf_amd.synthetic import bake_blobs
        bake_blobs(model.graph.nerf.tensorf, n_blobs=12, seed=0)
    views = make_views(opt, int(opt.data.num_views), seed=0, device=dev)
    pose, pose_GT = model.get_all_training_poses(opt, views["pose"])
    _, model.graph.sim3 = model.prealign_cameras(opt, pose, pose_GT)
    tv = make_views(opt, 1, seed=5, device=dev)
    var = Opt(dict(tv))
    var.idx = torch.arange(1, device=dev)
    g = model.graph
    g.eval()
    old_photo = opt.optim.test_photo
    opt.optim.test_photo = False
    t_render = float("nan")
    with torch.no_grad():
        if args.no_render:
            raise_skip = True
        else:
            raise_skip = False
            g.forward(opt, Opt(dict(var)), mode="eval")
        if not raise_skip:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                g.forward(opt, Opt(dict(var)), mode="eval")
            torch.cuda.synchronize()
            t_render = (time.perf_counter() - t0) / args.reps
    opt.optim.test_photo = old_photo
    torch.cuda.synchronize()
    model.evaluate_test_time_photometric_optim(opt, Opt(dict(var)))  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    v = model.evaluate_test_time_photometric_optim(opt, Opt(dict(var)))
    torch.cuda.synchronize()
    t_opt = (time.perf_counter() - t0) / args.test_iters
    batched = {}
    if args.batch_views and not args.graph and not args.fused:
        for V in [int(x) for x in args.batch_views.split(",")]:
            bvs = make_views(opt, V, seed=9, device=dev)
            vs = [Opt(idx=torch.arange(1, device=dev), pose=bvs.pose[i:i + 1], intr=bvs.intr[i:i + 1],
                      intr_inv=bvs.intr_inv[i:i + 1], image=bvs.image[i:i + 1]) for i in range(V)]
            model.evaluate_test_time_photometric_optim_batched(opt, [Opt(dict(x)) for x in vs])   # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.evaluate_test_time_photometric_optim_batched(opt, [Opt(dict(x)) for x in vs])
            torch.cuda.synchronize()
            batched[str(V)] = (time.perf_counter() - t0) / args.test_iters / V * 1e3
    rays = args.size * args.size
    print(json.dumps({
        "test_time_optim_batched_ms_per_view_iteration": batched or None,
        "eval_render": {"image": [args.size, args.size], "samples_per_ray": int(g.nerf.n_samples),
                        "grid": g.nerf.tensorf.gridSize.tolist(), "ms_per_image": t_render * 1e3,
                        "rays_per_s": rays / t_render, "launch": "hipGraph replay" if args.graph else "eager", "Msamples_per_s": rays * g.nerf.n_samples / t_render / 1e6},
        "test_time_optim": {"rays_per_iter": int(v.rgb.shape[1]), "ms_per_iter": t_opt * 1e3,
                            "iters": args.test_iters, "scene": args.scene,
                            "launch": dict(model._test_optim_graph.stats) if args.graph else "eager", "backward": "pose-only (no factor / weight gradients)",
                            "path": "staged kernels + autograd" if not args.fused else "single launch: render + loss + backward "
                                                                                    "to the rays (jt_pose_fused)"}}))


if __name__ == "__main__":
    main()
