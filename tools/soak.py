#!/usr/bin/env python3
"""Run the training loop of bat_blender_VM across the schedule's transitions at full scale: the 252^3 -> 400^3
upsampling (optimizer rebuilt, sample count changes) at iteration 9000 and the end of the blur schedule at 12000.
Checks that the loss stays finite and reports step times and peak memory.
usage: python tools/soak.py [first_it] [last_it]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8990, 9015)
    sys.argv = [sys.argv[0]]
    import bench
    from joint_tensorf_amd.options import make_options, Opt
    from joint_tensorf_amd.synthetic import make_views
    dev = "cuda:0"
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    np.random.seed(0)
    opt = make_options("bat_blender_VM", device=dev)
    ups = list(opt.train_schedule.upsample_iters)
    stage = sum(1 for u in ups if u <= a)          # grid stage that iteration `a` trains at
    _, _ = bench.stage_setup(opt, stage)
    opt.train_schedule.upsample_iters = [u for u in ups if u > a] or [10 ** 9]
    opt.nerf.n_rays = opt.train_schedule.n_rays_rest
    model = bench.build_model(opt, a, int(opt.data.num_views))
    views = make_views(opt, int(opt.data.num_views), seed=0, device=dev)
    log = []
    for it in range(a, b):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = model.train_iteration(opt, Opt(dict(views)))
        model.after_iteration(opt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        tf = model.graph.nerf.tensorf
        log.append(dict(it=it, ms=round(dt, 2), loss=float(loss.all), grid=tf.gridSize.tolist(), S=model.graph.nerf.n_samples,
                        blur=model.graph.resolve_blur(opt, "vis")[2] is not None,
                        lr=round(model.optim.param_groups[0]["lr"], 6)))
        assert np.isfinite(log[-1]["loss"]), log[-1]
    for r in log:
        print(json.dumps(r))
    print("peak memory %.1f GB" % (torch.cuda.max_memory_allocated() / 2 ** 30))


if __name__ == "__main__":
    main()
