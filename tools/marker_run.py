#!/usr/bin/env python3
"""A few training iterations with opt.profiling set: Model.train_iteration opens roctx ranges named like the reference's
`record_function` blocks (model/base.py:119-153, tensorBase.py:774).  Run under
    rocprofv3 --marker-trace --stats -d <out> -- python3 tools/marker_run.py
and the marker summary lists them (profiles/round3_marker_trace_summary.txt)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from joint_tensorf_amd.options import Opt, make_options
    from joint_tensorf_amd.synthetic import make_views
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    np.random.seed(0)
    opt = make_options("bat_blender_VM", device="cuda:0")
    opt.profiling = True
    stage, it0 = bench.stage_setup(opt, -1)
    opt.nerf.n_rays = opt.train_schedule.n_rays_rest
    model = bench.build_model(opt, it0, int(opt.data.num_views))
    var = make_views(opt, int(opt.data.num_views), seed=0, device="cuda:0")
    for _ in range(12):
        model.train_iteration(opt, Opt(dict(var)))
        model.after_iteration(opt)
    torch.cuda.synchronize()
    print("marker_run: 12 iterations, profiling ranges on:", bool(__import__("joint_tensorf_amd.ops", fromlist=["x"]).PROFILING))


if __name__ == "__main__":
    main()
