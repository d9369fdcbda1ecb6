#!/usr/bin/env python3
"""How long does the HOST need to enqueue one train step (no GPU sync inside the loop)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from joint_tensorf_amd.options import make_options, Opt
from joint_tensorf_amd.synthetic import make_views
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda:0"
torch.manual_seed(0); np.random.seed(0)
opt = make_options("bat_blender_VM", device=dev)
stage, it0 = bench.stage_setup(opt, stage)
opt.nerf.n_rays = 2048
model = bench.build_model(opt, it0, 100)
var_all = make_views(opt, 100, seed=0, device=dev)
def step():
    var = Opt(dict(var_all))
    model.train_iteration(opt, var)
    model.after_iteration(opt)
for _ in range(5): step()
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("stage %d: host enqueue %.3f ms/step, total %.3f ms/step" % (stage, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
