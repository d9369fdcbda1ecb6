#!/usr/bin/env python3
"""Where does the HOST time of one eager train step go?  cProfile over N steps, own time per step in microseconds."""
import cProfile
import os
import pstats
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from joint_tensorf_amd.options import Opt, make_options  # noqa: E402
from joint_tensorf_amd.synthetic import make_views  # noqa: E402

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = 50
torch.manual_seed(0)
np.random.seed(0)
opt = make_options("bat_blender_VM", device="cuda:0")
stage, it0 = bench.stage_setup(opt, stage)
opt.nerf.n_rays = 2048
model = bench.build_model(opt, it0, 100)
var_all = make_views(opt, 100, seed=0, device="cuda:0")


def step():
    model.train_iteration(opt, Opt(dict(var_all)))
    model.after_iteration(opt)


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
rows = sorted(((v[2], v[3], v[0], k) for k, v in st.stats.items()), reverse=True)
tot = sum(r[0] for r in rows)
print("stage %d: %.0f us of host time per step under cProfile" % (stage, tot / N * 1e6))
for tt, ct, nc, (f, ln, fn) in rows[:32]:
    print("%7.1f us own %7.1f us cum %5.1f calls  %s:%d %s" % (tt / N * 1e6, ct / N * 1e6, nc / N, os.path.basename(f), ln, fn))
