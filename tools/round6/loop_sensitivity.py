"""How sensitive is the REFERENCE's own training loop to round-off?  (build container only: imports /root/reference)

tests/test_engine_trace.py runs the build's loop on the draws of a recorded run of the reference's loop and finds the LLFF
losses agreeing to 1e-6 for 19 iterations, then 6e-4, then 0.3 -- at the iterations where the scheduled near plane passes
through zero.  This script shows the reference does the same to ITSELF: its loop on the same tiny scene and the same draws,
once as it is and once with every scene parameter multiplied by (1 + eps N(0,1)), eps = 1e-7 / 1e-6; printed is the relative
deviation of the photometric loss per iteration.  Result (profiles/round6_llff_loop_sensitivity.txt): 1e-7 in, 3e-3 out at
iteration 20 and 0.3 at iterations 39-41 -- the loop amplifies round-off by six orders of magnitude within 40 iterations,
so trajectories can only be compared BEFORE the first such event; later states are pinned by the in-loop renderer fixtures
(tools/make_engine_trace.py --snapshot: llff_loop_it21 / it40).

  python tools/round6/loop_sensitivity.py 0; python tools/round6/loop_sensitivity.py 1e-7; python tools/round6/loop_sensitivity.py 1e-6
"""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import make_engine_trace as MT
eps = float(sys.argv[1])
case, opt, var, m, bat, camera, seed = MT.build_reference_model("llff")
if eps:
    with torch.no_grad():
        g = torch.Generator().manual_seed(1)
        for p in m.graph.nerf.tensorf.parameters():
            p.mul_(1 + eps * torch.randn(p.shape, generator=g))
# same draws: reseed right before training
torch.manual_seed(123); np.random.seed(123)
losses = []
orig = m.train_iteration
def spy(o, v, l):
    loss = orig(o, v, l); losses.append(float(loss.render)); return loss
m.train_iteration = spy
m.train(opt)
json.dump(losses, open('/tmp/sens_%s.json' % sys.argv[1], 'w'))
