"""Does the GPU reach its clocks within bench.py's warm-up?  The in-step HIP-event timers of every forward / backward launch of a
fresh process, in launch order (priming, warm-up and timed steps alike).  GPU box only.
  python tools/round6/clock_ramp.py [bench.py flags]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bench
from joint_tensorf_amd import ops
keep = []
orig_main = bench.main
real_inst = bench.instep_roofline


def spy(timers, *a, **k):
    keep.extend(timers)
    return real_inst(timers, *a, **k)


bench.instep_roofline = spy
t0 = time.perf_counter()
bench.main()
torch.cuda.synchronize()
for kind in ("fwd", "bwd"):
    ms = [t[1].elapsed_time(t[2]) for t in keep if t[0] == kind]
    print(kind, "ms per launch in launch order:", " ".join("%.3f" % v for v in ms))
