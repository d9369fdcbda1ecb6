"""cProfile of the steady-state eager steps only (enabled at the 60th Graph.forward of the process): which Python frames the
host-bound step spends its time in.  Inflates every frame; read it as a ranking.  GPU box only."""
import os, sys, cProfile, pstats, io, gc
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bench
from joint_tensorf_amd.model import bat_hip
pr = cProfile.Profile()
n = [0]
f0 = bat_hip.Graph.forward
GC = []
def cb(phase, info):
    import time
    if phase == "start":
        GC.append([info["generation"], time.perf_counter()])
    else:
        GC[-1].append(time.perf_counter())
gc.callbacks.append(cb)
def fwd(self, *a, **k):
    n[0] += 1
    if n[0] == 60:
        del GC[:]
        pr.enable()
    return f0(self, *a, **k)
bat_hip.Graph.forward = fwd
try:
    bench.main()
finally:
    pr.disable()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s)
    st.sort_stats("tottime").print_stats(70)
    print(s.getvalue()[:14000])
    steps = n[0] - 60
    done = [g for g in GC if len(g) == 3]
    for gen in (0, 1, 2):
        d = [g[2] - g[1] for g in done if g[0] == gen]
        if d:
            print("gc generation %d: %d collections over %d steps, mean %.0f us, max %.0f us, %.1f us per step"
                  % (gen, len(d), steps, 1e6 * sum(d) / len(d), 1e6 * max(d), 1e6 * sum(d) / steps))
