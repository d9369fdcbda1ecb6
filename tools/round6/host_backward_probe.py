"""Which call inside ops.RenderRays.backward (autograd thread) takes the time in the slow steps: sys.setprofile inside the thread,
calls longer than 30 us logged with their line.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bench
from joint_tensorf_amd import ops
pc = time.perf_counter
LOGS = []
f0 = ops.RenderRays.backward


def bwd(*a, **k):
    stack, log = [], []

    def prof(frame, event, arg):
        t = pc()
        if event in ("call", "c_call"):
            stack.append((t, frame.f_code.co_name if event == "call" else getattr(arg, "__name__", str(arg)), frame.f_lineno))
        elif stack:
            t0, name, line = stack.pop()
            if t - t0 > 30e-6:
                log.append((t0, t - t0, name, line, len(stack)))
    t0 = pc()
    sys.setprofile(prof)
    try:
        return f0(*a, **k)
    finally:
        sys.setprofile(None)
        LOGS.append((pc() - t0, t0, log))


ops.RenderRays.backward = staticmethod(bwd)
bench.main()
for total, t0, log in LOGS[-6:]:
    print("RenderRays.backward %.0f us:" % (total * 1e6))
    for s, d, name, line, depth in sorted(log):
        print("   +%7.1f us  %s%-32s %7.1f us   (caller line %d)" % ((s - t0) * 1e6, "  " * depth, name, d * 1e6, line))
